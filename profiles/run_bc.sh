mkdir -p gpurun_out/bc
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches or decode_rgb or early_termination or split_march or specialised" > gpurun_out/bc/pytest.txt 2>&1
tail -4 gpurun_out/bc/pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/bc/level3.txt
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import ngf_amd
from ngf_amd import synth
from ngf_amd.cases import big_case, field_for_case
full = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
for preset in ("R1", "R2"):
    g, params, step = big_case("triplane", preset)
    for bc in (False, True):
        f = field_for_case(g, params, None, device="cuda", bake=True, bake_color=bc)
        for _ in range(3): f(full, N_samples=192, iteration=30001)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); f(full, N_samples=192, iteration=30001); b.record()
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        print(f"{preset} bake_color={bc}: {ms:.3f} ms = {640/ms:.1f} Mray/s")
        f.release()
PY
