"""Narrowing the InfoInv NGF_F_SPLIT_BF16 nondeterminism (profiles/exp_determinism_hammer.py: the split RENDER of infoinv_r1_on moved
in 2 of 600 rounds -- one 4-ray tile each time, |diff| <= 3e-4 -- while its march outputs, the fp32 kernel and the oracle never moved).
ONE handle per case, many launches, every output compared bitwise with the first:
    render (march + colour pass), decode_rgb (colour pass alone on fixed samples), for InfoInv split / fp32 and TriPlane split.
    python profiles/exp_determinism_fast.py [launches] [poison] [tile_w] [ablate] [case substring]
NGF_LIB=<path to another build of libngf_hip.so> selects an experiment build (profiles/exp_determinism_builds.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ngf_amd  # noqa
from ngf_amd import _lib, synth
if os.environ.get("NGF_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
from helpers import field_for_case, load_case

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
poison = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tile_w = int(sys.argv[3]) if len(sys.argv) > 3 else -1
ablate = int(sys.argv[4]) if len(sys.argv) > 4 else -1
only = sys.argv[5] if len(sys.argv) > 5 else ""
L = _lib.lib()
_lib.check(L.ngf_debug_set(b"ablate", ablate))
_lib.check(L.ngf_debug_set(b"poison", poison))
_lib.check(L.ngf_debug_set(b"tile_w", tile_w))
n = 203
coords = (synth.hash_uniform(79, 1, (n, 6)) * np.float32(2.2) - np.float32(1.1)).astype(np.float32)
coords[:, 2] = coords[:, 1]; coords[:, 4] = coords[:, 0]; coords[:, 5] = coords[:, 3]
dirs = synth.hash_normal(79, 2, (n, 3)).astype(np.float32)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
coords_t, dirs_t = torch.from_numpy(coords).cuda(), torch.from_numpy(dirs).cuda()
for name, kw, split in (("infoinv_r1_on", {"infoinv": True}, True), ("infoinv_r1_on", {"infoinv": True}, False),
                        ("infoinv_r1_off", {"infoinv": False}, True), ("triplane_r1_gauge", {"iteration": 30001}, True)):
    if only and only not in f"{name}/{'split' if split else 'fp32'}":
        continue
    g, params, step, mask = load_case(name)
    S = int(g["S"])
    rays = torch.from_numpy(g["rays"]).cuda()
    f = field_for_case(g, params, mask, split_bf16=split)
    mode = int(list(kw.values())[0] > 0)
    first_r = first_d = None
    bad_r, bad_d, tiles = 0, 0, {}
    t0 = time.time()
    for it in range(N):
        out = f(rays, N_samples=S, white_bg=True, **kw)
        r = torch.cat([out["rgb_map"], out["depth_map"][:, None]], 1)          # [n, 4]: colour and depth
        d = f.decode_rgb(coords_t, dirs_t, mode=mode) if name.startswith("infoinv") else None
        if first_r is None:
            first_r = r.clone(); first_d = None if d is None else d.clone()
            continue
        if not torch.equal(r, first_r):
            bad_r += 1
            rows = torch.nonzero((r != first_r).any(1)).flatten().tolist()
            tiles[tuple(rows)] = tiles.get(tuple(rows), 0) + 1
            if bad_r <= 5:
                print(f"  {name} split={split}: launch {it} render differs on rays {rows}, max abs {float((r - first_r).abs().max()):.3e}", flush=True)
                for q in rows:
                    print(f"     ray {q}: first {first_r[q].tolist()} now {r[q].tolist()} delta {(r[q] - first_r[q]).tolist()}", flush=True)
        if d is not None and not torch.equal(d, first_d):
            bad_d += 1
            rows = torch.nonzero((d != first_d).any(1)).flatten().tolist()
            if bad_d <= 5:
                print(f"  {name} split={split}: launch {it} decode_rgb differs on samples {rows}, max abs {float((d - first_d).abs().max()):.3e}", flush=True)
    torch.cuda.synchronize()
    print(f"{os.path.basename(os.path.dirname(_lib.SO_PATH))}/{name} split={split} tile_w={tile_w} poison={poison} ablate={ablate}: {N} launches, render moved {bad_r}x, decode_rgb moved {bad_d}x, {len(tiles)} distinct ray sets (top: {sorted(tiles.items(), key=lambda kv: -kv[1])[:4]}), {time.time() - t0:.0f} s", flush=True)
    f.release()
