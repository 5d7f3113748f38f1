cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r02i_pytest.txt
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 3 2>gpurun_out/r02i_bench.err | grep '^{' > gpurun_out/r02i_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r02i_bench.json')); print(d['value'], d['roofline']['frac'], json.dumps(d['roofline']['physical'])[:600]); print({k:(v.get('Mray/s') if isinstance(v,dict) else v) for k,v in d.get('extras',{}).items()})"
tail -3 gpurun_out/r02i_bench.err
