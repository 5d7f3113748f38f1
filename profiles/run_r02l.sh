cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "no_fold" 2>&1 | tail -5
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>gpurun_out/r02l_bench.err | grep '^{' > gpurun_out/r02l_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r02l_bench.json')); print(d['value']); print({k:(round(v.get('Mray/s',0),2) if isinstance(v,dict) and 'Mray/s' in v else v) for k,v in d.get('extras',{}).items()})"
tail -3 gpurun_out/r02l_bench.err
