cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5 > gpurun_out/r02t_pytest.txt
rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > gpurun_out/r02t_train.txt 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) >> gpurun_out/r02t_train.txt
rm -rf gpurun_out/ktt
timeout 200 python profiles/workload.py train_R1 50 >> gpurun_out/r02t_train.txt 2>&1
cat gpurun_out/r02t_pytest.txt; head -30 gpurun_out/r02t_train.txt; tail -2 gpurun_out/r02t_train.txt
