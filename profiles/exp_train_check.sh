# Training step: parity tests, then the per-kernel times of 20 iterations under rocprofv3 (gpurun -- bash profiles/exp_train_check.sh [tag]);
# NGF_ABLATE=524288 keeps the step on one stream (kernel times then add up to the iteration)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-cur}
timeout 240 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -4
rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > gpurun_out/train_${TAG}.txt 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) >> gpurun_out/train_${TAG}.txt
python profiles/timeline_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) > gpurun_out/train_${TAG}_timeline.txt 2>&1
rm -rf gpurun_out/ktt
head -${LINES_OUT:-24} gpurun_out/train_${TAG}.txt | cut -c1-150
