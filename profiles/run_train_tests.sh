mkdir -p gpurun_out/train
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/train/pytest_train.txt 2>&1
tail -15 gpurun_out/train/pytest_train.txt
python profiles/workload.py train_R1 20 2>&1 | grep "^train_R1"
