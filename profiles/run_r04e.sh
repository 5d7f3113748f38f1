# Final round-4 evidence run (gpurun): the whole GPU suite (plain and under the LDS / allocation poison knob), then profiles/collect_all_r04.sh.
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r04_pytest_gpu.txt
(NGF_TEST_POISON=3 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) >> gpurun_out/r04_pytest_gpu.txt
sha256sum neural-gauge-fields_amd/csrc/libngf_hip.so | cut -c1-16 >> gpurun_out/r04_pytest_gpu.txt
bash profiles/collect_all_r04.sh > gpurun_out/r04_collect.log 2>&1
bash profiles/exp_train_pmc.sh > gpurun_out/r04_train_pmc.txt 2>&1
tail -5 gpurun_out/r04_pytest_gpu.txt; cut -c1-400 gpurun_out/r04_bench.json
