import os, sys, traceback
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_parity as tp
import test_gpu_train as tt
import test_gpu_uv as tu
import test_gpu_fuzz as tf
fails = 0
for it in range(40):
    try:
        # dirty the LDS / caches with other kernels in varying order
        if it % 3 == 0: tt.test_gradients_match_autograd_oracle(0)
        if it % 3 == 1: tu.test_uv_split_bf16_keeps_the_fp32_tolerances("uv_sphere")
        if it % 3 == 2:
            for k in (3, 7, 11): tf.test_random_configuration_matches_oracle(k, True)
        for name in tp.INFOINV:
            tp.test_infoinv_split_bf16_keeps_fp32_accuracy(name)
    except AssertionError as e:
        fails += 1
        print("ITER", it, "FAILED")
        traceback.print_exc(limit=3)
print("done, failures:", fails)
