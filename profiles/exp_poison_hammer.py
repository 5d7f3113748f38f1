"""hunt the once-in-forty failure of test_infoinv_split_bf16_keeps_fp32_accuracy with poisoned LDS / allocations.
In ONE process (state persists between launches, like in the suite): N rounds; every round runs other kernels first (trainer, UV,
fuzz cases -- varying order) and then the InfoInv parity functions, all with ngf_debug_set("poison", 3): a launch that fills every
CU's LDS with quiet NaNs precedes every kernel of the library, new handles' allocations are NaN-filled before packing.  Between rounds
big NaN-filled torch tensors are allocated and freed so that recycled device memory is dirty too.
    python profiles/exp_poison_hammer.py [rounds] > profiles/r03_poison_hammer.txt"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ngf_amd  # noqa
from ngf_amd import _lib
import test_gpu_parity as tp
import test_gpu_train as tt
import test_gpu_uv as tu
import test_gpu_fuzz as tf

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
poison = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_lib.check(_lib.lib().ngf_debug_set(b"poison", poison))
fails, t0 = 0, time.time()
for it in range(rounds):
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(1 + it % 5)]     # dirty the allocator's recycled blocks
    del junk
    try:
        if it % 4 == 0: tt.test_gradients_match_autograd_oracle(0)
        if it % 4 == 1: tu.test_uv_split_bf16_keeps_the_fp32_tolerances("uv_sphere")
        if it % 4 == 2:
            for k in (3, 7, 11): tf.test_random_configuration_matches_oracle(k, True)
        if it % 4 == 3: tu.test_uv_matches_oracle_and_reference("uv_square")
        for name in tp.INFOINV:
            tp.test_infoinv_split_bf16_keeps_fp32_accuracy(name)
            tp.test_render_matches_oracle_and_reference(name, 0)
            tp.test_decode_rgb_matches_oracle(name, False)
        if it % 7 == 0:
            for name in tp.TRIPLANE:
                tp.test_render_matches_oracle_and_reference(name, it % 4)
    except Exception:
        fails += 1
        print("ROUND", it, "FAILED", flush=True)
        traceback.print_exc(limit=6)
        sys.stdout.flush()
print(f"rounds {rounds}, poison {poison}, failures {fails}, {time.time() - t0:.0f} s, lib sha {__import__('hashlib').sha256(open(_lib.SO_PATH,'rb').read()).hexdigest()[:16]}")
