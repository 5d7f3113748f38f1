#!/usr/bin/env python3
"""Experiment: where ngf_field_create spends its time (library built with -DNGF_EXP_CREATE_TIMES: make -C neural-gauge-fields_amd/csrc exp NAME=ct
DEFS=-DNGF_EXP_CREATE_TIMES; NGF_LIB=.../build/exp/ct/libngf_hip.so).  The phases are printed by the library on stderr, the last rebuild of five."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ngf_amd  # noqa
from ngf_amd import _lib
if os.environ.get("NGF_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
from helpers import big_case, field_for_case
g, params, step = big_case("triplane", "R1")
f = field_for_case(g, params, None, bake=True, bake_color=True)
f.handle(); torch.cuda.synchronize()
for k in range(5):
    f._handle_key = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sys.stderr.write(f"--- rebuild {k}\n"); sys.stderr.flush()
    f.handle(); torch.cuda.synchronize()
    sys.stderr.write(f"total {1e3 * (time.perf_counter() - t0):.2f} ms (python side included)\n")
# the Python side of a rebuild: the key, the descriptor, the create call, the destroy of the previous handle
import ctypes as C
L = _lib.lib()
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    key = f._param_key(); t1 = time.perf_counter()
    old, f._handle, f._handle_key = f._handle, None, None
    f.handle(); torch.cuda.synchronize(); t2 = time.perf_counter()
    L.ngf_field_destroy(old); torch.cuda.synchronize(); t3 = time.perf_counter()
    sys.stderr.write(f"python: _param_key {1e6 * (t1 - t0):.0f} us, descriptor + create {1e6 * (t2 - t1):.0f} us, destroy of the old handle {1e6 * (t3 - t2):.0f} us\n")
