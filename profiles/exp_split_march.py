#!/usr/bin/env python3
"""Experiment: the split march (tile_w rays x 64/tile_w consecutive steps per wave iteration) on shards of the 800x800 R1
frame.  Checks bit-identity against the one-ray-per-lane march and times every (tile_w, split) choice; the 8-GPU strong
scaling of ONE frame is bounded by the 100-row (80 000-ray) shard, the reference's renderer loop by its 4096-ray chunks."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
from ngf_amd import _lib, synth
_lib.knobs_from_env()
g, params, step = big_case("triplane", "R1")
f = field_for_case(g, params, None, device="cuda", bake=int(os.environ.get("BAKE", "0")))
out = {}
for name, rows in (("5 rows", (398, 403)), ("50 rows", (375, 425)), ("100 rows mid", (350, 450)), ("100 rows top", (0, 100)), ("200 rows", (300, 500)), ("400 rows", (200, 600)), ("frame", (0, 800))):
    rays = torch.from_numpy(synth.lookat_rays(800, 800, rows=rows)).cuda()
    for _ in range(2): r = f(rays, N_samples=192, iteration=30001)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); r = f(rays, N_samples=192, iteration=30001); b.record()
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in ev])
    print(f"  {name:14s} {rays.shape[0]:7d} rays: {ms:7.3f} ms  ({rays.shape[0] / ms / 1e3:5.1f} Mray/s)")
    out[name + "_rgb"] = r["rgb_map"].cpu().numpy(); out[name + "_depth"] = r["depth_map"].cpu().numpy()
np.savez(os.environ["OUT"], **out)
''' % (ROOT, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
ref = None
for bake in ("0", "1"):
    for tw, split in (("64", "0"), ("32", "0"), ("16", "0"), ("32", "1"), ("16", "1"), ("8", "1"), ("4", "1"), (None, None)):
        env = dict(os.environ, BAKE=bake, OUT=os.path.join(ROOT, "gpurun_out", "split_tmp.npz"))
        if tw: env.update(NGF_TILE_W=tw, NGF_SPLIT=split)
        print(f"bake={bake} tile_w={tw or 'auto'} split={split or 'auto'}")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(r.stdout, r.stderr[-400:] if r.returncode else "", end="")
        import numpy as np
        got = dict(np.load(env["OUT"]))
        if tw == "64": ref = got
        else:
            same = all(np.array_equal(ref[k], got[k]) for k in ref)
            print(f"  bit-identical to tile_w=64 unsplit: {same}")
