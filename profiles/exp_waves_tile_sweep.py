#!/usr/bin/env python3
"""Experiment: waves per CU x tile width of the split march on the headline frame (R1, faithful and baked)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for bd, bc in ((0, 0), (1, 0), (1, 1)):
    for w in ("8", "12", "16"):
        for tw in ("4", "8", "16"):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--extras", "0", "--cpu-seconds", "0",
                                "--bake-density", str(bd), "--bake-color", str(bc), "--knobs", f"waves={w},tile_w={tw}"], capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(f"bake {bd}{bc} waves {w} tile_w {tw}: failed {r.stderr[-200:]}")
                continue
            d = json.loads(line[0])
            print(f"bake {bd}{bc} waves {w:>2} tile_w {tw:>2}: {d['value']:6.2f} Mray/s  {d['roofline']['kernel_ms']:.3f} ms", flush=True)
