"""Where does a small launch lose its time?  NGF_EXP_TIMELINE build: every wave of the PRODUCTION render kernel writes wall-clock marks
(kernel entry, after the LDS image barrier, first tile, exit; 100 MHz s_memrealtime) and its tile / pass / iteration counts.
    make -C neural-gauge-fields_amd/csrc exp NAME=timeline DEFS=-DNGF_EXP_TIMELINE=1
    NGF_LIB=neural-gauge-fields_amd/csrc/build/exp/timeline/libngf_hip.so python profiles/exp_timeline.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ngf_amd  # noqa
from ngf_amd import _lib, synth
_lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
from ngf_amd.cases import big_case, field_for_case
L = _lib.lib()
model = os.environ.get("MODEL", "triplane")
g, params, step = big_case(model, "R1")
LEVEL = int(os.environ.get("LEVEL", "3"))          # 3 = the module default since round 4 (baked colour planes), 2 = round 3's
f = field_for_case(g, params, None, device="cuda", bake=True, bake_color=LEVEL >= 3)
print("level", LEVEL)
kw = dict(iteration=30001) if model == "triplane" else dict(infoinv=True)
h = f.handle()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
WAVES = 12 if model == "triplane" else 12
NW = 256 * WAVES

def launch(rays, tile_w=None):
    n = rays.shape[0]
    buf = torch.zeros(16 + 16 * NW, dtype=torch.int64, device="cuda")
    rgb, depth = torch.empty((n, 3), device="cuda"), torch.empty((n,), device="cuda")
    if tile_w: _lib.check(L.ngf_debug_set(b"tile_w", tile_w))
    for _ in range(3):
        buf.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.ngf_field_render(h, rays.data_ptr(), n, 192, 1, 1, None, rgb.data_ptr(), depth.data_ptr(), buf.data_ptr(), st))
        b.record()
        torch.cuda.synchronize()
    if tile_w: _lib.check(L.ngf_debug_set(b"tile_w", -1))
    return a.elapsed_time(b), buf[16:].view(NW, 16).cpu().numpy()

for rows, tw in (((350, 355), None), ((350, 360), None), ((350, 400), None), ((350, 450), None), ((350, 450), 8), ((300, 500), None), ((0, 800), None)):
    rays = torch.from_numpy(synth.lookat_rays(800, 800, rows=rows)).cuda()
    ms, t = launch(rays, tw)
    live = t[:, 0] > 0
    t = t[live].astype(np.float64)
    t0 = t[:, 0].min()
    us = lambda x: (x - t0) / 100.0          # 100 MHz -> microseconds
    start, ready, first, end = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3])
    worked = t[:, 4] > 0
    span = end.max()
    busy = (end - start)[worked]
    print(f"rows {rows} tile_w {tw or 'auto'}: {rays.shape[0]} rays, event {ms*1e3:.0f} us, kernel span {span:.0f} us; waves that ran {live.sum()}, that took tiles {worked.sum()}")
    print(f"   wave start: median {np.median(start):.1f} max {start.max():.1f} us;  LDS image ready - start: median {np.median(ready-start):.1f} max {(ready-start).max():.1f} us")
    print(f"   tiles per working wave: min {t[worked,4].min():.0f} median {np.median(t[worked,4]):.0f} max {t[worked,4].max():.0f};  passes/wave median {np.median(t[worked,5]):.0f} max {t[worked,5].max():.0f}; iterations/wave median {np.median(t[worked,6]):.0f}")
    q = np.percentile(end[worked], [0, 5, 25, 50, 75, 95, 100])
    print("   wave end percentiles 0/5/25/50/75/95/100 (us): " + " ".join(f"{v:.0f}" for v in q))
    print(f"   utilisation of the working waves over the span: {busy.sum() / (worked.sum() * span):.3f};  of all {NW} wave slots: {busy.sum() / (NW * span):.3f}")
    per_tile = (end - first)[worked] / t[worked, 4]
    print(f"   us per tile (per wave mean): median {np.median(per_tile):.1f}, p5 {np.percentile(per_tile,5):.1f}, p95 {np.percentile(per_tile,95):.1f}")
    # round 6: the launch's wave-time by what it was spent on -- a table that sums to the event time
    plan_r, plan_s = (C.c_int64 * 4)(), (C.c_int32 * 4)()
    nseg = L.ngf_debug_tile_plan(int(rays.shape[0]), 8 if model == "triplane" else 16, NW, -1, plan_r, plan_s) if not tw else 0
    seg_t = t[:, 8:12].sum(0) / 100.0                                   # us of wave time per plan segment
    seg_n = np.array([((t[:, 12].astype(np.int64) >> (16 * k)) & 0xffff).sum() for k in range(4)], np.float64)
    slots = float(NW)
    wave_time = span * slots
    tile_time = seg_t.sum()
    rows_out = [("launch -> first wave (event time - kernel span)", ms * 1e3 - span)]
    rows_out.append(("waves start after the launch's first wave (mean)", float(start.sum()) / slots))
    rows_out.append(("LDS image + barrier (mean per wave slot)", float((ready - start).sum()) / slots))
    if nseg and seg_n[0] > 0:
        per_ray0 = seg_t[0] / (seg_n[0] * (1 << plan_s[0]))
        rows_out.append((f"{int(seg_n[0])} tiles of {1 << plan_s[0]} rays: {seg_t[0] / seg_n[0]:.1f} us per tile = {per_ray0:.2f} us of wave time per ray", seg_t[0] / slots))
        for k in range(1, nseg):
            if seg_n[k] > 0:
                wk = 1 << plan_s[k]
                rows_out.append((f"{int(seg_n[k])} tiles of {wk} ray(s): {seg_t[k] / seg_n[k]:.1f} us per tile = {seg_t[k] / (seg_n[k] * wk):.2f} us per ray "
                                 f"({seg_t[k] / (seg_n[k] * wk) / per_ray0:.2f}x the wide tiles' price; at that price this segment would cost {seg_n[k] * wk * per_ray0 / slots:.1f} us)", seg_t[k] / slots))
    else:
        rows_out.append(("tiles", tile_time / slots))
    between = float((end - first)[worked].sum()) - tile_time
    rows_out.append(("between tiles (queue atomic, plan look-up)", between / slots))
    idle_tail = float((span - end).sum()) / slots + float((first - ready)[worked].sum()) / slots
    rows_out.append(("wave slots idle: before the first tile and after a wave's last tile until the launch's last wave ends", idle_tail))
    total = sum(v for _, v in rows_out)
    print(f"   where the launch's {ms * 1e3:.0f} us go (wave time / {NW} wave slots):")
    for name, v in rows_out:
        print(f"      {v:8.1f} us  {name}")
    print(f"      {total:8.1f} us  sum")
