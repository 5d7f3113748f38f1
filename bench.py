#!/usr/bin/env python3
"""bench.py -- Mray/s of the TriPlane ray-march hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W           (N > 1 without a launcher: starts its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one 800x800 frame (640 000 rays, 192 samples/ray, seeded
synthetic camera + weights, SURVEY.md section 8 D2, dense preset R1): at N=1 the whole frame on one GPU;
at N>1 the frame's rows are dealt out to the ranks in 10-row blocks (strong scaling of ONE frame, as
BASELINE.json's north_star asks) and every frame is followed by one RCCL all-gather of the composited pixels,
double-buffered so that frame k's exchange overlaps frame k+1's march.
Rays and parameters are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL between the ranks of one node moves memory handles as dmabuf on this driver; the runtime reads the switch when it starts (the first
# torch.cuda call), so it is set before torch is imported -- also for ranks a foreign launcher started with an environment that lacks it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The exchange is 1.3 MB per rank and rides beside the next frame's march; RCCL's kernel holds one CU per channel -- CUs the persistent render grid wants (a
# stand-in kernel that holds 16 / 32 / 64 CU slots behind every frame costs the pipelined step 0.6 / 2.5 / 4.4 %: profiles/r06_exchange_contention.txt).
# Eight channels move the 10 MB of a frame in well under a step; whoever launches the job may set another count.
os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
# The N > 1 pipeline keeps five streams busy (two render streams, RCCL's, the reorder's, the caller's); the HIP runtime maps a process's streams onto FOUR
# hardware queues unless told otherwise, and two streams that share a queue run one behind the other: beside an exchange that holds CUs one rank's pipelined
# step is 0.637 / 0.640 / 0.653 / 0.688 ms on four queues and 0.611 / 0.609 / 0.616 / 0.650 ms on eight (profiles/r06_exchange_contention.txt, sections 7-9).
# Read when the runtime starts, so it is set before torch is imported -- for the ranks of an N > 1 job only (the launcher's WORLD_SIZE is in the environment by
# then): the single-stream N = 1 frame and its extras stay on the runtime's default (measured indifferent: 148.4-148.6 Mray/s, training step 1.15 ms either way).
# The gain is not the same on every box / run (one of four runs showed none at 32 x 400 us); it was never a loss for two render streams and four or more buffers.
if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 or os.environ.get("NGF_BENCH_FORCE_DIST") == "1" or os.environ.get("NGF_BENCH_SELF_LAUNCHED") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32 matrix peak = fp32 vector peak (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz)
L2_PEAK_TBS = 34.5          # MI355X_MICROARCH.md: aggregate L2 bandwidth
PEAK_CLOCK_GHZ = 2.4        # the clock the 157.3 TFLOP/s peak is quoted at
VALU_FLOP_PER_WAVE_INST = 128.0   # one wave64 fp32 instruction at one FMA per lane: 2 SIMD cycles at the peak's 64 flop/clk/SIMD
H = W = 800
S = 192


def alg_bytes_per_ray(s_active: float, model: str) -> float:
    """every bilinear tap counted once, no cache credit."""
    if model == "triplane":
        return 24 + 16 + S * 864.0 + s_active * 2304.0
    return 24 + 16 + S * 1152.0 + s_active * 3456.0


def so_sha16():
    import hashlib
    from ngf_amd import _lib
    return hashlib.sha256(open(_lib.SO_PATH, "rb").read()).hexdigest()[:16]


PMC_ROUNDS = ("r06", "r05", "r04", "r03", "r02")


def load_pmc(tag):
    """profiles/<round>_<tag>_pmc.json of the newest round that has one (profiles/collect.sh + summarize_pmc.py, committed) or None;
    `stale` = collected with another build of the library than the one loaded now."""
    for rnd in PMC_ROUNDS:
        rel = f"profiles/{rnd}_{tag}_pmc.json"
        try:
            d = json.load(open(os.path.join(ROOT, rel)))
        except Exception:
            continue
        d["file"] = rel
        d["stale"] = d.get("so_sha16") != so_sha16()
        return d
    return None


# Necessary vector arithmetic of the level-3 formulation, in wave instructions (one instruction = one operation in each of 64 lanes); the
# itemised table is DESIGN.md section 4.11.  "Necessary" = the arithmetic the formulation (SURVEY.md Appendix A.1 with folds (i)-(iii)) prescribes
# for one lane's sample / one lane's share of a pass, counted at one instruction per fp32 operation (an FMA where the reference's op order allows
# one); addressing, masks, queue bookkeeping, lane permutations, selects of the collect loop are NOT in it.
NECESSARY_VALU = {"march_iteration": 258, "shade_pass": 321, "per_tile": 92}


NOTES = {
    "roofline.frac": "level-3 TriPlane headline: (executed fp32 MFMA flops + 128 flop x necessary wave64 VALU instructions) / kernel_ms / 157.3 TFLOP/s.  Everything in "
                     "it is measured by THIS run: the pass / evaluated-sample / tile counts come from a statistics launch of the same rays, kernel_ms is the median "
                     "of the timed loop's launches (HIP events).  Prices: the fp32 MFMA 16x16x4 occupies a SIMD for 32 cycles (2048 flop), a wave64 VALU instruction "
                     "for 2 cycles at the peak rate the 157.3 TFLOP/s figure assumes (1024 SIMDs x 64 flop/clk x 2.4 GHz); measured VALU prices are 2.7-8.3 cycles "
                     "(profiles/r05_micro_valu_cost.txt), which is why the counters' busy fraction (physical.simd_busy, ~0.93) is far above this work fraction.  "
                     "mfma_frac = the matrix flops alone.  Rounds 2-4 reported mfma_frac as frac, round 5 the counters' busy fraction.",
    "useful_op_frac": "(necessary VALU instructions x 4 cycles + executed MFMA cycles) / (issued VALU instructions x 4 + executed MFMA cycles): necessary = "
                      "NECESSARY_VALU (DESIGN.md section 4.11) x this run's march iterations (evaluated samples / 64, a lower bound: partial waves count "
                      "as fractions), shade passes and tiles; issued = SQ_INSTS_VALU of the committed PMC run minus its MFMA instructions; an fp32 MFMA "
                      "16x16x4 occupies the SIMD for 32 cycles, a wave64 VALU instruction for 4 (transcendentals 16: counted as 4 on both sides)",
    "simd_vector_datapath": "MFMA (fp32 and bf16) and VALU instructions of a SIMD execute one after the other on gfx950 (profiles/micro/"
                            "mfma_valu_overlap.hip, profiles/r02_micro_mfma_valu_overlap.txt: 8.75 ms of fp32 MFMA + 3.07 ms of VALU run together in "
                            "11.63 ms; bf16: 4.44 + 3.07 -> 7.39), so their busy cycles add and this sum is the binding roof of the kernel.  "
                            "SQ_ACTIVE_INST_VALU also counts the issue cycles of the MFMA instructions (profiles/r02_counter_semantics.txt: 4 of "
                            "every 33 fp32 / 16 bf16 MFMA cycles); they are subtracted.",
    "l1_tag_frac": "TCP_TOTAL_CACHE_ACCESSES of the committed PMC run / (256 CUs x the counters' effective clock x THIS run's kernel time): the L1 looks up one 64-byte "
                   "piece per cycle and CU; with the SIMD roof the other resource that paces the level-3 frame (DESIGN.md section 4.11, profiles/r05_level3_ablations.txt)",
    "hbm_fabric": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; includes Infinity-Cache hits",
    "algorithmic_d3": "model, not a bound (SURVEY 8 D3 accounting: every bilinear tap counted once, no cache credit; the 52 MB texture set is "
                      "L1/L2/Infinity-Cache resident, so this exceeds the HBM peak by construction)",
    "flops_counted": "EXECUTED fp32 MFMA flops (v_mfma_f32_16x16x4_f32: passes x MFMAs x 2048; equals rocprofv3 "
                     "SQ_INSTS_VALU_MFMA_MOPS_F32 x 512), layer 1 pre-composed with `basis`, per-ray view fold",
    "physical": "busy fractions are counter ratios of the committed rocprofv3 --pmc run of the same workload (file named in `source`, "
                "`pmc_stale` = collected with another build); rates divide the counters' bytes by THIS run's kernel time",
}


def physical_roofs(pmc, k_ms, share=1.0):
    """Physically bounded utilisation figures of the launch, every one <= 1 by construction, as bare numbers (prose: NOTES).  Busy
    fractions are counter ratios of the rocprofv3 run of this workload; rates divide the counters' bytes by THIS run's kernel time."""
    if pmc is None:
        return None
    t = k_ms * 1e-3
    out = {"source": pmc["file"], "pmc_stale": pmc["stale"], "kernel_ms_under_pmc": pmc.get("kernel_ms_under_pmc")}
    if "mfma_busy_frac" in pmc and "valu_busy_frac" in pmc:
        issue = pmc.get("mfma_issue_frac", 0.0)
        out.update({"mfma_busy": pmc["mfma_busy_frac"], "valu_busy_raw": pmc["valu_busy_frac"], "mfma_issue_in_valu": issue,
                    "simd_busy": pmc["mfma_busy_frac"] + max(pmc["valu_busy_frac"] - issue, 0.0)})
    if "ta_busy_frac" in pmc:
        out["ta_busy"] = pmc["ta_busy_frac"]
    if "lds_busy_frac" in pmc:
        out["lds_busy"] = pmc["lds_busy_frac"]
    if "l2_read_bytes_per_launch" in pmc:
        l2 = pmc["l2_read_bytes_per_launch"] * share / t / 1e12
        out.update({"l2_read_TBps": l2, "l2_frac": l2 / L2_PEAK_TBS, "l2_hit": pmc.get("l2_hit_frac"), "l1_hit": pmc.get("l1_hit_frac")})
    if "tcp_accesses_per_launch" in pmc and pmc.get("effective_clock_ghz"):
        # L1 (TCP) tag look-ups: one per cycle and CU, each serving a quad of lanes one 64-byte piece; 256 CUs
        out["l1_tag_frac"] = pmc["tcp_accesses_per_launch"] * share / (256.0 * pmc["effective_clock_ghz"] * 1e9 * t)
    if "hbm_traffic_bytes_per_launch" in pmc:
        hb = pmc["hbm_traffic_bytes_per_launch"] * share / t / 1e9
        out.update({"hbm_fabric_GBps": hb, "hbm_frac": hb / HBM_PEAK_GBS})
    for k in ("vgpr", "scratch_bytes_per_lane", "lds_bytes"):
        if k in pmc:
            out[k] = pmc[k]
    return out


def _r(x, nd=4):
    """Round floats for the compact line (significant digits, not decimals)."""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


LINE_LIMIT = 4096
FIXED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def compact_line(result: dict) -> str:
    """The ONE JSON line the driver parses (it keeps an ~8 KB tail of stdout): fixed keys, config, roofline (numbers only),
    cpu_baseline, parity, the multi-GPU figures -- never more than LINE_LIMIT bytes.  Everything else (extras, notes, the
    algorithmic D3 model) goes to bench_extras.json."""
    out = {k: result[k] for k in FIXED_KEYS}
    for k in ("value", "ms_per_step"):
        out[k] = _r(out[k], 6)
    out["config"] = result["config"]
    rf = result.get("roofline") or {}
    keep = ("bound", "achieved", "peak", "unit", "frac", "mfma_frac", "mfma_TFLOPs", "useful_op_frac", "traffic", "kernel", "kernel_ms", "flops_per_launch",
            "valu_insts_necessary", "clock_ghz", "active_samples_per_ray", "evaluated_samples_per_ray", "mlp_passes", "binding")
    crf = {k: _r(rf.get(k), 5) for k in keep if k in rf}
    ph = rf.get("physical")
    if ph:
        crf["physical"] = _r({k: ph[k] for k in ("simd_busy", "mfma_busy", "valu_busy_raw", "mfma_issue_in_valu", "ta_busy", "l1_tag_frac", "l2_hit", "hbm_frac",
                                               "pmc_stale", "source", "vgpr", "scratch_bytes_per_lane") if k in ph})
    out["roofline"] = crf
    if "cpu_baseline" in result:
        cb = result["cpu_baseline"]
        out["cpu_baseline"] = {"value": _r(cb["value"], 5), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:160]}
    if "parity" in result:
        out["parity"] = _r(result["parity"], 4)
    for k in ("ms_per_step_median", "launch_ms", "gathered_frame_bit_identical_to_single_gpu_render", "all_gather_ms", "all_gather_bytes_per_rank",
              "shard_kernel_ms", "critical_path_ms", "speedup_vs_cpu_port", "launch_ms_note"):
        if k in result:
            out[k] = _r(result[k], 5)
    if "extras" in result:      # headline numbers of the other configs only; the full entries are in the side file
        out["extras_Mray_s"] = {k: _r(v["Mray/s"], 4) for k, v in result["extras"].items() if isinstance(v, dict) and "Mray/s" in v}
        tr = result["extras"].get("train_step_R1", {})
        if "ms_per_iteration" in tr:
            out["train_ms_per_iteration"] = _r(tr["ms_per_iteration"], 4)
        ta = result["extras"].get("train_step_R1_autograd", {})
        if "ms_per_iteration" in ta:
            out["train_autograd_ms_per_iteration"] = _r(ta["ms_per_iteration"], 4)
        hbm = result["extras"].get("handle_build_ms", {})
        if hbm and "error" not in hbm:
            out["handle_build_ms"] = {k.replace("triplane_", ""): _r(v, 3) for k, v in hbm.items() if isinstance(v, float)}
        out["extras_file"] = "bench_extras.json"
    line = json.dumps(out, separators=(",", ":"))
    for k in ("extras_Mray_s", "parity", "speedup_vs_cpu_port"):     # never reached with today's keys; a guard, not a plan
        if len(line) <= LINE_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:
        raise RuntimeError(f"bench line is {len(line)} bytes (> {LINE_LIMIT}): the driver would not parse it")
    return line


def write_side_file(result: dict):
    """Full result (extras, notes, D3 model) next to bench.py and, when the directory exists, under gpurun_out/."""
    full = dict(result)
    full["notes"] = NOTES
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_extras.json"), "w") as fh:
                    json.dump(full, fh, indent=1)
            except OSError:
                pass


def plan_tiles(n, wide, resident):
    """Number of tiles of a render launch of n rays (ngf_debug_tile_plan: the library's own host arithmetic)."""
    import ctypes as C
    from ngf_amd import _lib
    r, sh = (C.c_int64 * 4)(), (C.c_int32 * 4)()
    k = _lib.lib().ngf_debug_tile_plan(int(n), int(wide), int(resident), -1, r, sh)
    return sum((int(r[i]) + (1 << int(sh[i])) - 1) >> int(sh[i]) for i in range(k))


def build_field(model, preset, device, bake=False, bake_color=False, no_fold=False, split_bf16=False):
    from ngf_amd.cases import big_case, field_for_case
    g, params, step = big_case(model, preset)
    f = field_for_case(g, params, None, device=device, bake=bake, bake_color=bake_color, no_fold=no_fold, split_bf16=split_bf16)
    f.handle()
    return f, g, params, step


# TEST knobs (tests/test_gpu_parity.py::test_bench_with_real_ranks_on_one_gpu): RCCL refuses two ranks on one device, gloo moves CUDA tensors through the host -- with
# NGF_BENCH_BACKEND=gloo NGF_BENCH_ONE_DEVICE=1 the N > 1 path runs as N real processes on ONE GPU (every rank its own row blocks, the exchange between processes,
# the ranks arriving at different times), which no single-rank run can show.  The line says so (config.test_backend); it is never a measurement.
BACKEND = os.environ.get("NGF_BENCH_BACKEND", "nccl")
ONE_DEVICE = os.environ.get("NGF_BENCH_ONE_DEVICE") == "1"
CLOCK_PREAMBLE_MS = 80.0          # untimed device work between the W warm-up steps and the timed region (time_steps): the GPU's clocks ramp for 25-30 ms after an idle period
PREAMBLE_REPORT = {}


def time_steps(fn, steps, warmup, device, dist_on, finish=lambda: None, marks=None, preamble=None):
    """The timed region of the contract: W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX over
    ranks.  `marks` (a list) receives one (start, end) HIP-event pair per timed step, recorded by `fn(pair)` around its render launch on
    the launch stream INSIDE the timed loop: their durations are the per-step launch times the line reports (median / min / max) and
    what roofline.kernel_ms is taken from -- the same launches the wall clock saw, not a second loop.
    `preamble`: the launch the clock preamble repeats.  It MUST NOT contain a collective: the preamble runs for a wall-clock time, so its
    trip count differs from rank to rank (at N = 1 it defaults to fn(None))."""
    import torch.distributed as dist
    # Clock preamble (round 6, profiles/r06_clock_ramp.txt): after an idle period -- the seconds this process spent building the field on the host --
    # the GPU's first 25-30 ms of work run up to 1.3x slower while its clocks ramp (every launch of a block that follows a pause, whatever the
    # launch size; the steady state is flat to 0.1 %).  W = 3 warm-up steps of an eight-rank shard are 2 ms and K = 20 timed steps 13 ms: the whole
    # timed region would sit on the ramp.  So the device is kept busy with the very launch that is measured for CLOCK_PREAMBLE_MS right in front of the
    # timed region; nothing of it is timed, and the line reports it (config.clock_preamble).
    # Order (last session of round 6): barrier, the contract's W warm-up steps, barrier, preamble, barrier + synchronize, K timed steps.  The warm-up
    # comes FIRST because at N > 1 it holds the first collectives of the job -- RCCL sets up its channels on a communicator's first exchange and first
    # barrier, host-side work of unknown length during which the GPU runs dry: with the preamble in front of the warm-up (as first written) the timed
    # region of a real multi-GPU run would have started on the ramp again.  The ranks meet at a barrier before the preamble (they arrive seconds apart),
    # run it by their own clocks for the same wall-clock time, and each leaves a few launches in its queue behind which the closing barrier's exchange
    # is enqueued, so the device stays busy while the host waits for the other ranks (they differ by less than one 4-launch batch).
    # The preamble is the shard's render launch alone -- every rank stops by its own clock, so a collective in this loop would be issued a different
    # number of times per rank and pair up with the warm-up's (or the barrier's) collectives.
    if preamble is None:
        if dist_on:
            raise ValueError("time_steps(dist_on=True) needs a collective-free `preamble` launch")
        preamble = lambda: fn(None)          # noqa: E731
    if dist_on:
        dist.barrier()
    for _ in range(warmup):
        fn(None)
    finish()
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    pre_t0, pre_n = time.perf_counter(), 0
    while (time.perf_counter() - pre_t0) * 1e3 < CLOCK_PREAMBLE_MS:
        for _ in range(4):
            preamble()
        pre_n += 4
        torch.cuda.synchronize(device)
    if dist_on:
        for _ in range(8):
            preamble()
        pre_n += 8
    PREAMBLE_REPORT.update({"launches": pre_n, "ms": (time.perf_counter() - pre_t0) * 1e3})
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        fn(pairs[k])
    finish()                 # the last frame's exchange + reorder belong to the timed region
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if marks is not None:
        marks.extend(pairs)
    return el


def kernel_ms(fn, steps, device):
    """Average duration of the render launch, HIP events on the stream it is launched on."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize(device)
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def cpu_baseline(params, g, step, rays_np, budget_s, f, kw):
    """The reference's CPU eager path (torch port, oracle/eager.py) on a bounded sample of the same frame."""
    from oracle.eager import EagerField
    e = EagerField(params, g["aabb"], step, g["near_far"], float(g["distance_scale"]), float(g["thr"]), str(g["model"]))
    chunk = 4096
    n_chunks = rays_np.shape[0] // chunk
    order = [(i * 37) % n_chunks for i in range(n_chunks)]          # spread over the frame
    # torch's intra-op pool degrades badly when oversubscribed on a many-core host (256 threads on the
    # MI355X box: >10 s per chunk vs <1 s with 16-32): pick the fastest thread count on a warm-up
    # chunk and report the count actually used.
    warm = torch.from_numpy(rays_np[:chunk])
    best = (None, 1e30)
    ncpu = os.cpu_count() or 1
    for nt in sorted({min(ncpu, k) for k in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        e.forward(warm[:512], S)
        t0 = time.perf_counter()
        e.forward(warm, S)
        dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (nt, dt)
    torch.set_num_threads(best[0])
    done, t_used, worst, worst_rel, sq, cnt, act = 0, 0.0, 0.0, 0.0, 0.0, 0, []
    for ci in order[:64]:
        r = torch.from_numpy(rays_np[ci * chunk:(ci + 1) * chunk])
        t0 = time.perf_counter()
        rgb, depth, a = e.forward(r, S)
        t_used += time.perf_counter() - t0
        act.append(a)
        done += 1
        got = f(r.to(f.device), N_samples=S, **kw)["rgb_map"].cpu()
        diff = (got - rgb).abs()
        worst = max(worst, float(diff.max()))
        worst_rel = max(worst_rel, float((diff / (rgb.abs() + 1e-6)).max()))      # rel with atol 1e-6
        sq += float((diff.double() ** 2).sum())
        cnt += diff.numel()
        if t_used >= budget_s:
            break
    mse = sq / max(cnt, 1)
    cpu_model = "?"
    try:
        cpu_model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    return {"value": done * chunk / t_used / 1e6, "unit": "Mray/s", "cores": torch.get_num_threads(), "kind": "port",
            "host": f"{cpu_model}, {os.cpu_count()} logical CPUs; thread counts tried: 8/16/32/64, fastest kept",
            "sample": f"{done} chunks x {chunk} rays of the same frame, torch-eager port of Base.forward, {t_used:.1f} s"}, \
           {"max_abs_err_vs_cpu_port": worst, "max_rel_err_vs_cpu_port": worst_rel, "psnr_vs_cpu_port_db": (200.0 if mse == 0 else -10 * np.log10(mse)),
            "cpu_active_fraction": float(np.mean(act))}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_command(n, argv, port):
    """`python bench.py --gpus N ...` started WITHOUT a launcher (no WORLD_SIZE in the environment) re-executes itself as N ranks of one
    node, one per GPU, exactly as the driver's own multi-GPU command line does; rank 0's JSON line is the job's last stdout line."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}", "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.join(ROOT, "bench.py")] + list(argv)


def self_launch(n, argv):
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not (ONE_DEVICE and have >= 1):
        raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s) (torch.cuda.device_count()); nothing was launched")
    env = dict(os.environ, NGF_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # RCCL across processes needs dmabuf IPC on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    cmd = self_launch_command(n, argv, free_port())
    print("bench.py: no launcher in the environment, starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preset", default="R1", choices=["R0", "R1", "R2"])
    ap.add_argument("--model", default="triplane", choices=["triplane", "infoinv"])
    ap.add_argument("--bake-density", type=int, default=1, help="1 = NGF_F_BAKE_DENSITY (pre-composed density planes): optimisation level 2, the "
                    "default of ngf_amd.triplane.TriPlane since round 3; 0 = level 1")
    ap.add_argument("--bake-color", type=int, default=1, help="1 = NGF_F_BAKE_COLOR (layer 1 pre-composed into 64-channel colour planes): optimisation level 3, the "
                    "default of ngf_amd.triplane.TriPlane since round 4; 0 = level 2")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="N=1 only: also time the other presets / variants (untimed region)")
    ap.add_argument("--knobs", default="", help="experiments only: comma-separated ngf_debug_set knobs, e.g. waves=12,tile_w=8,kernel=1")
    args = ap.parse_args()

    if args.gpus < 1 or (args.gpus > 1 and H % (args.gpus * 10) != 0):
        raise SystemExit(f"--gpus {args.gpus}: the frame's 800 rows are dealt out in 10-row blocks, so N must divide 80 (1, 2, 4, 5, 8, 10, 16 ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("NGF_BENCH_FORCE_LAUNCH") == "1"):
        # no launcher: be the launcher (NGF_BENCH_FORCE_LAUNCH=1 takes this route at N = 1 too, so that one GPU can test it)
        self_launch(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    if ONE_DEVICE:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit(f"LOCAL_RANK {local} but this node shows {torch.cuda.device_count()} GPU(s)")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    self_launched = os.environ.get("NGF_BENCH_SELF_LAUNCHED") == "1"
    dist_on = world > 1 or os.environ.get("NGF_BENCH_FORCE_DIST") == "1" or (self_launched and "MASTER_PORT" in os.environ)     # the env knobs exercise the RCCL path on one GPU
    if dist_on:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if BACKEND == "nccl":
            # RCCL's stream at high priority: a render launch is a persistent grid on every CU, and while frame k's exchange waits for CUs the next frame's
            # grid (the other render stream) is waiting for the same ones -- the exchange's few workgroups go first when a render workgroup leaves
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=10), pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=10))
        else:
            dist.init_process_group(BACKEND, timeout=datetime.timedelta(minutes=10))

    import ngf_amd  # noqa: F401
    from ngf_amd import dist as ndist
    from ngf_amd import synth

    if args.knobs:
        from ngf_amd import _lib
        if any(kv.split("=")[0].strip() in ("waves", "nstep", "profile", "kernel", "stage") for kv in args.knobs.split(",")):
            _lib._LIB = _lib._load(_lib.SO_PATH_EXP)          # these knobs select experiment kernels (libngf_hip_exp.so)
        for kv in args.knobs.split(","):
            k_, v_ = kv.split("=")
            _lib.check(_lib.lib().ngf_debug_set(k_.strip().encode(), int(v_)))
    model = args.model
    kw = {"iteration": 30001} if model == "triplane" else {"infoinv": True}
    f, g, params, step = build_field(model, args.preset, device, bool(args.bake_density), bool(args.bake_color))
    n_total = H * W
    # rows dealt out in blocks of ROW_BLOCK, round robin over the ranks (equal work per rank, ngf_amd/dist.py)
    ROW_BLOCK = 10
    per = n_total // world
    assert H % (world * ROW_BLOCK) == 0, "row-block sharding expects H divisible by world x 10"
    my_rows = ndist.interleaved_rows(H, world, rank, ROW_BLOCK)
    rays_np = np.concatenate([synth.lookat_rays(H, W, rows=r) for r in my_rows], 0)
    rays = torch.from_numpy(rays_np).to(device)
    n_local = rays.shape[0]
    assert n_local == per
    pipe = ndist.PipelinedGather(per, world, device, depth=ndist.PIPELINE_DEPTH) if dist_on else None          # six buffers: with frames on two render streams the exchange of frame k may finish while frame k + 2 or k + 3 is already marched (ngf_amd/dist.py)
    send, rgb_view, depth_view = ndist.shard_buffers(per, device)
    frame_no = [0]
    last_frame = [None]
    # the gathered frames in image order land in two alternating buffers (one strided copy per output and frame, no allocation in the loop)
    frame_out = [(torch.empty((n_total, 3), device=device), torch.empty((n_total,), device=device)) for _ in range(2)] if dist_on else None
    # ... copied there on a stream of their own: the render stream never waits for an exchange or a reorder (the timed region ends with a device-wide synchronise)
    side = torch.cuda.Stream(device) if dist_on else None
    # ... and consecutive frames are marched on TWO alternating render streams (round 6): on one stream frame k + 1 starts when the LAST wave of
    # frame k has ended, so every frame pays its launch's tail (wave slots idle at the ends: ~45 us of a 0.61 ms shard, profiles/r06_timeline.txt)
    # and the launch gap; on two streams the next frame's workgroups take the CUs the previous frame has already left.  One rank's shard of eight:
    # 0.658 -> 0.627 ms per step, same pixels (profiles/r06_two_streams.txt).  Launches of one handle may overlap (a queue slot per launch, 256 in flight).
    render_streams = ndist.render_streams(device) if dist_on else None

    # the frame IS an image (W rays per row; at N > 1 a rank's rows are whole image rows too): the launch may walk it in screen-space blocks
    # (ngf_field_render_image -- what ngf_amd.evalout.evaluation passes to renderer; same pixels bit for bit, tests/test_gpu_parity.py)
    kw = dict(kw, row_width=W)

    def render_only():
        f(rays, N_samples=S, white_bg=True, out=(rgb_view, depth_view), **kw)

    if dist_on:
        # every stream of the pipeline sees one launch before anything is timed: a HIP stream gets its hardware queue on its first use (milliseconds), and with
        # W = 1 the second render stream's first launch would otherwise fall into the timed region
        for s_ in list(render_streams) + [side]:
            with torch.cuda.stream(s_):
                if s_ is side:
                    frame_out[0][1][:1].zero_()
                else:
                    render_only()
        # ... and the communicator its first exchange of this size (RCCL sets up its channels on it): once on every rank, whatever W is
        dist.all_gather_into_tensor(pipe.recv[0], pipe.send[0][0])
        torch.cuda.synchronize(device)

    def step_fn(pair):
        # pair: (start, end) HIP events recorded around THIS step's render launch on the launch stream (None in the warm-up)
        if not dist_on:
            if pair: pair[0].record()
            render_only()
            if pair: pair[1].record()
            return
        # frame k: march into send buffer k % depth on render stream k % 2, start its all-gather on RCCL's stream, hand out frame k-1 (whose
        # exchange overlapped this march) in image order.  Every frame is complete when the timed region ends.
        # (the per-step event pair is not recorded here: on alternating streams a pair would span the other stream's frame as well -- the
        # shard's launch time is measured by a serial loop after the timed region)
        k = frame_no[0]
        with torch.cuda.stream(render_streams[k % len(render_streams)]):
            out_k = pipe.buffers(k)
            f(rays, N_samples=S, white_bg=True, out=out_k, **kw)
            pipe.submit(k)
        if k > 0:
            last_frame[0] = pipe.frame_in_image_order(k - 1, H, W, ROW_BLOCK, out=frame_out[(k - 1) % 2], stream=side)
        frame_no[0] = k + 1

    def finish():
        if dist_on and frame_no[0] > 0:
            last_frame[0] = pipe.frame_in_image_order(frame_no[0] - 1, H, W, ROW_BLOCK, out=frame_out[(frame_no[0] - 1) % 2], stream=side)

    marks = []
    elapsed = time_steps(step_fn, args.steps, args.warmup, device, dist_on, finish, None if dist_on else marks, preamble=render_only)
    ms_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed / 1e6
    # per-step launch durations of the TIMED loop (HIP events on the launch stream around every render launch); at N > 1 the frames of the timed
    # loop overlap on two render streams, so the shard's launch is timed afterwards: the same K launches one after the other on one stream
    if dist_on:
        pipe.drain()
        torch.cuda.synchronize(device)
        marks = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in marks:
            a.record()
            f(rays, N_samples=S, white_bg=True, out=(pipe.send[0][1], pipe.send[0][2]), **kw)
            b.record()
        torch.cuda.synchronize(device)
    launch_ms = np.array([a.elapsed_time(b) for a, b in marks], np.float64)
    launch_stats = {"min": float(launch_ms.min()), "median": float(np.median(launch_ms)), "max": float(launch_ms.max())}

    # dominant kernel: its duration = the median of the timed loop's own launches; algorithmic bytes from the measured active count
    k_ms = launch_stats["median"]
    f(rays, N_samples=S, white_bg=True, collect_stats=True, **kw)
    torch.cuda.synchronize(device)
    st = f.last_stats.cpu().numpy().astype(np.float64)
    s_active = st[1] / n_local
    bytes_launch = alg_bytes_per_ray(s_active, model) * n_local
    achieved = bytes_launch / (k_ms * 1e-3) / 1e9
    # Executed matrix work of the launch: passes x MFMAs per 16-sample pass x 2*16*16*4 flop (v_mfma_f32_16x16x4_f32);
    # agrees with rocprofv3's SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 (profiles/r01_pmc_*.txt).
    # (144 feature + 64 layer-2 MFMAs per pass, or 64 with the baked colour planes; the 16 view-input MFMAs are spent once per
    # 8-ray tile since the per-ray view fold)
    mfma_per_pass = 64 if args.bake_color else 208
    n_tiles = plan_tiles(n_local, 8, torch.cuda.get_device_properties(device).multi_processor_count * 12)      # launch_render's tile plan (csrc/ngf_field.hip)
    mfma_flops = (st[2] * mfma_per_pass + n_tiles * 16) * 2048.0 if model == "triplane" else None
    # rocprofv3 --pmc summary of THIS workload (profiles/collect.sh -> profiles/r02_<workload>_pmc.json); flagged stale when the
    # library it was collected with is not the one loaded now
    tag = f"{model}_{args.preset}" + ("_bd" if args.bake_density and not args.bake_color else "") + ("_bdc" if args.bake_color else "")
    pmc = load_pmc(tag)
    if mfma_flops is None and pmc is not None:
        # InfoInv runs its density MLP on the matrix pipe inside the march, so the executed flops are not a function of the pass
        # count alone: take SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 of the PMC run of this very workload
        mfma_flops = pmc.get("mfma_flops_per_dispatch")
    alg = {"unit": "GB/s", "achieved": achieved, "peak": HBM_PEAK_GBS, "frac_of_hbm_peak": achieved / HBM_PEAK_GBS, "bytes_per_launch": bytes_launch,
           "flops_per_launch": (S * 550.0 + s_active * 71400.0) * n_local if model == "triplane" else None}
    if mfma_flops is not None:
        tf = mfma_flops / (k_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                    "traffic": None if pmc is None else pmc.get("hbm_traffic_bytes_per_launch", 0.0) * n_local / n_total,
                    "flops_per_launch": mfma_flops}
    else:
        # no matrix work counted (e.g. InfoInv without a PMC summary): the physical HBM-side figure, or nothing -- never the model bytes
        hb = None if pmc is None or "hbm_traffic_bytes_per_launch" not in pmc else pmc["hbm_traffic_bytes_per_launch"] * n_local / n_total
        roofline = {"bound": "hbm", "achieved": None if hb is None else hb / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None if hb is None else hb / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": hb}
    roofline["physical"] = physical_roofs(pmc, k_ms, n_local / n_total)
    ph = roofline["physical"]
    if mfma_flops is not None:
        roofline.update({"mfma_frac": roofline["frac"], "mfma_TFLOPs": roofline["achieved"]})
    if model == "triplane" and args.bake_color and mfma_flops is not None:
        # What binds these launches is the SIMD's fp32 datapath: on gfx950 a SIMD executes its MFMA and its VALU instructions one after the
        # other (profiles/r02_micro_mfma_valu_overlap.txt), so `bound` names that roof.  `frac` is WORK / TIME / PEAK of THIS run (round 6,
        # VERDICT r5 item 7): work = the launch's executed fp32 matrix flops (pass and tile counts of this run's statistics launch) + its
        # NECESSARY vector arithmetic (NECESSARY_VALU x this run's march iterations, passes and tiles; one wave64 instruction = 64 lanes x one
        # FMA = 128 flop, i.e. 2 SIMD cycles at the peak rate of 64 flop/clk/SIMD -- the same price list as the matrix instruction's 32 cycles),
        # time = the median launch of the timed loop, peak = 157.3 TFLOP/s (1024 SIMDs x 64 flop/clk x 2.4 GHz).  The counters' busy fractions
        # (MFMA busy + VALU busy - MFMA issue: utilisation, not work) stay under `physical`, from the committed PMC run.
        need = (st[0] / 64.0) * NECESSARY_VALU["march_iteration"] + st[2] * NECESSARY_VALU["shade_pass"] + n_tiles * NECESSARY_VALU["per_tile"]
        work = mfma_flops + VALU_FLOP_PER_WAVE_INST * need
        tf_simd = work / (k_ms * 1e-3) / 1e12
        roofline.update({"bound": "simd", "achieved": tf_simd, "frac": tf_simd / MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                         "valu_insts_necessary": need, "clock_ghz": PEAK_CLOCK_GHZ,
                         "prices": {"mfma_16x16x4_f32_cycles": 32, "wave64_valu_cycles": 2, "flop_per_wave64_valu": VALU_FLOP_PER_WAVE_INST, "simds": 1024}})
        roofline["binding"] = ("simd fp32 datapath (MFMA + VALU share it): frac = (executed MFMA flops + 128 x necessary wave64 VALU instructions) / "
                               "kernel_ms / 157.3 TFLOP/s, all from this run; counter busy fractions under physical")
        if ph is not None and pmc.get("valu_insts_per_launch"):
            mfma_cyc = mfma_flops / 64.0                                   # 64 fp32 flops per SIMD cycle
            issued = (pmc["valu_insts_per_launch"] - pmc.get("mfma_flops_per_dispatch", 0.0) / 2048.0) * n_local / n_total
            roofline["useful_op_frac"] = (4.0 * need + mfma_cyc) / (4.0 * issued + mfma_cyc)
            roofline["valu_insts_issued"] = issued
    # SURVEY 8 D3's model against both peaks (both exceed 1 at level 3 -- the kernel does not do D3's work: folds (i)-(iii)
    # and exact early termination remove it, with green parity -- so D3 is a model, not a roof of this kernel)
    alg["frac_of_mfma_peak"] = None if alg["flops_per_launch"] is None else alg["flops_per_launch"] / (k_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF
    alg["note"] = "both fractions exceed 1 at levels 1-3 by construction: cache-resident taps counted as HBM bytes, pre-composed layers counted as flops"
    roofline.update({"kernel": "ngf::render_kernel", "kernel_ms": k_ms, "active_samples_per_ray": s_active,
                     "evaluated_samples_per_ray": st[0] / n_local,      # in-box samples the march evaluated (exact early termination skips the rest)
                      "mlp_passes": st[2], "algorithmic_d3": alg})

    result = {
        "metric": "Mray/sec (800x800 lego-style frame, 192 samples/ray)", "value": value, "unit": "Mray/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{'TriPlane' if model == 'triplane' else 'InfoInv'} 800x800 frame, S=192, preset {args.preset} "
                               f"(seeded random planes 256^2, dense density preset), gauge on, white_bg",
                   "rays_per_step": n_total, "samples_per_ray": S, "sharding": f"rays x{world} (10-row blocks, round robin) + double-buffered RCCL all_gather" + (", frames on two alternating render streams" if dist_on else ""),
                   "level": ("3 (module default: SURVEY 7 folds (i)-(iii) -- layer 1 o basis folded into 64-channel pre-activation planes, density_decoder into "
                             "1-channel planes, per-ray view fold)" if args.bake_color else "2 (layer 1 o basis, per-ray view fold, density_decoder folded into "
                             "1-channel planes)" if args.bake_density else "1 (layer 1 o basis, per-ray view fold)") if model == "triplane" else "default",
                   "bake_density": int(args.bake_density), "bake_color": int(args.bake_color)},
        "roofline": roofline,
        # per-step durations of the render launch inside the timed loop (HIP events): SURVEY 8 D1 asks for the median of >= 20 frames
        "ms_per_step_median": launch_stats["median"], "launch_ms": launch_stats,
        "value_from_median_launch": n_local / launch_stats["median"] / 1e3,
    }
    result["config"]["clock_preamble"] = {"launches": PREAMBLE_REPORT.get("launches"), "ms": _r(PREAMBLE_REPORT.get("ms", 0.0), 3),
                                          "what": "untimed render launches between the W warm-up steps and the timed region (GPU clocks ramp 25-30 ms after idle)"}
    if dist_on:
        result["config"]["rccl_max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS")
        result["config"]["hip_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default)")
    if BACKEND != "nccl" or ONE_DEVICE:
        result["config"]["test_backend"] = f"{BACKEND}, {'all ranks on one GPU' if ONE_DEVICE else 'one GPU per rank'}: a test of the N > 1 path, not a measurement"
    if dist_on:
        result["launch_ms_note"] = ("N > 1: the timed loop's frames overlap on two render streams; launch_ms / roofline.kernel_ms are the same K shard launches "
                                    "run one after the other on one stream after the timed region")
    if args.knobs:
        result["config"]["knobs"] = args.knobs
    result["config"]["launcher"] = "self (bench.py started its own torch.distributed.run)" if self_launched else ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "none (single process)")

    if dist_on:
        # untimed: the gathered, re-ordered frame of the last step against a direct render of the whole frame on this rank
        whole = torch.from_numpy(synth.lookat_rays(H, W)).to(device)
        ref = f(whole, N_samples=S, white_bg=True, **kw)
        same = bool(torch.equal(ref["rgb_map"], last_frame[0][0]) and torch.equal(ref["depth_map"], last_frame[0][1]))
        flag = torch.tensor([int(same)], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        result["gathered_frame_bit_identical_to_single_gpu_render"] = bool(flag.item())
        # untimed: the exchange alone (SURVEY 8 E1 asks for it next to the Mray/s): blocking all-gathers, HIP events
        pipe.drain()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record()
            dist.all_gather_into_tensor(pipe.recv[0], pipe.send[0][0])
            b.record()
        torch.cuda.synchronize(device)
        result["all_gather_ms"] = float(np.median([a.elapsed_time(b) for a, b in ev]))
        result["all_gather_bytes_per_rank"] = 4 * per * 4
        # every rank's shard launch (median of ITS timed steps): min / max over the ranks -- strong scaling pays for the slowest
        mine = torch.tensor([k_ms], device=device, dtype=torch.float64)
        allk = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        per_rank = [float(t.item()) for t in allk]
        result["shard_kernel_ms"] = {"min": min(per_rank), "max": max(per_rank), "rank0": k_ms}
        # untimed: one frame with nothing overlapped -- render, exchange, reorder one after the other on this rank -- so that a
        # scaling curve can be read: the pipelined step costs max(render, exchange) + reorder in the steady state, this is their sum
        cp = []
        for _ in range(5):
            e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            torch.cuda.synchronize(device)
            dist.barrier()
            e0.record()
            f(rays, N_samples=S, white_bg=True, out=(pipe.send[0][1], pipe.send[0][2]), **kw)
            e1.record()
            dist.all_gather_into_tensor(pipe.recv[0], pipe.send[0][0])
            e2.record()
            pipe.frame_in_image_order(0, H, W, ROW_BLOCK, out=frame_out[0])
            e3.record()
            torch.cuda.synchronize(device)
            cp.append((e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)))
        cpm = torch.tensor(np.median(np.array(cp), 0), device=device, dtype=torch.float64)
        dist.all_reduce(cpm, op=dist.ReduceOp.MAX)
        result["critical_path_ms"] = {"render": float(cpm[0]), "all_gather": float(cpm[1]), "reorder": float(cpm[2]),
                                      "sum": float(cpm.sum()), "note": "unpipelined, max over ranks"}

    if world == 1 and rank == 0:
        if args.cpu_seconds > 0:
            cb, par = cpu_baseline(params, g, step, rays_np, args.cpu_seconds, f, kw)
            result["cpu_baseline"] = cb
            result["parity"] = par
            result["speedup_vs_cpu_port"] = value / cb["value"]
        if args.extras:
            extras = {}
            # the other presets and the opt-in formulations: (model, preset, field flags, result tag, PMC tag)
            BD = {"bake": True, "bake_color": True}        # level 3 = the module default
            variants = (("triplane", "R0", BD, "", "_bdc"), ("triplane", "R2", BD, "", "_bdc"),
                        ("triplane", args.preset, {"bake": True}, "_level2", "_bd"),                        # level 2: layer 1 on the matrix pipe (round 3's default)
                        ("triplane", args.preset, {}, "_level1_no_bake", ""),                                # level 1: density_decoder on 16-channel taps
                        ("triplane", args.preset, {"no_fold": True}, "_level0_no_fold", "_nofold"),        # rgb_decoder as written: what the folds buy
                        ("triplane", args.preset, {"bake": True, "split_bf16": True}, "_level2_split_bf16", "_splitd"),   # colour MLP on bf16 MFMA, 3-term split operands
                        ("triplane", args.preset, {"split_bf16": True}, "_level1_split_bf16", "_split"),
                        ("triplane", "R2", {"bake": True, "split_bf16": True}, "_level2_split_bf16", "_splitd"),
                        ("triplane", args.preset, {"bake": True, "bake_color": True, "split_bf16": True}, "_level3_split_bf16", "_bdcs"),      # round 5: level 3, layer 2 as split bf16 products
                        ("triplane", "R2", {"bake": True, "bake_color": True, "split_bf16": True}, "_level3_split_bf16", "_bdcs"),
                        ("triplane", "R0", {}, "_level1_no_bake", ""),
                        ("triplane", "R2", {"bake": True}, "_level2", "_bd"),
                        ("infoinv", "R1", {}, "", ""),
                        ("infoinv", "R1", {"split_bf16": True}, "_split_bf16", "_split"))           # rgb_decoder + density MLP on bf16 MFMA, four lanes per sample
            for mdl, preset, flags, tag, ptag_sfx in variants:
                try:
                    fx, _, _, _ = build_field(mdl, preset, device, **flags)
                    kx = dict({"iteration": 30001} if mdl == "triplane" else {"infoinv": True}, row_width=W)
                    ms = kernel_ms(lambda: fx(rays, N_samples=S, white_bg=True, **kx), 5, device)      # includes 1st-call warm-up
                    ms = kernel_ms(lambda: fx(rays, N_samples=S, white_bg=True, **kx), 10, device)
                    fx(rays, N_samples=S, collect_stats=True, **kx)
                    sx = fx.last_stats.cpu().numpy().astype(np.float64)
                    sa = sx[1] / n_total
                    px = load_pmc(f"{mdl}_{preset}{ptag_sfx}")
                    entry = {"Mray/s": n_total / ms / 1e3, "kernel_ms": ms, "active_samples_per_ray": sa}
                    if flags.get("split_bf16") and mdl == "infoinv":
                        entry["note"] = "rgb_decoder and density MLP as 3-term split bf16 products on v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16"
                    elif flags.get("split_bf16"):
                        bf = sx[2] * (48 if flags.get("bake_color") else 168) * 2 * 16 * 16 * 32                    # executed bf16 MFMA flops: 168 (level 3: 48) v_mfma_f32_16x16x32_bf16 per pass
                        entry.update({"executed_bf16_mfma_TFLOPs": bf / (ms * 1e-3) / 1e12, "bf16_mfma_frac_of_2500": bf / (ms * 1e-3) / 1e12 / 2500.0})
                    else:
                        if mdl == "triplane":
                            per_pass = 548 if flags.get("no_fold") else (64 if flags.get("bake_color") else 208)
                            fl = sx[2] * per_pass * 2048.0 + (0 if flags.get("no_fold") else plan_tiles(n_total, 8, torch.cuda.get_device_properties(device).multi_processor_count * 12) * 16 * 2048.0)
                        else:
                            fl = None if px is None else px.get("mfma_flops_per_dispatch")
                        entry.update({"executed_mfma_TFLOPs": None if fl is None else fl / (ms * 1e-3) / 1e12,
                                      "mfma_frac_of_157.3": None if fl is None else fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF})
                    entry["physical"] = physical_roofs(px, ms)
                    entry["algorithmic_d3_GBps (model, not a bound)"] = alg_bytes_per_ray(sa, mdl) * n_total / (ms * 1e-3) / 1e9
                    extras[f"{mdl}_{preset}{tag}"] = entry
                    fx.release()
                except Exception as ex:  # an extra must never take the headline number down with it
                    extras[f"{mdl}_{preset}"] = {"error": repr(ex)}
            # The reference's own evaluation shape: renderer(rays, field, chunk=4096, N_samples=-1) = 884 steps (TriPlane/main.py:94,
            # FieldBase.py:71-72) through an alpha mask like every trained model carries (FieldBase.py:261-267) -- `mask`: the one the repo's own
            # updateAlphaMask((256,)*3) builds from the seeded field (main.py:330); `ball`: occupancy = a ball of radius 0.8, 15 % of the box (an
            # object in empty space, as a trained lego is); `lattice`: thin walls every 32 cells inside a ball (clutter: the finer block image's case).  Module default level, whole frame in one launch.
            from ngf_amd.fieldbase import renderer as _renderer
            for mdl, preset, shape in (("triplane", "R1", "mask"), ("triplane", "R2", "mask"), ("triplane", "R1", "ball"), ("triplane", "R1", "lattice"), ("infoinv", "R1", "mask"), ("infoinv", "R1", "ball")):
                key = f"{mdl}_{preset}_S884_{shape}"
                tri = mdl == "triplane"
                fkw = {"iteration": 30001} if tri else {"infoinv": True}
                try:
                    fx, gx, _, _ = build_field(mdl, preset, device, True, True) if tri else build_field(mdl, preset, device)
                    if shape == "mask":
                        fx.updateAlphaMask((256, 256, 256), **({} if tri else {"infoinv": True}))
                    elif shape == "lattice":          # cluttered occupancy: thin walls every 32 cells along all three axes inside a ball of radius 1.1 (profiles/workload.py _S884lattice)
                        from ngf_amd import triplane as _tp
                        ax = torch.linspace(-1.5, 1.5, 256)
                        zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
                        wall = (torch.arange(256) % 32) < 2
                        vol = (wall[:, None, None] | wall[None, :, None] | wall[None, None, :]) & ((xx ** 2 + yy ** 2 + zz ** 2) < 1.1 ** 2)
                        fx.alphaMask = _tp.AlphaGridMask(device, torch.tensor(np.asarray(gx["aabb"], np.float32)), vol.float().to(device))
                        fx.invalidate()
                    else:
                        from ngf_amd import triplane as _tp
                        ax = torch.linspace(-1.5, 1.5, 128)
                        zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
                        fx.alphaMask = _tp.AlphaGridMask(device, torch.tensor(np.asarray(gx["aabb"], np.float32)), ((xx ** 2 + yy ** 2 + zz ** 2) < 0.64).float().to(device))
                        fx.invalidate()
                    Sx = int(fx.nSamples)
                    go = lambda: _renderer(rays, fx, chunk=4096, N_samples=-1, white_bg=True, device=device, row_width=W, **({} if tri else fkw))
                    kernel_ms(go, 3, device)
                    ms = kernel_ms(go, 8, device)
                    fx(rays, N_samples=-1, collect_stats=True, **fkw)
                    sx = fx.last_stats.cpu().numpy().astype(np.float64)
                    pmx = load_pmc(f"triplane_{preset}_bdc_S884{shape}" if tri else f"infoinv_{preset}__S884{shape}")
                    if tri:
                        fl = sx[2] * 64 * 2048.0 + plan_tiles(n_total, 8, torch.cuda.get_device_properties(device).multi_processor_count * 12) * 16 * 2048.0
                    else:
                        fl = None if pmx is None else pmx.get("mfma_flops_per_dispatch")
                    extras[key] = {"Mray/s": n_total / ms / 1e3, "kernel_ms": ms, "samples_per_ray": Sx, "mask_occupancy": float(fx.alphaMask.alpha_volume.mean()),
                                   "evaluated_samples_per_ray": sx[0] / n_total, "active_samples_per_ray": sx[1] / n_total,
                                   "executed_mfma_TFLOPs": None if fl is None else fl / (ms * 1e-3) / 1e12,
                                   "mfma_frac_of_157.3": None if fl is None else fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                                   "physical": physical_roofs(pmx, ms)}
                    fx.release()
                except Exception as ex:
                    extras[key] = {"error": repr(ex)}
            # The reference's renderer is handed HOST rays and its drivers read the pixels on the host (TriPlane/main.py:60-71,93-106: `rays.to(device)`
            # per chunk, `.cpu()` of rgb / depth).  The headline `value` starts with the rays in HBM (the contract of the metric); this is the same
            # frame with the 15.4 MB H2D copy of the ray list and the 10.2 MB D2H of rgb + depth inside the clock -- pageable tensors (what
            # torch.from_numpy gives the reference's loop) and pinned ones -- and with the rays made on the device instead (N1: ngf_generate_rays).
            try:
                hbnd = {}
                rays_pageable = torch.from_numpy(rays_np)
                rays_pinned = rays_pageable.pin_memory()
                out_pinned = (torch.empty((n_total, 3)).pin_memory(), torch.empty((n_total,)).pin_memory())
                def frame_from_host(src, pinned_out):
                    rgb_d, dep_d = _renderer(src.to(device, non_blocking=True), f, chunk=4096, N_samples=S, white_bg=True, device=device, row_width=W)
                    if pinned_out:
                        out_pinned[0].copy_(rgb_d, non_blocking=True); out_pinned[1].copy_(dep_d, non_blocking=True)
                        torch.cuda.synchronize(device)
                    else:
                        rgb_d.cpu(); dep_d.cpu()
                for label, src, po in (("pageable", rays_pageable, False), ("pinned", rays_pinned, True)):
                    frame_from_host(src, po)
                    ts = []
                    for _ in range(10):
                        torch.cuda.synchronize(device)
                        t0 = time.perf_counter()
                        frame_from_host(src, po)
                        ts.append((time.perf_counter() - t0) * 1e3)
                    hbnd[label] = {"ms_per_frame": float(np.median(ts)), "Mray/s": n_total / float(np.median(ts)) / 1e3}
                hbnd["bytes"] = {"h2d_rays": int(rays_np.nbytes), "d2h_rgb_depth": int(n_total * 16)}
                hbnd["note"] = "wall clock of one frame: H2D of the ray list + render + D2H of rgb and depth; never `value` (inputs resident in HBM there)"
                extras["host_boundary_frame"] = hbnd
            except Exception as ex:
                extras["host_boundary_frame"] = {"error": repr(ex)}
            # the screen-space tile order against the list's own (row-major) order, alternating launches of one field
            try:
                ab = {}
                for mdl, preset, shape in (("triplane", "R1", ""), ("triplane", "R2", ""), ("triplane", "R2", "S884_mask")):
                    fx, _, _, _ = build_field(mdl, preset, device, True, True)
                    ns = S
                    if shape:
                        fx.updateAlphaMask((256, 256, 256))
                        ns = -1
                    t = {0: [], W: []}
                    for rep in range(6):
                        for rw in (0, W):
                            t[rw].append(kernel_ms(lambda: fx(rays, N_samples=ns, white_bg=True, iteration=30001, row_width=rw), 2, device))
                    ab[f"{mdl}_{preset}{'_' + shape if shape else ''}"] = {"row_major_ms": float(np.median(t[0][1:])), "image_blocks_ms": float(np.median(t[W][1:]))}
                    fx.release()
                extras["tile_order_ab"] = ab
            except Exception as ex:
                extras["tile_order_ab"] = {"error": repr(ex)}
            # What a parameter change costs before the next render: ngf_field_create = texture packing, the pre-compositions of
            # the level (fp64 folds), the MLP image; timed from invalidate() to the handle being ready, median of 5 (the render itself excluded).
            try:
                hb = {}
                for label, mdl, flags in (("triplane_level0", "triplane", {"no_fold": True}), ("triplane_level1", "triplane", {}),
                                          ("triplane_level2", "triplane", {"bake": True}), ("triplane_level3", "triplane", {"bake": True, "bake_color": True}),
                                          ("infoinv", "infoinv", {}), ("infoinv_split_bf16", "infoinv", {"split_bf16": True})):
                    fx, _, _, _ = build_field(mdl, "R1", device, **flags)
                    ts = []
                    for _ in range(5):
                        fx.invalidate()
                        torch.cuda.synchronize(device)
                        t0 = time.perf_counter()
                        fx.handle()
                        torch.cuda.synchronize(device)
                        ts.append((time.perf_counter() - t0) * 1e3)
                    hb[label] = float(np.median(ts))
                    fx.release()
                extras["handle_build_ms"] = hb
            except Exception as ex:
                extras["handle_build_ms"] = {"error": repr(ex)}
            try:    # BASELINE config 4: UV-Mapping (NeuTex) colour path, DTU camera 0 (800x600), 64 samples/ray, sphere gauge
                from ngf_amd import rays as nrays
                from ngf_amd import uvmapping
                up = synth.uvmapping_params(5, "sphere")
                v0 = synth.DTU_VIEW0              # camera 0 of the DTU scan the reference ships; rays made on the device (ngf_generate_rays_dtu)
                dirs_t = nrays.generate_rays_dtu(600, 800, v0["focal"], v0["princpt"], v0["rot"], rows=(252, 348), device=device)[None]   # 96 rows through the object: 76 800 rays
                cam_t = torch.tensor(v0["campos"], dtype=torch.float32)[None]
                n_uv = dirs_t.shape[1]
                Uj = torch.rand((1, n_uv, 64), device=device)
                for split in (False, True):
                    net = uvmapping.NeuTex(primitive_type="sphere", sample_num=64, device=device, split_bf16=split)
                    net.load_params(up)
                    for _ in range(2):          # two warm calls (the first packs the weights), then the median of five launches of 40-70 ms
                        net(cam_t, dirs_t, None, jitter_u=Uj)
                    try:                        # the UV handle's (re)build: ngf_uv_create = 29 layers packed / split
                        tb = []
                        for _ in range(3):
                            net.load_params(up)
                            torch.cuda.synchronize(device)
                            t0 = time.perf_counter()
                            net.handle()
                            torch.cuda.synchronize(device)
                            tb.append((time.perf_counter() - t0) * 1e3)
                        extras.setdefault("handle_build_ms", {})["uvmapping" + ("_split_bf16" if split else "")] = float(np.median(tb))
                    except Exception as ex:
                        extras.setdefault("handle_build_ms", {})["uvmapping_error"] = repr(ex)
                    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
                    for ea, eb in ev:
                        ea.record(); net(cam_t, dirs_t, None, jitter_u=Uj); eb.record()
                    torch.cuda.synchronize(device)
                    ms = float(np.median([ea.elapsed_time(eb) for ea, eb in ev]))
                    net(cam_t, dirs_t, None, jitter_u=Uj, collect_stats=True)
                    us = net.last_stats.cpu().numpy().astype(np.float64)
                    flops = us[1] * 16 * 2 * 1334592.0                                    # passes x 16 samples x 2 x MAC/sample
                    e = {"Mray/s": n_uv / ms / 1e3, "kernel_ms": ms, "rays": int(n_uv), "in_cube_samples_per_ray": us[0] / n_uv,
                         "algorithmic_TFLOPs_all_64_samples (model)": n_uv * 64 * 2 * 1334592.0 / (ms * 1e-3) / 1e12}
                    if split:
                        e.update({"fp32_equivalent_TFLOPs": flops / (ms * 1e-3) / 1e12, "physical": physical_roofs(load_pmc("uv_sphere_split"), ms)})
                    else:
                        e.update({"executed_TFLOPs": flops / (ms * 1e-3) / 1e12, "mfma_frac_of_157.3": flops / (ms * 1e-3) / 157.3e12,
                                  "physical": physical_roofs(load_pmc("uv_sphere"), ms)})
                    extras["uvmapping_sphere" + ("_split_bf16" if split else "")] = e
                    net.release()
            except Exception as ex:
                extras["uvmapping_sphere"] = {"error": repr(ex)}
            # eval output stage on the headline frame: device (HIP events) next to the reference's host way
            # (D2H of the float frame + numpy/scipy: oracle/evalout.py), one frame each
            try:
                from ngf_amd import evalout
                from oracle import evalout as ev_orc
                r = f(rays, N_samples=S, white_bg=True, **kw)
                gt = (r["rgb_map"] * 0.95 + 0.02).clamp(0, 1)
                run = lambda: evalout.frame_outputs(r["rgb_map"], r["depth_map"], H, W, (2.0, 6.0), gt)
                run()
                dev_ms = kernel_ms(run, 5, device)
                t0 = time.perf_counter()
                rgb_h = r["rgb_map"].clamp(0, 1).reshape(H, W, 3).cpu().numpy()
                dep_h = r["depth_map"].reshape(H, W).cpu().numpy()
                gt_h = gt.reshape(H, W, 3).cpu().numpy()
                ev_orc.depth_index(dep_h, (2.0, 6.0)); ev_orc.psnr(rgb_h, gt_h); s_host = ev_orc.rgb_ssim(rgb_h, gt_h, 1); ev_orc.frame_u8(rgb_h)
                host_ms = (time.perf_counter() - t0) * 1e3
                extras["eval_output_stage_800x800"] = {"device_ms": dev_ms, "host_numpy_scipy_ms": host_ms,
                                                       "ssim_device_minus_host": run()["ssim"] - s_host}
            except Exception as ex:
                extras["eval_output_stage_800x800"] = {"error": repr(ex)}
            # one training iteration at the reference's batch shape: 4096 random rays of the frame, the model's own
            # nSamples, gauge on; the CPU leg is ONE forward+backward through autograd of the eager port (oracle/train.py)
            try:
                from ngf_amd import train as ntrain
                ft, gt_, pt_, st_ = build_field("triplane", args.preset, device, True, False)
                Str = int(ft.nSamples)
                pick = (synth.hash_uniform(9, 1, (4096,)) * np.float32(rays_np.shape[0])).astype(np.int64)
                tr_rays = torch.from_numpy(rays_np[pick]).to(device)
                tr_rgb = torch.from_numpy(synth.hash_uniform(9, 2, (4096, 3))).to(device)
                # the Trainer's default (fail-safe: the active count is read on the host, one synchronisation per step; a longer list is cut into
                # chunks) -- what the headline "ms_per_iteration" is --, then the same iterations with speculative=True (opt-in: no host round trip)
                # on a field of its own, and the reference's loop as written on the differentiable forward + torch.optim.Adam
                trn = ntrain.Trainer(ft, batch_size=4096, max_samples=Str)
                for it in range(3):
                    trn.step(tr_rays, tr_rgb, it, N_samples=Str)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for it in range(10):
                    trn.step(tr_rays, tr_rgb, 3 + it, N_samples=Str)
                torch.cuda.synchronize(device)
                it_ms = (time.perf_counter() - t0) / 10 * 1e3
                tr_extra = {"ms_per_iteration": it_ms, "iterations_per_s": 1e3 / it_ms, "batch_rays": 4096, "samples_per_ray": Str,
                            "active_samples": trn.last_active, "scratch_GiB": trn.scratch_bytes() / 2 ** 30,
                            "mode": "Trainer default: activation rows for a third of the pairs, active count read on the host (1 sync / step), nothing ever skipped"}
                try:
                    fs_, _, _, _ = build_field("triplane", args.preset, device, True, False)
                    trs = ntrain.Trainer(fs_, batch_size=4096, max_samples=Str, speculative=True)
                    for it in range(3):
                        trs.step(tr_rays, tr_rgb, it, N_samples=Str)
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    for it in range(10):
                        trs.step(tr_rays, tr_rgb, 3 + it, N_samples=Str)
                    torch.cuda.synchronize(device)
                    tr_extra["speculative_rows_ms_per_iteration"] = (time.perf_counter() - t0) / 10 * 1e3
                    tr_extra["speculative_rows_overflows"] = trs.overflows()[0]
                    trs.release()
                    fs_.release()
                except Exception as ex:
                    tr_extra["speculative_rows_ms_per_iteration"] = repr(ex)
                # TriPlane/main.py:272-299 unchanged on the drop-in field: differentiable forward, torch's MSE + density_L1, total_loss.backward(), and
                # optimizer.step() of (a) torch.optim.Adam, as the reference writes it, (b) ngf_amd.optim.Adam -- the same constructor and state, one
                # fused C-ABI call per step that also keeps the engine's packed planes current
                for key_, cls_ in (("reference_loop_on_autograd_path_ms_per_iteration", "torch"), ("reference_loop_with_ngf_optim_ms_per_iteration", "ngf")):
                    try:
                        from ngf_amd import optim as noptim
                        fa_, _, _, _ = build_field("triplane", args.preset, device, True, False)
                        opt_ = (torch.optim.Adam if cls_ == "torch" else noptim.Adam)(fa_.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
                        def ref_iter(it):
                            out_ = fa_(tr_rays, is_train=True, white_bg=True, N_samples=Str, iteration=it)
                            loss_ = torch.mean((out_["rgb_map"] - tr_rgb) ** 2)
                            tot_ = loss_ + 8e-5 * fa_.density_L1()
                            opt_.zero_grad()
                            tot_.backward()
                            opt_.step()
                            return loss_
                        for it in range(3):
                            ref_iter(it)
                        torch.cuda.synchronize(device)
                        t0 = time.perf_counter()
                        for it in range(10):
                            ref_iter(3 + it).item()                      # the reference reads the loss every iteration (main.py:297)
                        torch.cuda.synchronize(device)
                        tr_extra[key_] = (time.perf_counter() - t0) / 10 * 1e3
                        fa_._grad_engine.release()
                        fa_.release()
                    except Exception as ex:
                        tr_extra[key_] = repr(ex)
                # A/B of the trainer's streams (it forks onto two streams of its own after the colour backward: DESIGN.md section 8 N3): ten
                # more iterations, alternately forked and with the whole step on the caller's stream -- the active count falls as the field
                # trains, so the two are interleaved; every step is synchronised here, which the headline figure above is not
                try:
                    from ngf_amd import _lib as _lb
                    ab = {0: [], 1: []}
                    for it in range(20):
                        one = it & 1
                        with _lb.knobs(ablate=(1 << 19) if one else 0):
                            torch.cuda.synchronize(device)
                            t0 = time.perf_counter()
                            trn.step(tr_rays, tr_rgb, 13 + it, N_samples=Str)
                            torch.cuda.synchronize(device)
                            ab[one].append((time.perf_counter() - t0) * 1e3)
                    tr_extra["streams_ab"] = {"forked_ms": float(np.median(ab[0])), "one_stream_ms": float(np.median(ab[1])),
                                              "note": "interleaved, synchronised per step; active samples at the end: %d" % trn.last_active}
                except Exception as ex:
                    tr_extra["streams_ab"] = repr(ex)
                # what bounds the backward: global float atomics cost one transaction per (instruction, 64-byte line), 21 G/s for the whole
                # device (profiles/micro/atomic_cost.hip -> profiles/r02_micro_atomic_cost.txt); one more iteration counts them
                try:
                    import ctypes as C
                    from ngf_amd import _lib
                    Lb = _lib.lib()
                    Lb.ngf_train_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
                    cnt = (C.c_uint64 * 16)()
                    with _lib.knobs(ablate=1 << 21):
                        _lib.check(Lb.ngf_train_debug_sections(trn._h, cnt))
                        trn.step(tr_rays, tr_rgb, 34, N_samples=Str)
                        _lib.check(Lb.ngf_train_debug_sections(trn._h, cnt))
                    n_d, n_c = int(cnt[8]), int(cnt[9])
                    floor_ms = (n_d + n_c) / 21.0e9 * 1e3
                    tr_extra["atomic_roof"] = {"line_transactions": {"density_gauge_backward": n_d, "colour_backward": n_c},
                                               "peak_G_transactions_per_s": 21.0, "peak_source": "profiles/r02_micro_atomic_cost.txt",
                                               "floor_ms": floor_ms, "floor_frac_of_iteration": floor_ms / it_ms}
                except Exception as ex:
                    tr_extra["atomic_roof"] = {"error": repr(ex)}
                # the step's roofs next to the measured time: matrix work, HBM traffic of the activation rows / per-pair buffers / Adam,
                # and the atomic floor above -- three DIFFERENT units of the device; their sum is a floor of a step that overlapped nothing
                try:
                    na = float(tr_extra["active_samples"])
                    mac = (144 + 16) * 64 + 64 * 64 + 64 * 3                                   # rgb_decoder with layer 1 o basis: MAC per active sample and pass
                    flops = 3 * 2 * mac * na + 2 * 2 * 64 * 144 * 144                          # forward, data gradients, weight gradients (+ the fold / unfold of basis)
                    row_b = 4.0 * ((144 + 16 + 64 + 64) * 2 + (64 + 64 + 16 + 144) * 2 + 144 + 18)   # rows: written once, read by backward + GEMMs; dF read by the scatter
                    pair_b = 4.0 * 17 * 2 * 4096 * Str                                         # per-(ray, sample) buffers: written and read once
                    adam_b = 4.0 * sum(p.numel() for p in ft.parameters()) * 7                 # p, m, v read + written, gradient read (+ the packed copy)
                    hbm_b = row_b * na + pair_b + adam_b
                    fl_ms, hb_ms = flops / (MFMA_F32_PEAK_TF * 1e12) * 1e3, hbm_b / (HBM_PEAK_GBS * 1e9) * 1e3
                    at_ms = tr_extra.get("atomic_roof", {}).get("floor_ms", 0.0) or 0.0
                    tr_extra["roofline"] = {"mfma_flops": flops, "mfma_floor_ms": fl_ms, "hbm_bytes_model": hbm_b, "hbm_floor_ms": hb_ms, "atomic_floor_ms": at_ms,
                                            "sum_of_floors_ms": fl_ms + hb_ms + at_ms, "frac_of_iteration": (fl_ms + hb_ms + at_ms) / it_ms,
                                            "note": "fp32 MFMA peak 157.3 TF, HBM 8 TB/s, 21 G atomic line transactions/s; the step is bound by the dependent "
                                                    "chains of its scatter / colour kernels, not by any of the three"}
                except Exception as ex:
                    tr_extra["roofline"] = {"error": repr(ex)}
                if args.cpu_seconds > 0:
                    from oracle import train as otrain
                    orc = otrain.EagerTrainer(pt_, gt_["aabb"], st_, gt_["near_far"], float(gt_["distance_scale"]), float(gt_["thr"]))
                    t0 = time.perf_counter()
                    orc.gradients(tr_rays.cpu(), tr_rgb.cpu(), Str, torch.rand(4096), True, 5)
                    tr_extra["cpu_port_forward_backward_s"] = time.perf_counter() - t0
                    tr_extra["cpu_threads"] = torch.get_num_threads()
                extras["train_step_R1"] = tr_extra
                if isinstance(tr_extra.get("reference_loop_with_ngf_optim_ms_per_iteration"), float):
                    extras["train_step_R1_autograd"] = {"ms_per_iteration": tr_extra["reference_loop_with_ngf_optim_ms_per_iteration"],
                                                        "with_torch_optim_Adam_ms": tr_extra.get("reference_loop_on_autograd_path_ms_per_iteration"),
                                                        "what": "TriPlane/main.py:272-299 as written on the drop-in field (differentiable forward, torch MSE + density_L1, "
                                                                "backward), optimizer = ngf_amd.optim.Adam(field.get_optparam_groups(), betas=(0.9, 0.99))"}
                trn.release()
                ft.release()
            except Exception as ex:
                extras["train_step_R1"] = {"error": repr(ex)}
            result["extras"] = extras
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it first so that the JSON line is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        write_side_file(result)
        print(compact_line(result), flush=True)


if __name__ == "__main__":
    main()
