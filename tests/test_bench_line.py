"""The final stdout line of bench.py must be parseable by the driver, which keeps an ~8 KB tail of stdout: round 2's line was
29 KB (extras inline, one 600-character note repeated 16 times) and the record held no value.  These tests drive bench.py's own
line builder (`compact_line`, `physical_roofs`, `load_pmc`) with a result of the shape main() assembles -- every optional block
present, every extras entry present -- and bound its size."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _fat_result(monkeypatch):
    monkeypatch.setattr(bench, "so_sha16", lambda: "0" * 16)
    pmc = bench.load_pmc("triplane_R1")
    assert pmc is not None, "profiles/ holds no PMC summary of the headline workload"
    phys = bench.physical_roofs(pmc, 9.8123456789)
    extras = {}
    for i in range(24):         # main() has 16 today
        extras[f"triplane_R{i}_some_long_variant_tag_split_bf16"] = {
            "Mray/s": 65.123456789 + i, "kernel_ms": 9.87654321, "active_samples_per_ray": 46.3121515625,
            "executed_mfma_TFLOPs": 82.123456, "physical": phys, "note": "x" * 600}
    extras["train_step_R1"] = {"ms_per_iteration": 1.23456789, "atomic_roof": {"floor_ms": 0.35, "note": "y" * 600}}
    extras["broken"] = {"error": "RuntimeError('" + "z" * 2000 + "')"}
    return {
        "metric": "Mray/sec (800x800 lego-style frame, 192 samples/ray)", "value": 65.30420130967548, "unit": "Mray/s", "n_gpus": 8,
        "steps": 20, "warmup": 5, "ms_per_step": 9.800288299447857, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "TriPlane 800x800 frame, S=192, preset R1 (seeded random planes 256^2, dense density preset), gauge on, white_bg",
                   "rays_per_step": 640000, "samples_per_ray": 192,
                   "sharding": "rays x8 (10-row blocks, round robin) + double-buffered RCCL all_gather", "bake_density": 0, "bake_color": 0,
                   "knobs": "waves=12,tile_w=8,kernel=1"},
        "roofline": {"bound": "mfma", "achieved": 82.29929213137213, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.5231995685401915,
                     "traffic": 720829197.7142856, "flops_per_launch": 807705214976.0, "physical": phys, "kernel": "ngf::render_kernel",
                     "kernel_ms": 9.814242553710937, "active_samples_per_ray": 46.3121515625, "evaluated_samples_per_ray": 141.418665625,
                     "mlp_passes": 1889939.0, "algorithmic_d3": {"unit": "GB/s", "achieved": 17778.6, "bytes_per_launch": 174483966208.0}},
        "cpu_baseline": {"value": 0.009773045519438195, "unit": "Mray/s", "cores": 32, "kind": "port",
                         "host": "AMD EPYC 9575F 64-Core Processor, 256 logical CPUs; thread counts tried: 8/16/32/64, fastest kept",
                         "sample": "29 chunks x 4096 rays of the same frame, torch-eager port of Base.forward, 12.2 s"},
        "parity": {"max_abs_err_vs_cpu_port": 1.1920928955078125e-06, "max_rel_err_vs_cpu_port": 2.337068735869252e-06,
                   "psnr_vs_cpu_port_db": 132.50515620545707, "cpu_active_fraction": 0.24215088836077986},
        "speedup_vs_cpu_port": 6682.07276634372, "gathered_frame_bit_identical_to_single_gpu_render": True, "all_gather_ms": 0.123456789,
        "all_gather_bytes_per_rank": 1280000, "shard_kernel_ms": {"min": 1.0456789, "max": 1.0656789, "rank0": 1.05123456},
        "critical_path_ms": {"render": 1.0656789, "all_gather": 0.123456789, "reorder": 0.0456789, "sum": 1.234814589, "note": "unpipelined, max over ranks"},
        "ms_per_step_median": 9.7912345678, "launch_ms": {"min": 9.7712345678, "median": 9.7912345678, "max": 9.9912345678},
        "value_from_median_launch": 65.4, "extras": extras}


def test_compact_line_fits_the_driver_tail(monkeypatch):
    res = _fat_result(monkeypatch)
    line = bench.compact_line(res)
    assert len(line) < bench.LINE_LIMIT <= 4096 and "\n" not in line
    d = json.loads(line)
    for k in bench.FIXED_KEYS:
        assert k in d
    assert d["value"] == pytest.approx(res["value"], rel=1e-5) and d["ms_per_step"] == pytest.approx(res["ms_per_step"], rel=1e-5)
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in rf
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    ph = rf["physical"]
    for k in ("simd_busy", "mfma_busy", "valu_busy_raw", "ta_busy", "l2_hit"):
        assert isinstance(ph[k], float) and 0.0 <= ph[k] <= 1.5
    assert not any(isinstance(v, str) and len(v) > 80 for v in ph.values())          # numbers and a file name, no prose
    cb = d["cpu_baseline"]
    assert set(cb) == {"value", "unit", "cores", "kind", "sample"} and cb["kind"] in ("port", "reference")
    assert d["all_gather_ms"] > 0 and d["gathered_frame_bit_identical_to_single_gpu_render"] is True
    assert set(d["launch_ms"]) == {"min", "median", "max"} and d["launch_ms"]["min"] <= d["ms_per_step_median"] <= d["launch_ms"]["max"]
    assert d["shard_kernel_ms"]["min"] <= d["shard_kernel_ms"]["max"] and set(d["critical_path_ms"]) >= {"render", "all_gather", "reorder", "sum"}
    assert len(d["extras_Mray_s"]) == 24 and d["train_ms_per_iteration"] > 0
    assert "extras" not in d and "notes" not in d


def test_compact_line_refuses_to_grow(monkeypatch):
    res = _fat_result(monkeypatch)
    res["config"]["workload"] = "w" * 5000
    with pytest.raises(RuntimeError):
        bench.compact_line(res)


def test_side_file_keeps_everything(monkeypatch, tmp_path):
    res = _fat_result(monkeypatch)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.write_side_file(res)
    full = json.load(open(tmp_path / "bench_extras.json"))
    assert len(full["extras"]) == 26 and "simd_vector_datapath" in full["notes"]
    assert full["roofline"]["algorithmic_d3"]["bytes_per_launch"] == res["roofline"]["algorithmic_d3"]["bytes_per_launch"]


@pytest.mark.parametrize("n", [2, 4, 8])
def test_self_launch_command_line(n):
    """`python bench.py --gpus N` without a launcher starts its own N ranks -- the command line is the driver's own
    multi-GPU one (one node, N processes, loop-back rendezvous) with the caller's arguments passed through unchanged."""
    argv = ["--gpus", str(n), "--steps", "7", "--warmup", "2"]
    cmd = bench.self_launch_command(n, argv, 29617)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and f"--nproc-per-node={n}" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29617"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv and all(a.startswith("-") or a in ("torch.distributed.run", "127.0.0.1", "29617") for a in cmd[1:script])
    p = bench.free_port()
    assert 1024 < p < 65536


def test_self_launch_refuses_more_ranks_than_gpus(monkeypatch):
    """--gpus N on a node with fewer GPUs: a clear non-zero exit, nothing launched, no hang."""
    import subprocess
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    called = []
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: called.append(a) or 0)
    with pytest.raises(SystemExit) as ex:
        bench.self_launch(8, ["--gpus", "8"])
    assert "1 GPU" in str(ex.value) and "--gpus 8" in str(ex.value) and not called
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    with pytest.raises(SystemExit) as ex:
        bench.self_launch(8, ["--gpus", "8", "--steps", "3"])
    assert ex.value.code == 0 and len(called) == 1 and called[0][0][-4:] == ["--gpus", "8", "--steps", "3"]


def test_bench_without_a_gpu_exits_with_a_message():
    """The product path has no CPU fallback: on a box without a GPU the bench says so (rc != 0) whatever --gpus is, and never waits for a rendezvous."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for n in ("1", "8"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", n], env=env, capture_output=True, text=True, timeout=300)
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
        assert r.returncode != 0 and "MI355X" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "10-row blocks" in r.stderr


def _worker_time_steps(rank, world, port, tmp):
    """bench.time_steps at world 2 without a device (gloo; the device calls are stubbed): rank 1 arrives 0.4 s late, every step's exchange carries
    the step's sequence number, and every rank must see ITS number from every other rank -- i.e. all ranks issued the same collectives in the same order."""
    import time
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class _Ev:
        def __init__(self, enable_timing=False): pass
        def record(self, *a): pass
    bench.torch.cuda.synchronize = lambda *a, **k: None
    bench.torch.cuda.Event = _Ev
    bench.CLOCK_PREAMBLE_MS = 60.0
    if rank == 1:
        time.sleep(0.4)                                   # a rank that built its field more slowly
    seq, pre, bad, works = [0], [0], [], []

    def step(pair):
        mine = torch.tensor([float(seq[0])])
        got = torch.empty(world)
        works.append((dist.all_gather_into_tensor(got, mine, async_op=True), got, seq[0]))
        seq[0] += 1

    def finish():
        while works:
            w, got, k = works.pop(0)
            w.wait()
            if not bool((got == float(k)).all()):
                bad.append((k, got.tolist()))

    def preamble():
        pre[0] += 1
        time.sleep(0.001 * (1 + 2 * rank))                # ranks get through a different number of preamble launches

    try:
        bench.time_steps(step, 5, 2, torch.device("cpu"), True, finish)
        refused = False
    except ValueError:
        refused = True                                    # no collective-free preamble launch given at N > 1: refused before anything is issued
    el = bench.time_steps(step, 5, 2, torch.device("cpu"), True, finish, None, preamble=preamble)
    finish()
    counts = [None] * world
    dist.all_gather_object(counts, (seq[0], pre[0]))
    ok = refused and not bad and seq[0] == 7 and all(c[0] == 7 for c in counts) and pre[0] >= 4 and el > 0
    open(os.path.join(tmp, f"ok{rank}"), "w").write(f"{int(ok)} refused={refused} bad={bad} counts={counts} elapsed={el}")
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_timed_region_issues_the_same_collectives_on_every_rank(tmp_path):
    """The clock preamble of bench.time_steps runs for a wall-clock time, so its trip count is rank-local: it must not contain a collective (as first
    written in round 6 it repeated the whole pipelined step, exchange included -- at N > 1 the ranks would have issued different numbers of
    all-gathers and paired them with the warm-up's or the barrier's; one GPU cannot show that).  World 2 over gloo, rank 1 late, different preamble
    trip counts per rank: every step's exchange pairs with the same step on the other rank, and a missing preamble launch is refused."""
    import torch.multiprocessing as mp
    port = bench.free_port()
    mp.spawn(_worker_time_steps, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [open(tmp_path / f"ok{r}").read() for r in range(2)]
    assert all(r.startswith("1 ") for r in res), res
    pre = [eval(r.split("counts=")[1].split(" elapsed")[0]) for r in res]
    assert pre[0][0][1] != pre[0][1][1], "the test meant the ranks to run different numbers of preamble launches"


def test_environment_defaults_are_set_before_the_runtime_starts():
    """bench.py fixes three switches BEFORE torch is imported (the HIP runtime / RCCL read them when they start): dmabuf IPC handles, RCCL's channel cap,
    and -- for the ranks of an N > 1 job only -- eight HIP hardware queues (the pipeline's five streams must not share one: profiles/r06_exchange_contention.txt).
    A value the launcher set wins."""
    import subprocess
    import sys
    code = ("import os, sys; sys.argv = ['bench.py']; import importlib.util as u; s = u.spec_from_file_location('bench_env', %r); m = u.module_from_spec(s); s.loader.exec_module(m); "
            "print(os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), os.environ.get('NCCL_MAX_NCHANNELS'), os.environ.get('GPU_MAX_HW_QUEUES'))" % os.path.join(ROOT, "bench.py"))

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "GPU_MAX_HW_QUEUES", "NCCL_MAX_NCHANNELS", "HSA_ENABLE_IPC_MODE_LEGACY", "NGF_BENCH_FORCE_DIST", "NGF_BENCH_SELF_LAUNCHED")}
        e.update(env)
        return subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300).stdout.split()

    assert run() == ["0", "8", "None"]                                        # N = 1: the runtime's default queue count
    assert run(WORLD_SIZE="8") == ["0", "8", "8"]                             # a rank of an 8-GPU job
    assert run(NGF_BENCH_SELF_LAUNCHED="1", WORLD_SIZE="1") == ["0", "8", "8"]
    assert run(WORLD_SIZE="8", GPU_MAX_HW_QUEUES="4", NCCL_MAX_NCHANNELS="16") == ["0", "16", "4"]      # the launcher's values win
