"""CPU: synthetic mixture / grid inputs and the bench's CPU-baseline leg (oracle timed on the host)."""
import numpy as np
import pytest

import bench
from oracle import oracle as O
from viprs_amd.utils import synthetic as syn


@pytest.fixture(scope="module")
def problem():
    ld = syn.make_ld(np.array([40, 90, 33]), low_memory=False, ld_dtype=np.dtype("float32"), seed=11)
    ss = syn.make_sumstats(ld, seed=11)
    return ld, ss, syn.make_inputs(ss)


def test_mixture_inputs_layout(problem):
    ld, ss, _ = problem
    x = syn.make_mixture_inputs(ss, 4)
    assert x["log_null_pi"].shape == (ld.m,) and x["log_null_pi"].dtype == np.float32
    for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
        assert x[k].shape == (ld.m, 4) and x[k].flags.c_contiguous and np.all(np.isfinite(x[k]))
    assert 0.0 < x["pi"] < 0.25


def test_grid_inputs_layout(problem):
    ld, ss, _ = problem
    x = syn.make_grid_inputs(ss, 6)
    for k in ("u_logs", "half_var_tau", "mu_mult"):
        assert x[k].shape == (ld.m, 6) and x[k].flags.f_contiguous and np.all(np.isfinite(x[k]))
    # columns differ (one model per column)
    assert not np.allclose(x["u_logs"][:, 0], x["u_logs"][:, 5])


@pytest.mark.parametrize("model, width", [("spike_slab", 1), ("mixture", 3), ("grid", 5)])
def test_cpu_baseline_leg(problem, model, width):
    ld, ss, inp = problem
    extra, pi0 = None, inp.pi
    if model == "mixture":
        extra = syn.make_mixture_inputs(ss, width)
        pi0 = extra.pop("pi")
    elif model == "grid":
        extra = syn.make_grid_inputs(ss, width)
        pi0 = extra.pop("pi")
    r = bench.cpu_baseline(ld, inp, 0.2, model, width, extra, pi0)
    assert r["value"] > 0 and r["single_thread_value"] > 0 and r["unit"] == "SNP-updates/s"
    assert r["kind"] == ("reference" if O.have_reference() else "port")
    assert f"model={model}" in r["sample"]


def test_strong_scaling_shards_hold_the_same_numbers_as_the_whole_workload():
    """bench.py --gpus N: the blocks of ONE workload are dealt to the ranks; a block's LD, summary statistics and
    inputs are the same numbers whichever rank builds it (every random draw is made for the whole workload)."""
    class A:
        math = "exact"
    sizes = np.array([40, 90, 33, 64, 120, 51, 77], dtype=np.int64)
    ld, ss, inp, m = bench.build_workload(A, sizes, None, 11, False, np.dtype("float32"))
    assert m == sizes.sum()
    parts = bench.shard_blocks_lpt(sizes, 3)
    assert sorted(b for p in parts for b in p) == list(range(len(sizes)))
    for mine in parts:
        ld_r, ss_r, inp_r, m_all = bench.build_workload(A, sizes, mine, 11, False, np.dtype("float32"))
        assert m_all == m
        idx = np.concatenate([np.arange(ld.block_start[b], ld.block_start[b + 1]) for b in mine])
        for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
            np.testing.assert_array_equal(getattr(inp_r, k), getattr(inp, k)[idx])
        # LD rows of the shard == the rows of the whole workload (symmetric form: a block's rows are its b x b entries)
        off = 0
        for b in mine:
            s, e = int(ld.block_start[b]), int(ld.block_start[b + 1])
            n = (e - s) ** 2
            np.testing.assert_array_equal(ld_r.ld_data[off:off + n], ld.ld_data[int(ld.ld_indptr[s]):int(ld.ld_indptr[e])])
            off += n


def test_chain_aware_block_sharding():
    from viprs_amd import parallel as P
    sizes = syn.block_sizes("cfg3")
    for world in (2, 4, 8):
        owner = P.shard_blocks(sizes, world)
        assert owner.shape == sizes.shape and set(owner) == set(range(world))
        loads = np.array([P.block_cost(sizes[owner == r]).sum() for r in range(world)])
        assert loads.max() / loads.mean() < 1.02                     # LPT over 1 700 blocks balances to ~1 %
        # the largest blocks go to different ranks (their serial chains are each rank's floor)
        top = np.argsort(-sizes)[:world]
        assert len(set(owner[top])) == world
    # small blocks cost their chain share, large blocks their bytes
    assert P.block_cost(100) == pytest.approx(100 * P.CHAIN_STEP_S / P.CHAIN_SLOTS)
    assert P.block_cost(4000) == pytest.approx(4000 * 4000 * 4 / P.HBM_STREAM_BYTES_PER_S)


def test_block_shard_slices_ld_rows_into_a_local_numbering():
    from viprs_amd.parallel import BlockShard
    for low_memory in (False, True):
        ld = syn.make_ld(np.array([5, 9, 4, 7]), low_memory=low_memory, seed=3)
        sh = BlockShard(ld.block_start, [3, 1])
        assert sh.m == 16 and list(sh.blocks) == [1, 3]
        lb, ip, data = sh.slice_ld(ld.ld_left_bound, ld.ld_indptr, ld.ld_data)
        ref = syn.make_ld(np.array([9, 7]), low_memory=low_memory, seed=3, rho=ld.rho[[1, 3]])
        np.testing.assert_array_equal(lb, ref.ld_left_bound)
        np.testing.assert_array_equal(ip, ref.ld_indptr)
        np.testing.assert_array_equal(data, ref.ld_data)
        full = np.arange(ld.m, dtype=np.float32)
        np.testing.assert_array_equal(sh.scatter(sh.take(full), np.zeros(ld.m, np.float32))[sh.index], full[sh.index])


_BENCH_WORKER = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from viprs_amd import _lib, plan as plan_mod

# ---- stubbed device layer: no GPU here; everything ABOVE it (sharding, workload construction, collectives, the JSON
# line) is bench.py's real multi-rank logic --------------------------------------------------------------------------
_lib.device_count = lambda: 1
class _FakeLib:
    def __getattr__(self, name):
        return lambda *a, **k: 0
_lib.lib = _FakeLib()
_lib.check = lambda rc: None

class FakePlan:
    def __init__(self, lb, ip, data, low_memory, device=0, math_mode="exact"):
        self.m = int(lb.shape[0]); self.n = 0
    @classmethod
    def synthetic(cls, ld, device=0, math_mode="exact"):
        assert ld.ld_data is None and ld.kind == "longrange" and len(ld.params) == len(ld.block_start) - 1   # a skeleton: nothing built on the host
        return cls(ld.ld_left_bound, ld.ld_indptr, None, ld.low_memory, device, math_mode)
    def timing_reset(self): self.n = 0
    def timing_history(self, which=0, capacity=256): return [0.5 + 0.01 * int(os.environ["RANK"])] * max(self.n, 1)
    def last_skipped(self): return 7
    def close(self): pass

class FakeState:
    def __init__(self, plan, dtype, model, width, placement=None): self.plan = plan
    def upload(self, name, arr): assert arr.shape[0] == self.plan.m, (name, arr.shape, self.plan.m)
    def reset(self, pi): pass
    def e_step(self, dq, active=None, sync=True): self.plan.n += 1; time.sleep(0.001)
    def synchronize(self): pass
    def close(self): pass

plan_mod.LDPlan, plan_mod.DeviceState = FakePlan, FakeState
import bench
sys.argv = ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "cfg2", "--cpu-seconds", "0"] + {extra!r}
bench.main()
print("RANK_DONE", os.environ["RANK"])
"""


@pytest.mark.parametrize("extra", [[], ["--scaling", "weak"]], ids=["strong-default", "weak"])
def test_bench_main_two_ranks_end_to_end_over_the_file_transport(tmp_path, extra):
    """bench.py's N > 1 path run for real with two processes (torch.distributed.run-style environment,
    VIPRS_BENCH_COMM=file, device layer stubbed): ONE JSON line from rank 0 whose `n_gpus`, `comm`, `snps_total`,
    `scaling` and per-rank vectors describe the default run (strong = BASELINE configs[2]: ONE workload sharded by LD block,
    the weak figure beside it under its own metric text) or, with --scaling weak, one workload per rank (the strong figure
    beside it)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_BENCH_WORKER.format(root=root, extra=extra))
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               VIPRS_BENCH_COMM="file", TMPDIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_DONE {r}" in o, o[-3000:]
    lines = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1].splitlines() if l.startswith("{")]   # rank 0 only, one line
    out = json.loads(lines[0])
    sizes = bench.config_sizes("cfg2", 7209)
    weak = "weak" in extra
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == ("weak" if weak else "strong")
    assert "file transport" in out["config"]["comm"]
    pr = out["per_rank"]
    assert ("1M SNPs, ~1700 LD blocks" in out["metric"]) and (("independent" in out["metric"]) == weak)
    if weak:
        assert out["config"]["snps_total"] == 2 * int(sizes.sum()) and pr["snps"] == [int(sizes.sum())] * 2
        assert "weak_scaling" not in out
        st = out["strong_scaling"]
        assert st["snps_total"] == int(sizes.sum()) and len(st["kernel_ms_avg_per_rank"]) == 2
        assert max(st["largest_block_per_rank"]) == int(sizes.max())
        assert st["value"] == pytest.approx(st["snps_total"] / (st["ms_per_step"] * 1e-3), rel=1e-6)
    else:
        # ONE workload: the shards partition its blocks, `value` counts its SNPs once
        assert out["config"]["snps_total"] == int(sizes.sum()) == sum(pr["snps"])
        assert sum(pr["ld_blocks"]) == len(sizes) and max(pr["largest_block"]) == int(sizes.max())
        parts = bench.shard_blocks_lpt(sizes, 2)
        assert pr["snps"] == [int(sizes[p].sum()) for p in parts]
        # (the default LD form is the reference's: upper-triangular, b (b - 1) / 2 stored entries per block, each used twice)
        assert pr["algorithmic_bytes"] == [int(4 * ((sizes[p] ** 2).sum() - sizes[p].sum()) + 68 * sizes[p].sum()) for p in parts]
        assert out["default_ld_form"]["is_value"] is True and "upper-triangular" in out["config"]["ld_form"]
        assert out["metric"] == "SNP-updates/sec/E-step (1M SNPs, ~1700 LD blocks)"          # BASELINE's metric, one workload
        assert out["weak_scaling"]["snps_per_gpu"] == int(sizes.sum()) and "INDEPENDENT" in out["weak_scaling"]["metric"]
        assert len(out["weak_scaling"]["kernel_ms_avg_per_rank"]) == 2
        c = out["strong_scaling_ceiling"]
        assert c["largest_block_snps"] == int(sizes.max()) and c["max_speedup_over_one_gpu"] >= 1.0
        assert c["largest_block_chain_ms"] == pytest.approx(int(sizes.max()) * 135e-6)
        assert out["startup_s_rank0"]["ld_entries"] == "generated on the device"
    assert out["value"] == pytest.approx(out["config"]["snps_total"] * 3 / (out["ms_per_step"] * 3e-3), rel=1e-6)
    assert pr["kernel_ms_avg"] == pytest.approx([0.5, 0.51]) and len(pr["time_model_ms"]) == 2
    assert out["roofline"]["kernel_ms_avg"] == pytest.approx(0.51)            # slowest rank
    assert out["roofline"]["algorithmic_bytes_per_launch"] == sum(pr["algorithmic_bytes"])
    # the transport cleaned up after itself (stale rendezvous files)
    left = [f for f in os.listdir(tmp_path) if f.startswith("viprs_filecomm")]
    assert left == [], left


# ---- the bare command: `python bench.py --gpus 2` with NO launcher environment -----------------------------------------
# The device layer is stubbed from OUTSIDE bench.py: a sitecustomize module on PYTHONPATH patches viprs_amd in every
# rank process (and only there: the launching parent never imports the library).
_SITE_STUB = r"""
import os, sys, time
if "RANK" in os.environ:
    sys.path.insert(0, {root!r})
    from viprs_amd import _lib, plan as plan_mod
    _lib.device_count = lambda: {ndev}
    class _FakeLib:
        def __getattr__(self, name):
            return lambda *a, **k: 0
    _lib.lib = _FakeLib()
    _lib.check = lambda rc: None
    class FakePlan:
        def __init__(self, lb, ip, data, low_memory, device=0, math_mode="exact"):
            self.m = int(lb.shape[0]); self.n = 0
        @classmethod
        def synthetic(cls, ld, device=0, math_mode="exact"):
            assert ld.ld_data is None and ld.kind == "longrange"
            return cls(ld.ld_left_bound, ld.ld_indptr, None, ld.low_memory, device, math_mode)
        def timing_reset(self): self.n = 0
        def timing_history(self, which=0, capacity=256): return [0.5 + 0.01 * int(os.environ["RANK"])] * max(self.n, 1)
        def last_skipped(self): return 7
        def close(self): pass
    class FakeState:
        def __init__(self, plan, dtype, model, width, placement=None): self.plan = plan
        def upload(self, name, arr): assert arr.shape[0] == self.plan.m
        def reset(self, pi): pass
        def e_step(self, dq, active=None, sync=True): self.plan.n += 1; time.sleep(0.001)
        def synchronize(self): pass
        def close(self): pass
    plan_mod.LDPlan, plan_mod.DeviceState = FakePlan, FakeState
"""


def _run_bare_bench(tmp_path, argv, ndev, env_extra):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "sitecustomize.py").write_text(_SITE_STUB.format(root=root, ndev=ndev))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VIPRS_BENCH_COMM")}
    env.update(PYTHONPATH=str(tmp_path) + os.pathsep + env.get("PYTHONPATH", ""), TMPDIR=str(tmp_path), **env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True,
                          timeout=600)


def test_bare_bench_gpus_2_starts_two_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` without WORLD_SIZE must not run ONE process and print n_gpus 2 -- the
    parent starts two fresh rank processes, relays rank 0's single line, and the line shows both ranks took part."""
    import json
    r = _run_bare_bench(tmp_path, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "cfg2", "--cpu-seconds", "0"],
                        ndev=2, env_extra={"VIPRS_BENCH_COMM": "file"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    out = json.loads(lines[0])
    sizes = bench.config_sizes("cfg2", 7209)
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["rccl_ranks"] is None
    assert "file transport" in out["config"]["comm"]
    pr = out["per_rank"]
    # the default series over N is strong (BASELINE configs[2]): ONE workload sharded over the ranks; the weak figure sits beside it
    assert out["scaling"] == "strong" and sum(pr["snps"]) == int(sizes.sum()) == out["config"]["snps_total"]
    assert out["weak_scaling"]["snps_total"] == 2 * int(sizes.sum())
    assert pr["kernel_ms_avg"] == pytest.approx([0.5, 0.51])                 # each rank reported under its own RANK
    assert out["roofline"]["peak"] == 2 * bench.HBM_PEAK_GBS


def test_bare_bench_refuses_more_ranks_than_devices_and_mismatched_world(tmp_path):
    """No mislabelled line, ever: more ranks than HIP devices (outside the labelled dry-run transport) and a launcher
    world size different from --gpus both end non-zero without a JSON line."""
    r = _run_bare_bench(tmp_path, ["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "cfg1", "--cpu-seconds", "0"],
                        ndev=1, env_extra={})
    assert r.returncode != 0 and "{" not in r.stdout, (r.stdout, r.stderr[-2000:])
    assert "only 1 HIP device" in r.stderr
    r = _run_bare_bench(tmp_path, ["--gpus", "4", "--steps", "2", "--warmup", "1", "--config", "cfg1", "--cpu-seconds", "0"],
                        ndev=4, env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "{" not in r.stdout and "WORLD_SIZE=2" in r.stderr


def test_sweep_time_model_names_its_bound():
    """Secondaries carry the time model of their sweep: max(bytes / copy rate, largest block's chain, total chain
    work / chains in flight) -- a reader can tell "chain-bound at 90 % of its floor" from "3 x off"."""
    sizes = bench.config_sizes("cfg3", 7209)
    by = 4 * int((sizes.astype(np.int64) ** 2).sum()) + 68 * int(sizes.sum())
    t, bound, ns, terms = bench.sweep_time_model(sizes, by, "spike_slab", "exact")
    assert bound == "hbm_stream" and t == pytest.approx(by / 6.3e12 * 1e3) and ns == 135.0
    t8, bound8, _, terms8 = bench.sweep_time_model(sizes, by // 4, "spike_slab", "exact")       # int8 LD: the chain of the largest block
    assert bound8 == "largest_block_chain" and t8 == pytest.approx(sizes.max() * 135e-6)
    tf, boundf, nsf, _ = bench.sweep_time_model(sizes, by // 4, "spike_slab", "fast")
    assert nsf < ns and tf < t8
    tg, boundg, _, _ = bench.sweep_time_model(sizes, by, "grid", "exact")
    assert boundg == "chain_throughput" and tg == pytest.approx(sizes.sum() * 430e-6 / 256)


def test_bench_under_the_real_torch_launcher(tmp_path):
    """The driver's own N > 1 command line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 ...` (device layer stubbed through sitecustomize, file transport): the
    environment the elastic agent really sets (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / TORCHELASTIC_RUN_ID, one shared
    parent process) drives the rendezvous keys, and rank 0 prints the one line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "sitecustomize.py").write_text(_SITE_STUB.format(root=root, ndev=2))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VIPRS_BENCH_COMM")}
    env.update(PYTHONPATH=str(tmp_path) + os.pathsep + env.get("PYTHONPATH", ""), TMPDIR=str(tmp_path), VIPRS_BENCH_COMM="file")
    port = 29900 + os.getpid() % 90
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "cfg2", "--cpu-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and len(out["per_rank"]["snps"]) == 2
    assert out["per_rank"]["kernel_ms_avg"] == pytest.approx([0.5, 0.51])
    left = [f for f in os.listdir(tmp_path) if f.startswith("viprs_filecomm") or f.startswith("viprs_comm")]
    assert left == [], left


def test_clean_kernel_times_uses_the_host_stamp_not_the_kernel_time():
    """`bench.clean_kernel_times`: a sweep leaves the mean only on EVIDENCE -- host time stamped inside its event bracket --
    never because its kernel time is large."""
    import bench
    k = [2.0954, 0.6944, 0.6946, 0.7016, 1.9, 0.6925]
    host = [1.4198, 0.3322, 0.0078, 0.0075, 0.0068, 0.0077]
    good, bad = bench.clean_kernel_times(k, host)
    assert bad == [0, 1] and good == [0.6946, 0.7016, 1.9, 0.6925]          # the 1.9 ms sweep has no host time: it stays
    assert bench.clean_kernel_times(k, []) == (k, [])                        # no stamps (older library): nothing is dropped
    assert bench.clean_kernel_times([0.7, 0.7], [0.5, 0.6]) == ([0.7, 0.7], [0, 1])      # never an empty mean
