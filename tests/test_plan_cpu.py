"""CPU: host logic of viprs_amd.plan that needs no device (the library calls are replaced by a recording fake)."""
import ctypes

import numpy as np


def test_placement_probe_keeps_the_fastest_candidate(monkeypatch):
    """`DeviceState._probe_placement` (viprs_amd/plan.py): six allocations, probe sweeps on each, the fastest kept and
    handed out zeroed, the other five destroyed -- driven here by a fake library whose sweep time depends on the handle."""
    from viprs_amd import plan as P

    level = {1: 0.745, 2: 0.692, 3: 0.690, 4: 0.709, 5: 0.689, 6: 0.703}      # ms per sweep of candidate 1 .. 6
    log = {"created": [], "destroyed": [], "uploads": {}, "current": None}

    class FakeLib:
        def viprs_state_create(self, out, plan, ftype, kind, width):
            h = log["next"] = log.get("next", 0) + 1          # handles are never reused (states of earlier cases die late)
            log["created"].append(h)
            ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = h
            return 0

        def viprs_state_destroy(self, h):
            log["destroyed"].append(h.value)
            return 0

        def viprs_state_upload(self, h, field, ptr):
            log["uploads"].setdefault(h.value, []).append(field)
            return 0

        def viprs_state_reset(self, h, pi):
            return 0

        def viprs_state_e_step(self, h, dq, active, n, sync):
            fake_plan.sweeps.append(level[h.value])
            return 0

        def viprs_state_synchronize(self, h):
            return 0

    class FakePlan:
        m, n_blocks, ld_dtype, handle = 250_000, 40, np.dtype(np.int8), ctypes.c_void_p(7)
        sweeps = []
        n_dense = 40

        def info(self, key):
            assert key == P.L.INFO_N_DENSE
            return self.n_dense

        def timing_reset(self):
            self.sweeps.clear()

        def timing_history(self, which=0, capacity=256):
            return list(self.sweeps)

    fake_plan = FakePlan()
    monkeypatch.setattr(P.L, "lib", FakeLib())
    monkeypatch.setattr(P.L, "check", lambda rc: None)
    ds = P.DeviceState(fake_plan, "float32", "spike_slab", 1, placement="probe")
    assert log["created"] == [1, 2, 3, 4, 5, 6]
    assert ds._h.value == 5 and ds.placement["chosen"] == 4                       # the 0.689 ms candidate
    assert sorted(log["destroyed"]) == [1, 2, 3, 4, 6]
    assert ds.placement["kernel_ms_min"] == [0.745, 0.692, 0.690, 0.709, 0.689, 0.703]
    # the survivor was zeroed after the probe: its last uploads cover every field of a spike-and-slab state
    assert set(log["uploads"][5][-9:]) == {P.DeviceState.FIELDS[k] for k in
                                           ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult", "var_gamma", "var_mu", "eta",
                                            "q", "eta_diff")}
    # small plans, float64 states, grid states and placement="off" are left alone
    log["created"].clear()
    P.DeviceState(fake_plan, "float32", "spike_slab", 1, placement="off")
    P.DeviceState(fake_plan, "float64", "spike_slab", 1, placement="probe")
    P.DeviceState(fake_plan, "float32", "grid", 32, placement="probe")
    fake_plan.m = 1000
    P.DeviceState(fake_plan, "float32", "spike_slab", 1, placement="probe")
    # ... and so are plans without dense blocks (windowed components: no panel-kernel bracket to rank by) and wide mixtures
    fake_plan.m, fake_plan.n_dense = 250_000, 0
    P.DeviceState(fake_plan, "float32", "spike_slab", 1, placement="probe")
    fake_plan.n_dense = 40
    P.DeviceState(fake_plan, "float32", "mixture", 12, placement="probe")
    assert len(log["created"]) == 6

    # timings that say nothing (no bracket recorded: zeros): no decision, the first allocation is kept
    log["created"].clear(); log["destroyed"].clear()
    saved = dict(level)
    for k in list(level):
        level[k] = 0.0
    for h in range(1, 60):
        level.setdefault(h, 0.0)
    ds = P.DeviceState(fake_plan, "float32", "spike_slab", 1, placement="probe")
    assert ds.placement["decided"] is False and ds.placement["chosen"] == 0 and ds._h.value == log["created"][0]
    assert set(log["created"][1:]) <= set(log["destroyed"]) and log["created"][0] not in log["destroyed"]

    # a probe that fails (out of memory on a candidate, a hand-off time-out on a shared GPU) must not fail the constructor:
    # the state that already exists is handed out, zeroed, and `placement` says what happened
    log["created"].clear(); log["destroyed"].clear()
    boom = {"n": 0}
    real_create = FakeLib.viprs_state_create

    def failing_create(self, out, plan, ftype, kind, width):
        boom["n"] += 1
        if boom["n"] == 3:
            raise MemoryError("hipMalloc: out of memory")
        return real_create(self, out, plan, ftype, kind, width)
    monkeypatch.setattr(FakeLib, "viprs_state_create", failing_create)
    ds = P.DeviceState(fake_plan, "float32", "spike_slab", 1, placement="probe")
    assert ds._h.value == log["created"][0] and "MemoryError" in ds.placement["error"]
    assert set(log["created"][1:]) <= set(log["destroyed"]) and log["created"][0] not in log["destroyed"]   # candidates freed
    assert set(log["uploads"][ds._h.value][-9:]) == {P.DeviceState.FIELDS[k] for k in
                                                    ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult", "var_gamma", "var_mu",
                                                     "eta", "q", "eta_diff")}
