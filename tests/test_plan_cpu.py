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
            h = len(log["created"]) + 1
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
    assert len(log["created"]) == 4
