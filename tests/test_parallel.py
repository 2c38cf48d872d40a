"""CPU: the file rendezvous of the multi-GPU path (rank 0's RCCL unique id / FileComm's directory nonce reach the
other ranks without any rank ever accepting a file of an earlier, crashed run) and the FileComm test transport."""
import os
import threading

import numpy as np
import pytest

from viprs_amd import parallel as P


def _run_ranks(world, fn):
    out, err = [None] * world, []

    def go(r):
        try:
            out[r] = fn(r)
        except Exception as e:              # noqa: BLE001
            err.append((r, e))

    ts = [threading.Thread(target=go, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not err, err
    return out


def test_root_broadcast_ignores_stale_files_of_a_crashed_run(tmp_path):
    base = str(tmp_path / "viprs_comm_test.id.0")
    # litter of an earlier run with the same key: an id file of the old scheme, a stale request, a stale response
    for name, payload in ((base, b"S" * 128), (base + ".req.1.deadbeefdeadbeef", b""), (base + ".rsp.1.deadbeefdeadbeef", b"S" * 128),
                          (base + ".rsp.2.0123456789abcdef", b"S" * 128)):
        with open(name, "wb") as f:
            f.write(payload)
    fresh = b"F" * 128
    bcs = [None] * 3

    def rank(r):
        bcs[r] = P._RootBroadcast(r, base, (lambda: fresh) if r == 0 else None, timeout_s=30)
        return bcs[r].payload

    got = _run_ranks(3, rank)
    assert got == [fresh, fresh, fresh]                     # nobody took the stale id
    bcs[0].finish()
    left = [f for f in os.listdir(tmp_path) if ".req." in f or ".rsp." in f]
    assert left == []                                       # rank 0 swept this base's litter, stale files included


def test_exchange_unique_id_rejects_a_malformed_id(tmp_path, monkeypatch):
    monkeypatch.setenv("VIPRS_COMM_ID_FILE", str(tmp_path / "id"))
    try:
        P._exchange_unique_id(0, 2, lambda: b"short")
    except RuntimeError as e:
        assert "malformed" in str(e)
    else:
        raise AssertionError("a 5-byte id was accepted")


def test_launch_key_names_the_parent_process_and_its_start_time(monkeypatch):
    monkeypatch.setenv("MASTER_PORT", "29512")
    monkeypatch.setenv("VIPRS_RUN_ID", "abc")
    k = P._launch_key()
    parts = k.split("_")
    assert parts[0] == "29512" and "abc" in parts and str(os.getppid()) in parts
    assert parts[-1].isdigit() and int(parts[-1]) > 0       # the parent's start time: a reused pid is a different key


def test_file_comm_collectives_and_cleanup(tmp_path):
    root = str(tmp_path / "fc")
    # an earlier run's directory of the same launch key with step files in it: never read
    os.makedirs(root + ".0.feedfacefeedface")
    np.save(root + ".0.feedfacefeedface/1_1.npy", np.array([999.0]))
    seqs = [P._COMM_SEQ]

    def rank(r):
        # (threads share the module-level communicator counter; give every rank the same base explicitly)
        P._COMM_SEQ = seqs[0]
        c = P.FileComm(r, 3, root=root)
        s = c.allreduce_sum(np.array([1.0 + r, 10.0]))
        m = c.allreduce_max(np.array([float(r)]))
        for _ in range(5):
            c.barrier()
        d = c.root
        c.close()
        return s, m, d

    res = _run_ranks(3, rank)
    for s, m, d in res:
        np.testing.assert_array_equal(s, [6.0, 30.0])
        np.testing.assert_array_equal(m, [2.0])
    assert len({d for _, _, d in res}) == 1 and not res[0][2].endswith("feedfacefeedface")
    assert not os.path.exists(res[0][2])                    # removed by close()


def test_root_broadcast_survives_glob_characters_and_fails_fast(tmp_path):
    """a '[' in TMPDIR / the run id must not hide the request files from rank 0 (glob.escape), and when rank 0
    fails after the broadcast started, a waiting rank raises at once instead of waiting out the 300 s timeout."""
    import threading
    from viprs_amd.parallel import _RootBroadcast
    base = str(tmp_path / "weird[1]*dir?" / "id")
    os.makedirs(os.path.dirname(base))
    got = {}
    root = _RootBroadcast(0, base, lambda: b"x" * 128)
    t = threading.Thread(target=lambda: got.setdefault("p", _RootBroadcast(1, base, timeout_s=20).payload))
    t.start(); t.join(30)
    assert got.get("p") == b"x" * 128
    root.finish()
    # failure path
    root2 = _RootBroadcast(0, base + ".b", lambda: b"y")
    root2._stop.set(); root2._thread.join()            # rank 0 stops answering (as if stuck in a failing collective) ...
    root2.fail()                                       # ... and reports it
    err = {}
    def waiter():
        try:
            _RootBroadcast(2, base + ".b", timeout_s=60)
        except RuntimeError as e:
            err["e"] = str(e)
    import time
    t0 = time.time()
    t = threading.Thread(target=waiter); t.start(); t.join(30)
    assert "rank 0 reported a failure" in err.get("e", "") and time.time() - t0 < 10


def test_root_broadcast_failure_marker_does_not_outlive_its_launch(tmp_path, monkeypatch):
    """on a FIXED base (VIPRS_COMM_ID_FILE, or a reused port / run id / parent) a failed launch must not fail
    the next one: failure markers are keyed by the requester's nonce, rank 0 clears stale ones when it starts, and a
    second launch on the same base goes through while the first one's rank 0 is still in its failure grace period."""
    import threading
    import time
    from viprs_amd.parallel import _RootBroadcast
    monkeypatch.setattr(_RootBroadcast, "FAIL_GRACE_S", 1.0)
    base = str(tmp_path / "fixed.id")
    # launch 1: rank 0 fails; a waiting rank and a late rank both fail fast
    root1 = _RootBroadcast(0, base, lambda: b"a" * 128)
    root1.fail()
    root1.finish()                                           # (RcclComm's `finally`): must not hide the failure
    for _ in range(2):
        t0 = time.time()
        with pytest.raises(RuntimeError, match="rank 0 reported a failure"):
            _RootBroadcast(1, base, timeout_s=20)
        assert time.time() - t0 < 5
    # a stale marker as a crashed process would leave it (old scheme: base-wide `.err`; new scheme: somebody else's nonce)
    for stale in (base + ".err", base + ".err.1.deadbeefdeadbeef"):
        open(stale, "wb").close()
    time.sleep(1.3)                                          # launch 1's grace period ends; its helper sweeps and exits
    root1._thread.join(5)
    assert not root1._thread.is_alive()
    open(base + ".err", "wb").close()
    # launch 2 on the same base
    root2 = _RootBroadcast(0, base, lambda: b"b" * 128)
    assert not os.path.exists(base + ".err")
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("p", _RootBroadcast(1, base, timeout_s=20).payload))
    t.start(); t.join(30)
    assert got.get("p") == b"b" * 128
    root2.finish()
    assert [f for f in os.listdir(tmp_path) if f.startswith("fixed.id")] == []
