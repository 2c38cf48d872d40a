"""Randomised parity runs (development tool, the oracle is the checker as in tests/): for SECONDS, draw a block-size mix
(tiny, panel-edge, medium / large / very large team classes, many-block mixes), an LD form, an LD dtype, a model
(spike-and-slab, mixture K, grid G with a scattered active list) and a state precision, run two sweeps through the C ABI
shim and through the oracle, and require `==` (fp32 AND float64 states).  Prints one line per case and a summary;
exit code 1 on the first mismatch (the case's seed is printed: `python tools/fuzz_parity.py 0 SEED` re-runs it alone).
    python tools/fuzz_parity.py [SECONDS] [SEED]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests import helpers as H
from tests.test_gpu_models import _run_grid, _run_mix
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs
from viprs_amd.utils import synthetic as syn
from viprs_amd.vi import e_step_hip as S

STATE = ("var_gamma", "var_mu", "eta", "q", "eta_diff")
EDGE = [1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 511, 512, 513, 1535, 1536, 1537, 1599, 1600, 1601,
        1791, 1792, 1793, 2303, 2304, 2305]


def draw_sizes(rng, budget):
    """Block sizes with sum of squares <= budget (the oracle's cost)."""
    style = rng.integers(0, 6)
    sizes = []
    if style == 0:      # many small blocks
        sizes = list(rng.integers(1, 400, size=rng.integers(5, 60)))
    elif style == 1:    # panel / class edges
        sizes = list(rng.choice(EDGE, size=rng.integers(1, 6)))
    elif style == 2:    # one large team block + company
        sizes = [int(rng.integers(2304, 5200))] + list(rng.integers(1, 900, size=rng.integers(0, 8)))
    elif style == 3:    # medium team blocks
        sizes = list(rng.integers(1600, 2304, size=rng.integers(1, 4))) + list(rng.integers(1, 700, size=rng.integers(0, 6)))
    elif style == 4:    # a mix of all classes
        sizes = [int(rng.integers(2304, 4000)), int(rng.integers(1600, 2304))] + list(rng.integers(1, 1600, size=rng.integers(1, 10)))
    else:               # a very large block
        sizes = [int(rng.integers(5200, 7000))] + list(rng.integers(1, 300, size=rng.integers(0, 4)))
    rng.shuffle(sizes)
    out, cost = [], 0
    for s in sizes:
        if cost + int(s) ** 2 > budget and out:
            continue
        out.append(int(s)); cost += int(s) ** 2
    return out


def one_case(seed):
    rng = np.random.default_rng(seed)
    model = rng.choice(["spike_slab", "spike_slab", "mixture", "grid"])
    f64 = bool(rng.integers(0, 4) == 0) and model != "grid"
    low_memory = bool(rng.integers(0, 2))
    dt = [np.float32, np.float32, np.int8, np.int16][rng.integers(0, 4)]
    if f64 and rng.integers(0, 3) == 0:
        dt = [np.float64, np.int32][rng.integers(0, 2)]
    budget = 3.0e7 if model == "spike_slab" else 1.6e7 if model == "mixture" else 2.5e7
    sizes = draw_sizes(rng, budget)
    T = np.float64 if f64 else np.float32
    kind = ["longrange", "sample", "ar1"][rng.integers(0, 3)]
    try:
        ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=dt, seed=int(rng.integers(1, 1 << 30)),
                                       kind=kind, float_precision=T)
    except np.linalg.LinAlgError:          # (the generator's Cholesky factor of a quantised block: not this tool's subject)
        return f"seed {seed}: generator could not build the problem, skipped", []
    desc = f"seed {seed}: {model}{' f64' if f64 else ''} {'upper' if low_memory else 'sym'} {np.dtype(dt).name} {kind} sizes {sizes[:8]}{'...' if len(sizes) > 8 else ''} ({len(sizes)} blocks, {sum(sizes)} SNPs)"
    if model == "spike_slab":
        st0 = inp.state_copy()
        ref = H.run_oracle(ld, inp, st0, sweeps=2)
        got = H.run_hip(ld, inp, st0, sweeps=2)
    elif model == "mixture":
        K = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 10, 12, 16, 20]))
        desc += f" K={K}"
        mix, st0 = _mixture_inputs(ld, ss, K, T=T) if f64 else _mixture_inputs(ld, ss, K)
        ref = _run_mix(O, ld, inp, mix, st0, 2)
        got = _run_mix(S, ld, inp, mix, st0, 2)
    else:
        G = int(rng.choice([1, 5, 12, 32, 33, 40]))
        n_active = int(rng.integers(1, G + 1))
        desc += f" G={G} active={n_active}"
        os.environ["VIPRS_GRID_MFMA"] = "1" if rng.integers(0, 4) else "0"
        g, st0 = _grid_inputs(ld, ss, G)
        active = rng.permutation(G)[:n_active].astype(np.int32)
        ref = _run_grid(O, ld, inp, g, st0, active)
        got = _run_grid(S, ld, inp, g, st0, active)
        desc += f" mfma={os.environ['VIPRS_GRID_MFMA']}"
    bad = []
    for k in STATE:
        a, b = np.asarray(got[k]), np.asarray(ref[k])
        if not np.array_equal(a, b):            # (float64 states too since round 5: glibc's exp and the reference's summation order)
            bad.append(f"{k}: {int((a != b).sum())} of {a.size} differ")
    return desc, bad


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 1000000
    t0, n = time.time(), 0
    while True:
        desc, bad = one_case(seed + n)
        print(("FAIL " if bad else "ok   ") + desc + ("  " + "; ".join(bad) if bad else ""), flush=True)
        n += 1
        if bad:
            sys.exit(1)
        if time.time() - t0 >= seconds:
            break
    print(f"{n} cases, all equal to the oracle", flush=True)


if __name__ == "__main__":
    main()
