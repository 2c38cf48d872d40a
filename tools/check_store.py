#!/usr/bin/env python3
"""Pin `viprs_amd.io.zarr_ld.ZarrLDMatrix` against magenpy -- to be run WHERE MAGENPY IS INSTALLED.

The reference reads LD stores through magenpy (viprs/model/VIPRS.py:153-172: `ld_mat.load(return_symmetric=...,
dtype=...)` -> `ld_data / ld_indptr / leftmost_idx`; :186-191: `ld_mat.get_lambda_min(min_max_ratio=1e-3)`; store format:
docs/download_ld.md:6-10).  magenpy is neither in the reference tree nor in the image this repository was written in, so
the reader's parity is UNPINNED.  This tool closes the gap on any machine that has magenpy and one real
store:

    python tools/check_store.py /path/to/ld/chr_22 [--fixture tests/golden/magenpy_store_chr22.npz] [--rows 4096]

It compares, with `==` on integer / index data and on the float arrays,
  * `matrix/indptr` and `matrix/data` as this reader decodes them   vs  magenpy's `LDMatrix.from_path(...).load(...)`
    for the stored (upper-triangular, stored dtype) form, the float32-dequantised form and, where magenpy hands it out,
    the symmetric form against `viprs_amd.plan.LDPlan.from_upper(...).windows()` (needs a GPU; skipped without one);
  * `get_lambda_min()` and `get_lambda_min(min_max_ratio=1e-3)` under BOTH candidate formulas -- and says which one
    magenpy agrees with (then set `ZarrLDMatrix.lambda_min_formula` accordingly, viprs_amd/io/zarr_ld.py);
and writes a fixture (.npz: the first `--rows` rows of the store as magenpy loads them + magenpy's lambda_min values + the
store's attributes) that `tests/test_zarr_ld.py::test_reader_matches_magenpy_fixture` picks up when present.

Exit status 0 = everything compared equal; 1 = a difference (printed); 2 = magenpy not importable.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("store", help="one chromosome's LD store (a zarr group with matrix/data and matrix/indptr)")
    ap.add_argument("--fixture", default=None, help="where to write the .npz fixture (default: tests/golden/magenpy_store_<chrom>.npz)")
    ap.add_argument("--rows", type=int, default=4096, help="rows of the store kept in the fixture")
    ap.add_argument("--min-max-ratio", type=float, default=1e-3)
    args = ap.parse_args()

    try:
        import magenpy as mgp
    except Exception as e:                                       # noqa: BLE001
        print(f"magenpy is not importable here ({type(e).__name__}: {e}); this tool needs it", file=sys.stderr)
        return 2
    from viprs_amd.io import zarr_ld as Z

    ok = True

    def same(name, a, b):
        nonlocal ok
        a, b = np.asarray(a), np.asarray(b)
        eq = a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)
        print(f"  {'OK  ' if eq else 'DIFF'} {name}: ours {a.dtype}{a.shape} magenpy {b.dtype}{b.shape}"
              + ("" if eq or a.shape != b.shape else f"  ({int((a != b).sum())} entries differ)"))
        ok = ok and eq

    ours = Z.ZarrLDMatrix(args.store)
    theirs = mgp.LDMatrix.from_path(args.store)
    print(f"store {args.store}: {ours.n_snps} SNPs, stored dtype {ours.stored_dtype}, magenpy {getattr(mgp, '__version__', '?')}")

    fix = {"attrs_json": np.array(json.dumps(ours.attrs)), "magenpy_version": np.array(str(getattr(mgp, "__version__", "?")))}
    # ---- stored form + float32 form --------------------------------------------------------------------------------
    for tag, dtype in (("stored", None), ("float32", np.float32)):
        mine = ours.load(return_symmetric=False, dtype=dtype)
        kw = {"return_symmetric": False}
        if dtype is not None:
            kw["dtype"] = dtype
        theirs.load(**kw)
        t_data, t_ip, t_lb = theirs.ld_data, theirs.ld_indptr, theirs.leftmost_idx
        print(f"[{tag}]")
        same("ld_indptr", mine.ld_indptr.astype(np.int64), np.asarray(t_ip).astype(np.int64))
        same("leftmost_idx", mine.leftmost_idx.astype(np.int64), np.asarray(t_lb).astype(np.int64))
        same("ld_data", mine.ld_data, np.asarray(t_data))
        r = min(args.rows, ours.n_snps)
        n = int(np.asarray(t_ip)[r])
        fix[f"{tag}_indptr"] = np.asarray(t_ip)[:r + 1].copy()
        fix[f"{tag}_leftmost_idx"] = np.asarray(t_lb)[:r].copy()
        fix[f"{tag}_data"] = np.asarray(t_data)[:n].copy()
        theirs.release() if hasattr(theirs, "release") else None
    # ---- symmetric windows (device expansion vs magenpy's host expansion) ---------------------------------------------
    try:
        from viprs_amd import _lib
        have_gpu = _lib.device_count() > 0
    except Exception:                                            # noqa: BLE001
        have_gpu = False
    if have_gpu:
        from viprs_amd.plan import LDPlan
        up = ours.load()
        plan = LDPlan.from_upper(up.ld_indptr, up.ld_data)
        lb, ip = plan.windows()
        theirs.load(return_symmetric=True)
        print("[symmetric windows]")
        same("leftmost_idx", lb.astype(np.int64), np.asarray(theirs.leftmost_idx).astype(np.int64))
        same("ld_indptr", ip.astype(np.int64), np.asarray(theirs.ld_indptr).astype(np.int64))
        plan.close()
        theirs.release() if hasattr(theirs, "release") else None
    else:
        print("[symmetric windows] skipped: no HIP device")
    # ---- lambda_min ----------------------------------------------------------------------------------------------------
    print("[lambda_min]")
    r = args.min_max_ratio
    lm0 = float(theirs.get_lambda_min())
    lm_r = float(theirs.get_lambda_min(min_max_ratio=r))
    fix["lambda_min_r0"], fix["lambda_min_r"], fix["min_max_ratio"] = np.array(lm0), np.array(lm_r), np.array(r)
    mine0 = ours.get_lambda_min()
    print(f"  {'OK  ' if mine0 == lm0 else 'DIFF'} get_lambda_min(): ours {mine0!r} magenpy {lm0!r}")
    ok = ok and mine0 == lm0
    verdict = None
    for formula in ("one_plus_r", "one_minus_r"):
        v = ours.get_lambda_min(min_max_ratio=r, formula=formula)
        hit = (v == lm_r) or (abs(v - lm_r) <= 1e-12 * max(abs(lm_r), 1e-300))
        print(f"  {'MATCH' if hit else 'no   '} formula {formula}: ours {v!r} magenpy {lm_r!r}")
        if hit and verdict is None:
            verdict = formula
    fix["lambda_min_formula"] = np.array(verdict or "")
    if verdict is None:
        print("  NEITHER candidate formula reproduces magenpy's get_lambda_min(min_max_ratio): read magenpy's LDMatrix."
              "get_lambda_min and restate it in viprs_amd/io/zarr_ld.py")
        ok = False
    else:
        print(f"  => set ZarrLDMatrix.lambda_min_formula = {verdict!r}")

    chrom = ours.chromosome if ours.chromosome is not None else os.path.basename(os.path.normpath(args.store))
    out = args.fixture or os.path.join(ROOT, "tests", "golden", f"magenpy_store_{chrom}.npz")
    np.savez_compressed(out, **fix)
    print(f"fixture written: {out} ({os.path.getsize(out) / 1e6:.2f} MB)")
    print("RESULT:", "all equal" if ok else "DIFFERENCES FOUND")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
