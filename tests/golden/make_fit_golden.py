#!/usr/bin/env python3
"""Generates tests/golden/fit_*.npz by importing the REFERENCE's own Python layer (VIPRS.fit(),
e_step(), m_step(), elbo(); VIPRSMix) from /root/reference in the authoring container.

The reference cannot be imported as shipped here because (i) its compiled extension
`viprs.model.vi.e_step_cpp` is not built in the read-only tree and (ii) its data layer `magenpy`
(un-vendored PyPI dependency, 0.2.0) is not installed.  This script therefore registers, IN MEMORY
ONLY (nothing is written anywhere):
  * `viprs.model.vi.e_step_cpp` -> thin forwarders to oracle/_ref/libviprs_ref.so, i.e. the
    reference's own e_step.hpp compiled where it lies (same kernels the Cython module wraps);
  * a minimal `magenpy` namespace: the `GWADataLoader` class object (isinstance check only),
    `utils.compute_utils.is_numeric`, `stats.h2.ldsc.simple_ldsc` (raises: theta_0 is always given).
The data loader handed to the reference is a plain array-backed object (tests/golden/_refdata.py
semantics are restated below); hyper-parameters start from a fixed theta_0, so no RNG enters.

Only the resulting ARRAYS are committed (inputs + per-iteration history + final posterior); neither
reference source nor these stubs travel with the fixtures.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O                      # noqa: E402
from viprs_amd.utils import synthetic as syn        # noqa: E402


def install_stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    mg = mod("magenpy")

    class GWADataLoader:      # isinstance() target only
        pass

    mg.GWADataLoader = GWADataLoader
    mod("magenpy.utils")
    cu = mod("magenpy.utils.compute_utils")
    cu.is_numeric = lambda x: isinstance(x, (int, float, np.number, np.ndarray))
    mod("magenpy.stats")
    mod("magenpy.stats.h2")
    ldsc = mod("magenpy.stats.h2.ldsc")

    def simple_ldsc(*a, **k):
        raise RuntimeError("simple_ldsc is not available in the golden generator")
    ldsc.simple_ldsc = simple_ldsc
    mgu = mod("magenpy.utils.system_utils")
    mgu.makedir = lambda *a, **k: None
    mp = mod("magenpy.utils.model_utils")
    mp.merge_snp_tables = lambda *a, **k: None

    es = mod("viprs.model.vi.e_step_cpp")
    es.cpp_e_step = lambda *a: O.cpp_e_step(*a, kind="reference")
    es.cpp_e_step_mixture = lambda *a: O.cpp_e_step_mixture(*a, kind="reference")
    es.cpp_e_step_grid = lambda *a: O.cpp_e_step_grid(*a, kind="reference")
    es.check_blas_support = O.check_blas_support
    es.check_omp_support = O.check_omp_support
    return GWADataLoader


def make_loader(GWADataLoader, chrom_sizes, ld_dtype, seed, ld_kind="ar1", n_of=None, h2_of=None):
    """Array-backed loader with the attributes VIPRS.__init__/BayesPRSModel.__init__ read
    (VIPRS.py:153-191, BayesPRSModel.py:59-142)."""
    lds, sss, inputs = {}, {}, {}
    for ci, (chrom, sizes) in enumerate(chrom_sizes.items()):
        ld_sym = syn.make_ld(sizes, low_memory=False, ld_dtype=ld_dtype, seed=seed + ci, kind=ld_kind)
        ld_up = syn.make_ld(sizes, low_memory=True, ld_dtype=ld_dtype, seed=seed + ci, kind=ld_kind)
        ss_kw = dict(n=(n_of or {}).get(chrom, 1e5), h2=(h2_of or {}).get(chrom, 0.2))
        ss = syn.make_sumstats(ld_sym, seed=seed + ci, **ss_kw)
        # marginal effects of a second cohort on the same LD (same causal effects, independent noise): what
        # BayesPRSModel.pseudo_validate() scores the fit against (BayesPRSModel.py:184-187, 397-410)
        ss.validation_std_beta = syn.make_sumstats(ld_sym, seed=seed + ci, noise_seed=seed + ci + 5000, **ss_kw).std_beta

        class LOP:
            def __init__(self, l):
                self.ld_data, self.ld_indptr, self.leftmost_idx = l.ld_data, l.ld_indptr, l.ld_left_bound

        class LDM:
            stored_dtype = np.dtype(ld_dtype)

            def __init__(self, s, u):
                self._s, self._u = s, u

            def load(self, return_symmetric=False, dtype=None):
                l = self._s if return_symmetric else self._u
                if dtype is not None and np.dtype(dtype) != l.ld_data.dtype:
                    # dequantise at load, as magenpy does when dequantize_on_the_fly=False
                    scale = l.dq_scale
                    l = syn.SyntheticLD(l.ld_left_bound, l.ld_indptr, (l.ld_data * scale).astype(dtype),
                                        l.block_start, l.rho, l.low_memory, 1.0)
                return LOP(l)

            def get_lambda_min(self, min_max_ratio=1e-3):
                return 0.0

        class SS:
            def __init__(self, s):
                self.n_per_snp = s.n_per_snp
                self._b = s.std_beta

            def get_snp_pseudo_corr(self):
                return self._b

        lds[chrom], sss[chrom] = LDM(ld_sym, ld_up), SS(ss)
        inputs[chrom] = (ld_sym, ld_up, ss)

    gdl = GWADataLoader()
    gdl.ld, gdl.sumstats_table, gdl.genotype = lds, sss, None
    gdl.shapes = {c: int(sum(s)) for c, s in chrom_sizes.items()}
    gdl.m = int(sum(gdl.shapes.values()))
    gdl.n = 1e5
    gdl.get_ld_matrices = lambda: lds
    return gdl, inputs


def main():
    assert O.have_reference()
    GWADataLoader = install_stubs()
    sys.path.insert(0, "/root/reference")
    import viprs                                    # the reference, imported where it lies
    from viprs.model.VIPRS import VIPRS
    from viprs.model.VIPRSMix import VIPRSMix
    print("reference viprs", viprs.__version__)

    cases = [
        ("fit_ss_1chr_upper", VIPRS, {22: [300, 250, 350]}, dict(low_memory=True), {}),
        ("fit_ss_1chr_sym", VIPRS, {22: [300, 250, 350]}, dict(low_memory=False), {}),
        ("fit_ss_2chr_upper", VIPRS, {21: [200, 180], 22: [150, 330]}, dict(low_memory=True), {}),
        ("fit_ss_fixed_sigma", VIPRS, {22: [400, 260]}, dict(low_memory=True, fix_params={"sigma_epsilon": 0.85}), {}),
        ("fit_mix_k4_upper", VIPRSMix, {22: [300, 250, 350]}, dict(low_memory=True, K=4), {}),
        # round 3: LD whose far field matters (long-range, non-Toeplitz blocks), int8-quantised and dequantised
        # on the fly -- the reference's published store format and its default LD form; the fixture carries the
        # int8 upper-triangular LD itself
        ("fit_ss_lr_int8_upper", VIPRS, {22: [420, 300, 180]}, dict(low_memory=True, dequantize_on_the_fly=True),
         dict(ld_kind="longrange", ld_dtype=np.int8)),
        ("fit_mix_k4_lr_int8_sym", VIPRSMix, {22: [380, 150, 330]}, dict(low_memory=False, K=4, dequantize_on_the_fly=True),
         dict(ld_kind="longrange", ld_dtype=np.int8)),
        # float_precision='float64' (VIPRS.py:72): the double state end to end, both LD forms
        ("fit_ss_f64_lr_int8_upper", VIPRS, {22: [420, 300, 180]},
         dict(low_memory=True, dequantize_on_the_fly=True, float_precision="float64"), dict(ld_kind="longrange", ld_dtype=np.int8)),
        ("fit_ss_f64_lr_int8_sym", VIPRS, {21: [260, 330], 22: [150, 200]},
         dict(low_memory=False, dequantize_on_the_fly=True, float_precision="float64"), dict(ld_kind="longrange", ld_dtype=np.int8)),
        ("fit_mix_k4_f64_lr_int8_upper", VIPRSMix, {22: [380, 150, 330]},
         dict(low_memory=True, K=4, dequantize_on_the_fly=True, float_precision="float64"), dict(ld_kind="longrange", ld_dtype=np.int8)),
    ]
    for name, cls, chrom_sizes, kw, ld_kw in cases:
        if ONLY and name not in ONLY:
            continue
        fit_kw = {}
        ld_dtype, ld_kind = ld_kw.get("ld_dtype", np.float32), ld_kw.get("ld_kind", "ar1")
        gdl, inputs = make_loader(GWADataLoader, chrom_sizes, ld_dtype, seed=301, ld_kind=ld_kind)
        theta_0 = {"pi": 0.01, "sigma_epsilon": 0.8}
        if "K" in kw:                                   # no RNG: explicit mixing proportions
            theta_0 = {"pis": 0.01 * np.array([0.4, 0.3, 0.2, 0.1]), "sigma_epsilon": 0.8}
        model = cls(gdl, **kw)
        model.fit(max_iter=60, theta_0=dict(theta_0), disable_pbar=True, **fit_kw)
        model.validation_std_beta = {c: inputs[c][2].validation_std_beta for c in chrom_sizes}
        out = dict(pseudo_r2=np.float64(model.pseudo_validate()),        # BayesPRSModel.py:397-410
                   ld_kind=ld_kind, dequantize_on_the_fly=bool(kw.get("dequantize_on_the_fly", False)),
                   float_precision=str(kw.get("float_precision", "float32")),
                   theta0_pi=0.01, theta0_sigma_epsilon=0.8, n=gdl.n, low_memory=kw.get("low_memory", True),
                   K=kw.get("K", 0), theta0_pis=np.asarray(theta_0.get("pis", [])), fix_sigma_epsilon=kw.get("fix_params", {}).get("sigma_epsilon", np.nan),
                   chroms=np.array(sorted(chrom_sizes)),
                   elbo_history=np.array(model.history["ELBO"], dtype=np.float64),
                   nit=model.optim_result.nit, success=bool(model.optim_result.success),
                   message=str(model.optim_result.message),
                   final_pi=np.asarray(model.pi, dtype=np.float64), final_tau_beta=np.asarray(model.tau_beta, dtype=np.float64),
                   final_sigma_epsilon=np.float64(model.sigma_epsilon), final_sigma_g=np.float64(model._sigma_g))
        for c in sorted(chrom_sizes):
            ld_sym, ld_up, ss = inputs[c]
            out[f"sizes_{c}"] = np.array(chrom_sizes[c])
            out[f"std_beta_{c}"] = ss.std_beta
            out[f"n_per_snp_{c}"] = ss.n_per_snp
            out[f"rho_{c}"] = ld_sym.rho
            out[f"validation_std_beta_{c}"] = ss.validation_std_beta
            if ld_kind != "ar1":                     # AR(1) LD is rebuilt from rho; anything else travels
                out[f"ld_upper_indptr_{c}"] = ld_up.ld_indptr
                out[f"ld_upper_data_{c}"] = ld_up.ld_data
            out[f"pip_{c}"] = model.pip[c]
            out[f"post_mean_beta_{c}"] = model.post_mean_beta[c]
            out[f"post_var_beta_{c}"] = model.post_var_beta[c]
            out[f"q_{c}"] = model.q[c]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "pseudo R2", out["pseudo_r2"], "nit", model.optim_result.nit, model.optim_result.message, "ELBO", model.history["ELBO"][-1],
              "pi", model.pi, "sig_eps", model.sigma_epsilon)


def grid_cases():
    """VIPRSGrid (serial grid fits through VIPRS.fit, pathwise and independent) + HyperparameterGrid."""
    GWADataLoader = sys.modules["magenpy"].GWADataLoader
    from viprs.model.gridsearch.HyperparameterGrid import HyperparameterGrid
    from viprs.model.gridsearch.VIPRSGrid import VIPRSGrid
    chrom_sizes = {22: [300, 250, 350]}
    for pathwise in (True, False):
        gdl, inputs = make_loader(GWADataLoader, chrom_sizes, np.float32, seed=401)
        grid = HyperparameterGrid(sigma_epsilon_steps=2, pi_steps=3, n_snps=gdl.m, h2_est=0.2, h2_se=0.1)
        model = VIPRSGrid(gdl, grid, low_memory=True)
        model.fit(pathwise=pathwise, max_iter=80, disable_pbar=True)
        model.validation_std_beta = {c: inputs[c][2].validation_std_beta for c in chrom_sizes}
        pseudo_r2 = np.asarray(model.pseudo_validate(), dtype=np.float64)     # one value per grid model
        vr = model.validation_result
        name = "fitgrid_pathwise" if pathwise else "fitgrid_independent"
        out = dict(pseudo_r2=pseudo_r2, n=gdl.n, pathwise=pathwise, chroms=np.array([22]), grid_sigma_epsilon=vr["sigma_epsilon"].to_numpy(),
                   grid_pi=vr["pi"].to_numpy(), elbo=vr["ELBO"].to_numpy().astype(np.float64),
                   converged=vr["Converged"].to_numpy(), messages=np.array(list(vr["Optimization_message"])),
                   nit=np.array([r.nit for r in model.optim_results]), tau_beta=np.asarray(model.tau_beta, dtype=np.float64),
                   sigma_g=np.asarray(model._sigma_g, dtype=np.float64))
        for c in (22,):
            ld_sym, ld_up, ss = inputs[c]
            out[f"sizes_{c}"] = np.array(chrom_sizes[c])
            out[f"std_beta_{c}"] = ss.std_beta
            out[f"n_per_snp_{c}"] = ss.n_per_snp
            out[f"rho_{c}"] = ld_sym.rho
            out[f"validation_std_beta_{c}"] = ss.validation_std_beta
            out[f"pip_{c}"] = model.pip[c]
            out[f"post_mean_beta_{c}"] = model.post_mean_beta[c]
            out[f"post_var_beta_{c}"] = model.post_var_beta[c]
            out[f"q_{c}"] = model.q[c]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "pseudo R2", pseudo_r2, "ELBO", out["elbo"], "nit", out["nit"], out["messages"][:2])


def sub_loader(GWADataLoader, gdl, c):
    """What `gdl.split_by_chromosome()[c]` hands to a per-chromosome fit (bin/viprs_fit:232-238): a loader that holds
    chromosome c only."""
    sub = GWADataLoader()
    sub.ld, sub.sumstats_table, sub.genotype = {c: gdl.ld[c]}, {c: gdl.sumstats_table[c]}, None
    sub.shapes = {c: gdl.shapes[c]}
    sub.m = int(gdl.shapes[c])
    sub.n = float(np.max(gdl.sumstats_table[c].n_per_snp))
    sub.get_ld_matrices = lambda: sub.ld
    return sub


def per_chromosome_cases():
    """The reference's DEFAULT mode: one independent VIPRS model per chromosome (bin/viprs_fit:232-238, :1079-1086).  Every
    chromosome is fitted by the reference ON ITS OWN loader; the fixture carries each fit's trajectory and posterior.
    Chromosomes of different size, sample size and heritability, so that the fits stop at different iterations."""
    GWADataLoader = sys.modules["magenpy"].GWADataLoader
    from viprs.model.VIPRS import VIPRS
    cases = [
        ("fitchr_ss_4chr_upper", {19: [260, 140], 20: [330, 200, 90], 21: [180], 22: [150, 300]}, dict(low_memory=True), {},
         dict(n_of={19: 1e5, 20: 4e4, 21: 2e5, 22: 1e5}, h2_of={19: 0.3, 20: 0.1, 21: 0.05, 22: 0.2})),
        ("fitchr_ss_3chr_lr_int8_upper", {20: [420, 130], 21: [300, 180], 22: [250]}, dict(low_memory=True, dequantize_on_the_fly=True),
         dict(ld_kind="longrange", ld_dtype=np.int8), dict(n_of={20: 1e5, 21: 5e4, 22: 1e5}, h2_of={20: 0.2, 21: 0.3, 22: 0.08})),
        ("fitchr_ss_2chr_sym_fixed_sigma", {21: [260, 200], 22: [310]}, dict(low_memory=False, fix_params={"sigma_epsilon": 0.9}), {},
         dict(n_of={21: 1e5, 22: 6e4})),
        # VIPRSMix per chromosome (the same fan-out, whatever the model class: bin/viprs_fit:1079-1086)
        ("fitchr_mix_k4_3chr_upper", {20: [280, 160], 21: [350], 22: [200, 120, 90]}, dict(low_memory=True, K=4), {},
         dict(n_of={20: 1e5, 21: 5e4, 22: 2e5}, h2_of={20: 0.25, 21: 0.1, 22: 0.15})),
    ]
    from viprs.model.VIPRSMix import VIPRSMix
    for name, chrom_sizes, kw, ld_kw, data_kw in cases:
        if ONLY and name not in ONLY:
            continue
        ld_dtype, ld_kind = ld_kw.get("ld_dtype", np.float32), ld_kw.get("ld_kind", "ar1")
        gdl, inputs = make_loader(GWADataLoader, chrom_sizes, ld_dtype, seed=611, ld_kind=ld_kind, **data_kw)
        theta_0 = {"pi": 0.01, "sigma_epsilon": 0.8}
        cls = VIPRS
        if "K" in kw:                                   # no RNG: explicit mixing proportions
            theta_0, cls = {"pis": 0.01 * np.array([0.4, 0.3, 0.2, 0.1]), "sigma_epsilon": 0.8}, VIPRSMix
        out = dict(ld_kind=ld_kind, dequantize_on_the_fly=bool(kw.get("dequantize_on_the_fly", False)),
                   float_precision=str(kw.get("float_precision", "float32")), theta0_pi=0.01, theta0_sigma_epsilon=0.8,
                   low_memory=kw.get("low_memory", True), K=kw.get("K", 0), theta0_pis=np.asarray(theta_0.get("pis", [])),
                   fix_sigma_epsilon=kw.get("fix_params", {}).get("sigma_epsilon", np.nan), chroms=np.array(sorted(chrom_sizes)))
        for c in sorted(chrom_sizes):
            model = cls(sub_loader(GWADataLoader, gdl, c), **{k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()})
            model.fit(max_iter=100, theta_0=dict(theta_0), disable_pbar=True)
            model.validation_std_beta = {c: inputs[c][2].validation_std_beta}
            ld_sym, ld_up, ss = inputs[c]
            out.update({
                f"n_{c}": model.n, f"elbo_history_{c}": np.array(model.history["ELBO"], dtype=np.float64),
                f"nit_{c}": model.optim_result.nit, f"success_{c}": bool(model.optim_result.success),
                f"message_{c}": str(model.optim_result.message), f"final_pi_{c}": np.asarray(model.pi, dtype=np.float64),
                f"final_tau_beta_{c}": np.asarray(model.tau_beta, dtype=np.float64), f"final_sigma_epsilon_{c}": np.float64(model.sigma_epsilon),
                f"final_sigma_g_{c}": np.float64(model._sigma_g), f"pseudo_r2_{c}": np.float64(model.pseudo_validate()),
                f"sizes_{c}": np.array(chrom_sizes[c]), f"std_beta_{c}": ss.std_beta, f"n_per_snp_{c}": ss.n_per_snp,
                f"rho_{c}": ld_sym.rho, f"validation_std_beta_{c}": ss.validation_std_beta, f"pip_{c}": model.pip[c],
                f"post_mean_beta_{c}": model.post_mean_beta[c], f"post_var_beta_{c}": model.post_var_beta[c], f"q_{c}": model.q[c]})
            if ld_kind != "ar1":
                out[f"ld_upper_indptr_{c}"] = ld_up.ld_indptr
                out[f"ld_upper_data_{c}"] = ld_up.ld_data
            print(name, "chr", c, "nit", model.optim_result.nit, model.optim_result.message, "ELBO", model.history["ELBO"][-1],
                  "pi", model.pi, "sig_eps", model.sigma_epsilon)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


ONLY = sys.argv[1:]          # e.g. `make_fit_golden.py fit_ss_lr_int8_upper`: regenerate the named fixtures only

if __name__ == "__main__":
    main()
    if not ONLY or any(n.startswith("fitgrid") for n in ONLY):
        grid_cases()
    if not ONLY or any(n.startswith("fitchr") for n in ONLY):
        per_chromosome_cases()
