"""Development tool: the fit fixtures (trajectories captured from the reference) reproduced with math_mode='fast'."""
import sys, os, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_fit import build_model, FIT
for path in FIT:
    fx = np.load(path)
    if "float_precision" in fx and str(fx["float_precision"]) != "float32":
        continue
    for mode in ("exact", "fast"):
        model, theta = build_model(fx, e_step="hip", math_mode=mode)
        model.fit(max_iter=60, theta_0=theta)
        h, ref = np.array(model.history["ELBO"]), fx["elbo_history"]
        n = min(len(h), len(ref))
        dev = np.max(np.abs(h[:n] - ref[:n]) / np.maximum(np.abs(ref[:n]), 1e-300))
        pipd = max(np.max(np.abs(model.pip[c] - fx[f"pip_{c}"])) for c in model.pip)
        pmd = max(np.max(np.abs(model.post_mean_beta[c] - fx[f"post_mean_beta_{c}"]) / (np.abs(fx[f"post_mean_beta_{c}"]) + 2e-7 / 2e-3)) for c in model.pip)
        print(f"{os.path.basename(path)[:-4]:28s} {mode}: nit {model.optim_result.nit} (ref {int(fx['nit'])}) ELBO rel dev {dev:.2e} "
              f"(abs {np.max(np.abs(h[:n]-ref[:n])):.3g}) pip max abs dev {pipd:.2e} post_mean rel {pmd:.2e} "
              f"sigma_eps rel {abs(float(model.sigma_epsilon)/float(fx['final_sigma_epsilon'])-1):.1e} msg={model.optim_result.message == str(fx['message'])}", flush=True)
