import sys, time, numpy as np
sys.path.insert(0, '.')
from viprs_amd.data import ArrayDataLoader
from viprs_amd.model import VIPRS
from viprs_amd.utils import synthetic as syn
sizes = syn.block_sizes("cfg3")
t=time.time(); gdl = ArrayDataLoader.synthetic({1: list(sizes)}, forms=("symmetric",)); print("data", time.time()-t)
for resident in (True, False):
    t=time.time(); m = VIPRS(gdl, low_memory=False, device_resident=resident); print("init", time.time()-t)
    t=time.time(); m.fit(max_iter=30, theta_0={"pi":0.01, "sigma_epsilon":0.8}); dt=time.time()-t
    print("resident", resident, "fit", dt, "s", m.optim_result.nit, "iters ->", dt/max(1,m.optim_result.nit)*1e3, "ms/iter", m.optim_result.message, "ELBO", m.history["ELBO"][-1], "pi", m.pi, "h2", m.get_heritability())
    for p in m._plans.values(): p.close()
