import sys, numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from viprs_amd.utils import synthetic as syn
for sizes, lm in (([500], False), ([64], False), ([128], False), ([130], True)):
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=lm, seed=11)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0)
    got = H.run_hip(ld, inp, st0)
    print("== sizes", sizes, "low_memory", lm)
    for k in H.STATE:
        bad = np.nonzero(~(got[k] == ref[k]))[0]
        nan = np.nonzero(np.isnan(got[k]))[0]
        print(k, "nbad", len(bad), "first bad", bad[:8], "nnan", len(nan), "first nan", nan[:5])
        if len(bad):
            i = bad[0]; print("   got", got[k][i:i+4], "ref", ref[k][i:i+4])
