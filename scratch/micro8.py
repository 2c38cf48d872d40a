import sys, numpy as np
sys.path.insert(0, '.')
exec(open('scratch/micro1.py').read().split("for math in")[0])
run([650]*1536, "exact", low_memory=True)
run([650]*1536, "exact", low_memory=False)
from viprs_amd.utils import synthetic as syn
s = syn.block_sizes("cfg3")
run(list(s), "exact", low_memory=True)
