import sys, numpy as np
sys.path.insert(0, '.')
exec(open('scratch/micro3.py').read().split("run2(s[s>=1536]")[0])
import os
print("LARGE", os.environ.get("VIPRS_LARGE_BLOCK"), "MEDIUM", os.environ.get("VIPRS_MEDIUM_BLOCK"))
run2([3619]*8, "3619 x8")
run2([3619]*32, "3619 x32")
run2([2048]*256, "2048 x256")
