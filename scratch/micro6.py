import sys, os, numpy as np
sys.path.insert(0, '.')
exec(open('scratch/micro3.py').read().split("run2(s[s>=1536]")[0])
print("TEAM0", os.environ.get("VIPRS_TEAM0"), "TEAM1", os.environ.get("VIPRS_TEAM1"), "LARGE", os.environ.get("VIPRS_LARGE_BLOCK"), "MEDIUM", os.environ.get("VIPRS_MEDIUM_BLOCK"))
run2([3619]*8, "3619 x8")
run2([2000]*24, "2000 x24")
run2(s, "all")
