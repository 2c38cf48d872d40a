import sys, numpy as np
sys.path.insert(0, '.')
exec(open('scratch/micro3.py').read().split("run2(s[s>=1536]")[0])
for k in (1, 2, 4, 8, 16, 32, 64):
    run2([3619]*k, f"3619 x{k}")
for k in (1, 8, 64, 256):
    run2([2048]*k, f"2048 x{k}")
for k in (1, 8, 64, 256):
    run2([1400]*k, f"1400 x{k}")
