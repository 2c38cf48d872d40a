import sys, numpy as np
sys.path.insert(0, '.')
exec(open('scratch/micro1.py').read().split("for math in")[0])
for math in ("exact",):
    for b in (1024, 2048, 3619, 6000):
        run([b], math)
    run([650]*1536, math)
