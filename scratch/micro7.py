import sys, os, numpy as np
sys.path.insert(0, '.')
exec(open('scratch/micro3.py').read().split("run2(s[s>=1536]")[0])
tag = " ".join(f"{k[6:]}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("VIPRS_"))
run2(s, tag[:28])
