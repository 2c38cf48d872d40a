"""Prints the headline, the secondaries and the CPU baseline of a bench.py JSON line.
    python tools/show_bench.py bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"{d['metric']}: {d['value'] / 1e9:.3f} G {d['unit']}, {d['ms_per_step']:.3f} ms/step, kernel {r['kernel_ms_avg']:.3f} ms, "
      f"roofline frac {r['frac']:.3f}, traffic {r.get('traffic')}")
for s in d["config"].get("secondary") or []:
    if "value" not in s:                      # fit-iteration entries and the like: printed as they are
        print("  " + json.dumps(s)[:300])
        continue
    print(f"  {s['name'][:86]:86s} {s['value'] / 1e6:8.1f} M/s  kernel {s['kernel_ms_avg']:.3f} ms  frac {s['roofline_frac']:.3f}")
c = d.get("cpu_baseline") or {}
print(f"cpu_baseline: {c.get('value', 0) / 1e6:.1f} M/s ({c.get('kind')}, {c.get('cores')} cores); best exact: {c.get('best_exact')}")
