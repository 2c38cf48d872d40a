"""Per-sweep timeline of the panel kernels from a rocprofv3 kernel trace: for the last sweeps, every estep kernel's
start offset and duration (us) relative to the sweep's first kernel.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary
    python tools/class_timeline.py DIR [n_sweeps]"""
import csv, glob, os, sys
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sweeps, cur = [], []
for r in rows:
    name = r["Kernel_Name"]
    if "sweep_prologue_kernel" in name:
        if cur:
            sweeps.append(cur)
        cur = []
    if "estep" in name or "commit_team" in name or "stream_delay" in name or "sweep_prologue" in name:
        cur.append(r)
if cur:
    sweeps.append(cur)
def short(n):
    n = n.replace("void viprs::", "").replace("viprs::", "")
    return n[:100]
for sw in sweeps[-n_last:]:
    t0 = min(int(r["Start_Timestamp"]) for r in sw)
    t1 = max(int(r["End_Timestamp"]) for r in sw)
    print(f"---- sweep: {(t1 - t0) / 1e3:.1f} us")
    for r in sw:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?'):>8} wg {r.get('Workgroup_Size', '?'):>4} "
              f"vgpr {r.get('VGPR_Count', '?'):>4} lds {r.get('LDS_Block_Size', '?'):>7}  {short(r['Kernel_Name'])}")
