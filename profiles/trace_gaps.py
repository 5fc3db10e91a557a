"""Per-step GPU timeline from a rocprofv3 kernel trace: kernel durations and idle gaps."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
# keep the steady-state region: launches whose grid matches the small config
sel = [r for r in rows if "agx::" in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) in (grid,)]
sel = sel[len(sel) // 2:]  # second half = timed region
names = defaultdict(list)
gaps = []
for a, b in zip(sel, sel[1:]):
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in sel:
    names[r["Kernel_Name"].split("(")[0][-60:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in names.items():
    print(f"{k:62s} n={len(v):5d} avg {sum(v)/len(v)/1e3:7.2f} us  min {min(v)/1e3:6.2f}")
gaps.sort()
print("gaps between consecutive agx kernels: median %.2f us, mean %.2f us, p90 %.2f us" % (gaps[len(gaps)//2]/1e3, sum(gaps)/len(gaps)/1e3, gaps[int(len(gaps)*0.9)]/1e3))
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
print("span %.1f us, busy %.1f us (%.0f%%), kernels %d" % (span/1e3, busy/1e3, 100*busy/span, len(sel)))
