"""Durations of the band reduction's kernels as a function of their position in the run (rocprofv3 --kernel-trace CSV):
   python tools/kernel_profile_by_order.py kernel_trace.csv NAME [bins]
prints, per bin of consecutive dispatches of the LAST step, the mean duration in us."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
steps = 3
per = len(rows) // steps
sel = rows[-per:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
print(f"{sys.argv[2]}: {len(rows)} dispatches, {per} per step, step total {sum(d) / 1e3:.2f} ms")
for b in range(nb):
    seg = d[b * per // nb:(b + 1) * per // nb]
    if seg:
        print(f"  {b * per // nb:5d}..{(b + 1) * per // nb:5d}: mean {sum(seg) / len(seg):8.1f} us  max {max(seg):8.1f}")
