"""Duration of every launch of one kernel in a rocprofv3 kernel trace, in launch order, with the idle gap in front of it: what shows
the clock transient of the first call after an idle period.  usage: launch_index.py <kernel_trace.csv> <kernel name substring> [grid size]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"] and (len(sys.argv) < 4 or r["Grid_Size_X"] == sys.argv[3])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev, call, idx = None, 0, 0
print("# call launch duration_us gap_before_us   (%s, %d launches)" % (sys.argv[2], len(rows)))
tot = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    if prev and gap > 200.0:
        call, idx = call + 1, 0
    print("%d %2d %.1f %.1f" % (call, idx, (e - s) / 1e3, gap))
    tot.setdefault(call, []).append((e - s) / 1e3)
    prev, idx = e, idx + 1
for c, v in tot.items():
    print("# call %d: %d launches, mean %.1f us, min %.1f, max %.1f" % (c, len(v), sum(v) / len(v), min(v), max(v)))
