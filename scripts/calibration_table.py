"""Reads calibrate_schedule.py outputs (one file per geometry x input class) and prints, per iteration count I and schedule k,
the worst error over all sets and the number of utterances beyond 1e-4 / 3e-5; then the smallest k per I with none beyond 3e-5.
usage: calibration_table.py DIR"""
import glob, os, re, sys
rows = {}
pat = re.compile(r"^(\w) (\w+)\s+I=\s*(\d+) k=\s*(\d+): n=(\d+) median (\S+) p99 (\S+) p99.9 (\S+) max (\S+) \| over 1e-4: (\d+)\s+over 3e-5: (\d+)\s+over 1e-5: (\d+)")
sets = set()
for f in sorted(glob.glob(os.path.join(sys.argv[1], "cal_*.txt"))):
    for line in open(f):
        m = pat.match(line)
        if m:
            g, c, I, k, n, med, p99, p999, mx, o4, o35, o5 = m.groups()
            rows.setdefault((int(I), int(k)), {})[(g, c)] = (int(n), float(mx), int(o4), int(o35), float(p999))
            sets.add((g, c))
sets = sorted(sets)
print("sets: " + ", ".join("%s/%s" % s for s in sets))
print("%4s %4s | %s | worst max | over 1e-4 | over 3e-5 | utterances" % ("I", "k", " ".join("%-9s" % ("%s/%s" % (g, c[:3])) for g, c in sets)))
best = {}
for (I, k) in sorted(rows):
    r = rows[(I, k)]
    if len(r) != len(sets):
        continue
    worst = max(v[1] for v in r.values()); o4 = sum(v[2] for v in r.values()); o35 = sum(v[3] for v in r.values()); n = sum(v[0] for v in r.values())
    print("%4d %4d | %s | %.2e | %9d | %9d | %d" % (I, k, " ".join("%.2e " % r[s][1] for s in sets), worst, o4, o35, n))
    if o35 == 0 and I not in best:
        best[I] = k
print("smallest calibrated k per I with no utterance beyond 3e-5 in any set:", best)
