"""A/B timing of library builds: interleaved rounds, one subprocess per (round, build); prints the dominant kernel's ms per launch.
usage: ab_time.py ROUNDS GEOM lib1.so lib2.so ...   (GEOM: T = N 4096 / hop 1280, D = 2048 / 256, S = 1024 / 256)"""
import ast, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, geom, libs = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "prof_lib.py"), l, "256", "400", "20", geom], capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print("FAILED", l, out.stderr[-400:]); continue
        tm = ast.literal_eval(line[-1])
        res[l].append(tm["dom_ms"] / tm["iter_launches"])
for l in libs:
    v = sorted(res[l])
    if v: print("%-28s ms/launch: min %.4f median %.4f max %.4f  (n=%d)" % (os.path.basename(l), v[0], v[len(v) // 2], v[-1], len(v)))
