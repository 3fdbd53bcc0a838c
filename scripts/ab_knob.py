"""A/B timing of ONE build under an environment knob read at gomel_create: interleaved rounds, one subprocess per (round, setting); prints the
fp64 fused kernel's ms per launch (config 3: 256 x 400 frames, fp64 throughout, 20 launches after GOMEL_RUNS warm calls).
usage: ab_knob.py ROUNDS KNOB value1 value2 ...      e.g.  ab_knob.py 5 GOMEL_FUSED_D2 0 1"""
import ast, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, knob, values = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
res = {v: [] for v in values}
for r in range(rounds):
    for v in values:
        env = dict(os.environ, GOMEL_FP64_ITERS="-1", GOMEL_RUNS="2", GOMEL_PATH="2")
        env[knob] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "prof_run.py"), "256", "400", "20", "T"], capture_output=True, text=True, env=env)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print("FAILED", knob, v, out.stderr[-400:]); continue
        tm = ast.literal_eval(line[-1])
        res[v].append(tm["dom64_ms"] / max(1, tm["iter64_launches"]))
for v in values:
    x = sorted(res[v])
    if x: print("%s=%-6s k_gl_fused_d* ms/launch: min %.4f median %.4f max %.4f  (n=%d)" % (knob, v, x[0], x[len(x) // 2], x[-1], len(x)))
