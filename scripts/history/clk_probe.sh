#!/bin/bash
# effective clock of the dominant kernel for several builds of the library, same box, same call: GRBM_GUI_ACTIVE / 8 XCDs / kernel time
# usage: scripts/clk_probe.sh OUTDIR lib1.so lib2.so ...
OUT=$1; shift
mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$(pwd)
for lib in "$@"; do
  n=$(basename $lib .so)
  for rep in 1 2; do
    (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$ROOT/$OUT/$n.$rep" -- python3 "$ROOT/scripts/prof_lib.py" "$ROOT/$lib" 256 400 20 T > "$ROOT/$OUT/$n.$rep.log" 2>&1)
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$n.$rep.kt" -- python3 "$ROOT/scripts/prof_lib.py" "$ROOT/$lib" 256 400 20 T > "$ROOT/$OUT/$n.$rep.kt.log" 2>&1)
  done
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for lib in sys.argv[2:]:
    n = os.path.basename(lib)[:-3]
    for rep in (1, 2):
        g = w = 0.0; c = 0
        for f in glob.glob("%s/%s.%d/**/*counter_collection.csv" % (out, n, rep), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_gl_fused" in r["Kernel_Name"]:
                    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": g += float(r["Counter_Value"]); c += 1
                    if r["Counter_Name"] == "SQ_WAVE_CYCLES": w += float(r["Counter_Value"])
        t = []
        for f in glob.glob("%s/%s.%d.kt/**/*kernel_trace.csv" % (out, n, rep), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_gl_fused" in r["Kernel_Name"]: t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
        if c and t:
            ms = sum(t) / len(t)
            print("%-22s rep %d: kernel %.4f ms (trace, n=%d)  GRBM/8 = %.0f cycles/launch -> %.3f GHz if same duration; wave quad-cycles %.1f M" % (n, rep, ms, len(t), g / c / 8, g / c / 8 / ms / 1e6, w / c / 1e6))
PY
