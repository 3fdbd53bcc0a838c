"""profiles/issue.json from pmc_summary.py outputs: the vector-issue side of the iteration kernels (bench.py's roofline.flop_side quotes it).
usage: issue_json.py ROUND key=summary.txt:kernel_prefix ...   e.g.  issue_json.py r05 cfg3:fused_d=profiles/r05/pmc_fp64_cfg3_summary.txt:k_gl_fused_d
Per launch (mean over the dispatches of the pass): SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = the fraction of a resident wave's cycles in which it
issues a vector instruction (one wave per SIMD: = the SIMD's vector-issue utilisation); SQ_INSTS_VALU = vector instructions (per wave)."""
import ast
import json
import sys

rnd = sys.argv[1]
out = {}
for spec in sys.argv[2:]:
    key, rest = spec.split("=", 1)
    path, prefix = rest.rsplit(":", 1)
    c = {}
    for line in open(path):
        name, _, tail = line.partition(" {")
        if prefix + "<" not in name and not name.strip().endswith(prefix):
            continue
        c.update(ast.literal_eval("{" + tail.split("} dispatches")[0] + "}"))
    if "SQ_WAVE_CYCLES" not in c:
        continue
    out[key] = {"valu_issue_frac": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4), "sq_active_inst_valu": c["SQ_ACTIVE_INST_VALU"],
                "sq_wave_cycles": c["SQ_WAVE_CYCLES"], "sq_insts_valu": c.get("SQ_INSTS_VALU"), "sq_insts_lds": c.get("SQ_INSTS_LDS"),
                "sq_wait_any_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_ANY" in c else None,
                "sq_waves": c.get("SQ_WAVES"), "from": rnd, "source": path,
                "how": "rocprofv3 --pmc, one counter set per pass (scripts/pmc_passes.sh), mean per launch"}
print(json.dumps(out, indent=1))
