"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/traffic_refresh.sh.  usage: traffic_json.py DIR ROUND > traffic.json"""
import collections, csv, glob, json, os, sys
d, rnd = sys.argv[1], sys.argv[2]
KERNEL = {"fused": "k_gl_fused<", "fused_d": "k_gl_fused_d<", "persistent": "k_gl_persist<", "persistent_d": "k_gl_persist_d<"}
GEO = {"cfg3": (4096, 1280, 256 * 400, 1), "cfg4": (1024, 256, 256 * 400, 1), "cfg2": (4096, 1280, 342, 60)}   # N, hop, frames per launch, iterations per launch
ARGS = {"cfg3": "256 400 6 T", "cfg4": "256 400 6 S", "cfg2": "1 342 60 T"}
out = {}
for key in ("cfg3:fused", "cfg3:fused_d", "cfg4:fused", "cfg4:fused_d", "cfg2:persistent", "cfg2:persistent_d"):
    wl, kk = key.split(":")
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(os.path.join(d, "%s.%s" % (key, ctr), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr and r["Kernel_Name"].replace("void ", "").startswith(KERNEL[kk]):
                    v.append(float(r["Counter_Value"]))
        vals[ctr] = v
    if not vals["FETCH_SIZE"] or not vals["WRITE_SIZE"]:
        continue
    fetch, write = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]), sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
    N, hop, frames, iters = GEO[wl]
    esz = 8 if kk.endswith("_d") else 4
    alg = frames * iters * (2 * hop + N // 2 + 1) * esz
    e = {"hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)), "fetch_size_kb_raw": fetch, "write_size_kb_raw": write, "from": rnd,
         "launches_averaged": len(vals["FETCH_SIZE"]),
         "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as read.",
         "command": "GOMEL_FP64_ITERS=%s scripts/traffic_refresh.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 scripts/prof_run.py %s  [%s...]" % (
             "-1" if kk.endswith("_d") else "0", ARGS[wl], KERNEL[kk]),
         "algorithmic_bytes_per_launch": alg, "ratio": round((2 * fetch + write) * 1024 / alg, 3)}
    if kk.endswith("_d"):
        e["note"] = "algorithmic bytes in the kernel's own fp64 storage (2 x SURVEY 8(d)'s fp32 figure)"
    if kk.startswith("persistent"):
        e["note"] = e.get("note", "") + ("; " if e.get("note") else "") + "ONE launch = all 60 iterations of one 342-frame utterance; the exchange between frames is HBM traffic by design (write-through stores + gather loads), DESIGN.md 4.3b"
    out[key] = e
print(json.dumps(out, indent=1))
