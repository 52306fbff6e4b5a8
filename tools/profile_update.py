#!/usr/bin/env python3
"""After tools/profile_round.sh and tools/profile_sq.sh: copy the summaries under gpurun_out/prof into profiles/ and
refresh profiles/traffic.json (HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024) and profiles/valu.json
(VALU / LDS wave-instructions per launch) from the counter passes.  usage: tools/profile_update.py [round tag, default r01]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def parse(path):
    out, k = {}, None
    for line in open(path):
        if not line.startswith(" "):
            k = line.strip()
        elif line.strip():
            name, val = line.split()[:2]
            out.setdefault(k, {})[name] = float(val)
    return out


def find(d, prefix):
    return next(v for k, v in d.items() if k.startswith(prefix))


f, w, q = (parse(os.path.join(SRC, n)) for n in ("pmc_fetch.txt", "pmc_write.txt", "pmc_sq.txt"))
kernels = {"force": "force_rows_kernel", "density": "density_record_kernel", "advect_hash": "advect_hash_kernel",
           "rank_reorder": "rank_reorder_kernel", "scatter": "scatter_bounds_kernel"}
t = json.load(open(os.path.join(DST, "traffic.json")))
for key, k in kernels.items():
    t["dam16m:" + key] = (2 * find(f, k)["FETCH_SIZE"] + find(w, k)["WRITE_SIZE"]) * 1024
json.dump(t, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
v = json.load(open(os.path.join(DST, "valu.json")))
for key in ("force", "density"):
    r = find(q, kernels[key])
    v["dam16m:" + key] = {"valu_wave_insts": r["SQ_INSTS_VALU"], "lds_wave_insts": r["SQ_INSTS_LDS"]}
json.dump(v, open(os.path.join(DST, "valu.json"), "w"), indent=1)
for src, dst in (("kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("pmc_fetch.txt", f"{tag}_pmc_fetch_size.txt"),
                 ("pmc_write.txt", f"{tag}_pmc_write_size.txt"), ("pmc_sq.txt", f"{tag}_pmc_sq.txt")):
    shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
print("profiles/ updated; now run bench.py and copy its line to profiles/%s_bench.json" % tag)
