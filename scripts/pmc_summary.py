#!/usr/bin/env python3
"""Post-process the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into
profiles/rNN_pmc_summary.csv and profiles/rNN_pmc_traffic.json.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <windows_per_gpu> <outdir> [round prefix, default r02]"""
import csv, collections, json, sys, re

def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        agg[name].append(float(r["Counter_Value"]))
    return agg

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
W, out = int(sys.argv[3]), sys.argv[4]
RN = sys.argv[5] if len(sys.argv) > 5 else "r02"
names = sorted(set(fetch) | set(write))
with open(f"{out}/{RN}_pmc_summary.csv", "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,FETCH_SIZE_KB_max,WRITE_SIZE_KB_max\n")
    for n in names:
        fv, wv = fetch.get(n, [0.0]), write.get(n, [0.0])
        f.write(f"{n},{max(len(fv), len(wv))},{sum(fv)/len(fv):.3f},{sum(wv)/len(wv):.3f},{max(fv):.3f},{max(wv):.3f}\n")
traffic = {"windows_per_gpu": W,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes of `python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline --no-host-legs`; counter unit KB; MEAN over the launches (later iterations run fewer windows, like the "
                   "bench's achieved figure); hbm_bytes_per_launch = (FETCH_SIZE + WRITE_SIZE) * 1024.  MI355X_MICROARCH.md: "
                   "FETCH_SIZE is exact-by-half only for 16-B/lane streaming reads; these kernels read 8 B/lane (f64 elements), "
                   "a width the guide calls uncalibrated, so no factor is applied; WRITE_SIZE is taken as is"}
for n in names:
    base = n.split("<")[0] if not n.startswith("k_proj_linearize") else n
    if base in ("k_build_solve_sb", "k_build_solve_st", "k_schur_split", "k_schur_fold", "k_proj_linearize<0>", "k_sweep_mfma", "k_rank1_mfma", "k_dogleg", "k_proj_linearize<1>", "k_lin_gram", "k_step_control"):
        fv, wv = fetch.get(n, [0.0]), write.get(n, [0.0])
        key = base
        traffic[key] = {"FETCH_SIZE_KB_mean": sum(fv) / len(fv), "WRITE_SIZE_KB_mean": sum(wv) / len(wv),
                        "hbm_bytes_per_launch": (sum(fv) / len(fv) + sum(wv) / len(wv)) * 1024.0}
json.dump(traffic, open(f"{out}/{RN}_pmc_traffic.json", "w"), indent=1)
print(open(f"{out}/{RN}_pmc_summary.csv").read())
