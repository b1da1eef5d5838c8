"""Post-processing of tools/case_profile.sh: per residual kernel of one bench_all case, the rocprofv3 average duration, the
algorithmic bytes (4 E_in for a forward launch, 8 E_in for an adjoint launch: read every field once, write every gradient
once), the achieved rate against the 8 TB/s HBM peak, and the HBM-side traffic from the counters (2 x FETCH_SIZE +
WRITE_SIZE in KB: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes)."""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, filt = sys.argv[1], sys.argv[2]
out = os.path.join(root, "gpurun_out")
sys.path.insert(0, root)
import re
src = open(os.path.join(root, "tools", "bench_all.py")).read()
case = None
for m in re.finditer(r'\("([^"]+)", "(\w+)", dict\(([^)]*)\), \(([\d, ]+)\)', src):
    if filt.lower() in m.group(1).lower():
        case = m
        break
shape = [int(v) for v in case.group(4).split(",")]
log = open(os.path.join(out, f"{tag}_stats.log")).read()
nf = int(re.search(r"F=(\d+)", log).group(1))
e_in = nf
for v in shape:
    e_in *= v
line = [ln for ln in log.splitlines() if "fwd+bwd" in ln][-1]
dur = {}
for f in glob.glob(f"{out}/{tag}_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "pde_amd" in r["Name"]:
            dur[r["Name"].split("(")[0].replace("void pde_amd::", "")] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write"):
    for f in glob.glob(f"{out}/{tag}_{kind}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "pde_amd" in k:
                agg[k.split("(")[0].replace("void pde_amd::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"case": case.group(1), "class": case.group(2), "shape": shape, "fields": nf, "E_in": e_in, "step": line.strip(), "kernels": {}}
print(line.strip())
print(f"{'kernel':62s} {'calls':>5s} {'avg_us':>9s} {'alg_MB':>8s} {'TB/s':>6s} {'frac':>5s} {'traffic_MB':>10s} {'ratio':>5s}")
for k, (us, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
    if not any(s in k for s in ("fwd_", "bwd_")):
        continue
    alg = (8 if k.startswith("bwd_") else 4) * e_in
    fs = agg[k].get("FETCH_SIZE", [])
    ws = agg[k].get("WRITE_SIZE", [])
    traffic = (2 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024 if fs and ws else None
    rate = alg / (us * 1e-6) / 1e12
    res["kernels"][k] = {"avg_us": us, "calls": calls, "algorithmic_bytes": alg, "achieved_TBps": rate, "frac_of_8TBps": rate / 8,
                         "traffic_bytes": traffic, "traffic_over_algorithmic": traffic / alg if traffic else None}
    print(f"{k[:62]:62s} {calls:5d} {us:9.1f} {alg / 1e6:8.1f} {rate:6.2f} {rate / 8:5.2f} {traffic / 1e6 if traffic else float('nan'):10.1f} {traffic / alg if traffic else float('nan'):5.2f}")
json.dump(res, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
