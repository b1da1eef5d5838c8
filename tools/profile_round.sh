#!/bin/bash
# Run on the GPU box: kernel-trace statistics and HBM-traffic counters of the default bench.py workload.
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_stats/, gpurun_out/<tag>_summary.txt
# The counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass) without any trace domain.
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-cases --no-traffic > $out/${tag}_stats.log 2>&1 || { echo "stats run failed"; tail -5 $out/${tag}_stats.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 $root/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-other-cases --no-traffic --no-graph > $out/${tag}_fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 $root/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-other-cases --no-traffic --no-graph > $out/${tag}_write.log 2>&1 || { echo "write pass failed"; exit 1; }
python3 - "$out" "$tag" <<'PY' | tee $out/${tag}_summary.txt
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]
print("== rocprofv3 --kernel-trace --stats: python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-cases --no-traffic")
for f in glob.glob(f"{out}/{tag}_stats/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows:
        if float(r['Percentage']) < 0.05:
            continue
        print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write"):
    for f in glob.glob(f"{out}/{tag}_{kind}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "pde_amd" in k:
                agg[k.split("(")[0].replace("void pde_amd::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== HBM-side traffic per launch (separate --pmc passes; FETCH_SIZE doubled: on gfx950 it tallies 128-B requests at 64 B)")
res = {}
for k, v in agg.items():
    fs = sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [1])), 1)
    ws = sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [1])), 1)
    rd, wr = 2 * fs * 1024, ws * 1024
    print(f"{k[:60]:60s} FETCH_SIZE {fs:12.1f} KB  WRITE_SIZE {ws:12.1f} KB  -> read {rd/1e6:8.1f} MB + write {wr/1e6:8.1f} MB = {(rd+wr)/1e6:8.1f} MB")
    res[k] = {"fetch_size_kb": fs, "write_size_kb": ws, "read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
json.dump(res, open(f"{out}/{tag}_traffic.json", "w"), indent=1)
PY
