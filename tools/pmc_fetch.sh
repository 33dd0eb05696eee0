#!/usr/bin/env bash
# usage (GPU box): bash tools/pmc_fetch.sh <tag>   -- one FETCH_SIZE / WRITE_SIZE pass pair on the 32-frame PMC workload; env passes through
set -uo pipefail
TAG="$1"; export TMPDIR=/tmp
OUT="$PWD/gpurun_out/pmcf_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
PMC=(--no-cpu-baseline --steps 1 --warmup 0 --frames-per-step 32 --resident-frames 32)
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/p1" -- python3 bench.py "${PMC[@]}" > "$OUT/b1.log" 2>&1; echo "rc=$?"
timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/p2" -- python3 bench.py "${PMC[@]}" > "$OUT/b2.log" 2>&1; echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tsdf_integrate_kernel<false" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:16s} mean {sum(v)/len(v):12.1f}  n={len(v)}")
PY
find "$OUT" -name "*.csv" -delete
