#!/usr/bin/env bash
# FETCH_SIZE / WRITE_SIZE / TCP accesses of the update kernel for a TSDF variant: bash tools/scratch/pmc_var.sh <tag>  (env passes through)
set -uo pipefail
TAG="$1"; export TMPDIR=/tmp
OUT="$PWD/gpurun_out/pmcv_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
PMC=(--no-cpu-baseline --no-rows --steps 1 --warmup 0 --frames-per-step 32 --resident-frames 32)
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -- python3 bench.py "${PMC[@]}" > "$OUT/b$i.log" 2>&1; echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tsdf_integrate_kernel<false" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:34s} mean {sum(v)/len(v):14.1f}  n={len(v)}")
PY
find "$OUT" -name "*.csv" -delete
