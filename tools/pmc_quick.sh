#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/pmc_quick.sh <tag> [extra env assignments for the experiments flavour]
# Separate rocprofv3 --pmc passes (never combined with tracing) over ONE step of 64 frames of the headline bench; prints the
# per-dispatch mean of every counter for tsdf_update_kernel.
set -uo pipefail
TAG="${1:-pmc}"; shift || true
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/pmc_${TAG}"; rm -rf "$OUT"; mkdir -p "$OUT"
ARGS=(--no-cpu-baseline --no-rows --no-single --steps 1 --warmup 0 --frames-per-step 64 --resident-frames 512)
i=0
GROUPS_DEFAULT="FETCH_SIZE;WRITE_SIZE TCC_HIT_sum TCC_MISS_sum;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS;GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum;TCP_TOTAL_CACHE_ACCESSES_sum TA_TOTAL_WAVEFRONTS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
IFS=';' read -ra GRPS <<< "${PMC_GROUPS:-$GROUPS_DEFAULT}"      # PMC_GROUPS="A B;C D": counter groups, one pass each
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  env "$@" timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -- python3 bench.py "${ARGS[@]}" > "$OUT/bench$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "tl3d" not in k: continue
        agg[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k in sorted(agg):
        print(k, file=fh)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"    {c:36s} mean {sum(v)/len(v):16.1f}  max {max(v):16.1f}  (n={len(v)})", file=fh)
print(open(os.path.join(out, "summary.txt")).read())
PY
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
