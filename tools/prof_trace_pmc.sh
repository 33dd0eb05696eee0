#!/usr/bin/env bash
# usage (on the GPU box, from the repo root): bash tools_prof.sh <tag> [bench args...]
# Pass 1: kernel-trace + stats.  Passes 2..: PMC counters, one TCC group per pass (FETCH_SIZE takes 3 of the 4 TCC
# slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with tracing.
set -uo pipefail
TAG="$1"; shift
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/prof_${TAG}"
mkdir -p "$OUT"
SMALL=(--no-cpu-baseline --steps 1 --warmup 0 --frames-per-step 4 --resident-frames 4)
echo "[prof] trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_trace.log" 2>&1; echo "rc=$?"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  echo "[prof] pmc pass $i: $grp"
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 bench.py "${SMALL[@]}" > "$OUT/bench_pmc$i.log" 2>&1
  echo "rc=$?"
done
python3 tools_prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
