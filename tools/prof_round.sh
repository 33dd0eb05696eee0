#!/usr/bin/env bash
# usage (GPU box, repo root):  bash tools/prof_round.sh r01
# 1. rocprofv3 --kernel-trace --stats of the default bench command (the roofline numbers must agree with its
#    average duration for tsdf_integrate_kernel);
# 2. separate --pmc passes (never combined with tracing; FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950)
#    on the same workload with one step of 32 frames;
# 3. summary + profiles-ready files under gpurun_out/prof_<tag>/ (copy what should be judged into profiles/).
set -uo pipefail
TAG="${1:-r01}"
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/prof_${TAG}"
rm -rf "$OUT"; mkdir -p "$OUT"
echo "[prof] kernel trace of: python3 bench.py --no-cpu-baseline"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_trace.log" 2>&1
echo "rc=$?"
PMC=(--no-cpu-baseline --steps 1 --warmup 0 --frames-per-step 32 --resident-frames 32)
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TA_TOTAL_WAVEFRONTS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  echo "[prof] pmc pass $i: $grp"
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 bench.py "${PMC[@]}" > "$OUT/bench_pmc$i.log" 2>&1
  echo "rc=$?"
done
# calibration pass: FETCH_SIZE of the free-space bricks alone (16-B-per-lane reads are tallied at half their bytes)
echo "[prof] pmc calibration pass: FETCH_SIZE, free-space bricks only"
TL3D_DEBUG_ONLY=2 timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_free" -- python3 bench.py "${PMC[@]}" > "$OUT/bench_pmc_free.log" 2>&1
echo "rc=$?"
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep what is judged (stats csv, summary, traffic json, bench logs); drop the raw per-dispatch csvs (hundreds of MB)
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
