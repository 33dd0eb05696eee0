#!/usr/bin/env bash
# usage (GPU box, repo root):  bash tools/prof_round.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench command (the roofline numbers must agree with its
#    average duration for tsdf_integrate_kernel);
# 2. separate --pmc passes (never combined with tracing; FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950)
#    on the same workload with one step of 64 frames (the first 64 of the 512-frame orbit), for the default build (free-space bricks counted) and for
#    TL3D_FREE_COUNTERS=0 (free-space bricks streamed, the round-1 formulation);
# 3. calibration passes: FETCH_SIZE on access patterns with a KNOWN byte count (TL3D_TSDF_VARIANT=2 reads every record of
#    every listed brick with 8 B per lane; TL3D_DEBUG_ONLY=2 streams the free-space bricks alone with 16 B per lane);
# 4. summary + profiles-ready files under gpurun_out/prof_<tag>/ (copy what should be judged into profiles/).
set -uo pipefail
TAG="${1:-r02}"
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/prof_${TAG}"
rm -rf "$OUT"; mkdir -p "$OUT"
echo "[prof] kernel trace of: python3 bench.py --no-cpu-baseline --no-rows"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline --no-rows > "$OUT/bench_trace.log" 2>&1
echo "rc=$?"
echo "[prof] kernel trace of the per-row measurements: python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 (rows on)"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_rows" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > "$OUT/bench_trace_rows.log" 2>&1
echo "rc=$?"
PMC=(--no-cpu-baseline --no-rows --steps 1 --warmup 0 --frames-per-step 64 --resident-frames 512)      # 64 frames, 0.7 degrees apart
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TA_TOTAL_WAVEFRONTS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  echo "[prof] pmc pass $i: $grp"
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 bench.py "${PMC[@]}" > "$OUT/bench_pmc$i.log" 2>&1
  echo "rc=$?"
done
echo "[prof] calibration: FETCH_SIZE when every record of every listed brick is read (variant 2: 8 B per lane, known bytes)"
TL3D_TSDF_VARIANT=2 timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal_v2" -- python3 bench.py "${PMC[@]}" > "$OUT/bench_cal_v2.log" 2>&1
echo "rc=$?"
echo "[prof] round-1 formulation (free-space bricks streamed): FETCH_SIZE, WRITE_SIZE, and FETCH_SIZE of the free-space bricks alone"
TL3D_FREE_COUNTERS=0 timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/str_1" -- python3 bench.py "${PMC[@]}" > "$OUT/bench_str1.log" 2>&1
echo "rc=$?"
TL3D_FREE_COUNTERS=0 timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/str_2" -- python3 bench.py "${PMC[@]}" > "$OUT/bench_str2.log" 2>&1
echo "rc=$?"
TL3D_FREE_COUNTERS=0 TL3D_DEBUG_ONLY=2 timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/str_free" -- python3 bench.py "${PMC[@]}" > "$OUT/bench_str_free.log" 2>&1
echo "rc=$?"
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep what is judged (stats csv, summary, traffic json, bench logs); drop the raw per-dispatch csvs (hundreds of MB)
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
