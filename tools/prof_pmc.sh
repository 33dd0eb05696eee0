#!/usr/bin/env bash
# usage: bash tools_pmc.sh <tag> "<counter group>" ["<counter group>" ...]   (env TL3D_DEBUG_ONLY etc. pass through)
# PMC passes only (never combined with tracing), small bench configuration, summary of the tl3d kernels.
set -uo pipefail
TAG="$1"; shift
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/pmc_${TAG}"
mkdir -p "$OUT"
SMALL=(--no-cpu-baseline --steps 1 --warmup 0 --frames-per-step 4 --resident-frames 4)
i=0
for grp in "$@"; do
  i=$((i+1))
  echo "[pmc] pass $i: $grp"
  timeout -k 5 75 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 bench.py "${SMALL[@]}" > "$OUT/bench_pmc$i.log" 2>&1
  echo "rc=$?"
done
python3 tools_prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
