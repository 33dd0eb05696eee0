#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/round_all.sh r03  -- the round's profile set (tools/prof_round.sh) and the four full-length
# BASELINE configuration runs (tools/run_config.py); everything lands under gpurun_out/.
set -uo pipefail
TAG="${1:-r04}"
bash tools/prof_round.sh "$TAG" > "gpurun_out/prof_round_${TAG}.log" 2>&1; echo "prof_round rc=$?"
: > "gpurun_out/${TAG}_config_runs.jsonl"
for c in 2 3 4 4b; do
  timeout -k 5 300 python3 tools/run_config.py --config $c --out "gpurun_out/${TAG}_config_runs.jsonl" > "gpurun_out/run_config_$c.log" 2>&1; echo "config $c rc=$?"
done
bash tools/pmc_icp.sh "$TAG" > "gpurun_out/pmc_icp_${TAG}.log" 2>&1; echo "pmc_icp rc=$?"
timeout -k 5 200 python3 tools/bench_icp.py --json "gpurun_out/${TAG}_bench_icp.json" > "gpurun_out/${TAG}_bench_icp.txt" 2>&1; echo "bench_icp rc=$?"
tail -5 "gpurun_out/prof_round_${TAG}.log"
