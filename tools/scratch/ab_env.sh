#!/usr/bin/env bash
# usage: bash tools/scratch/ab_env.sh VAR v1 v2 [reps]   -- bench headline for two settings of one environment knob, interleaved
VAR="$1"; A="$2"; B="$3"; REPS="${4:-3}"
for i in $(seq 1 "$REPS"); do
  for v in "$A" "$B"; do
    env "$VAR=$v" python bench.py --no-cpu-baseline --no-rows $AB_ARGS 2>/dev/null > /tmp/ab.json
    python - "$VAR" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); r = d["roofline"]
print(sys.argv[1], sys.argv[2], d["value"], r["us_per_frame"], (r.get("single_frame_per_sweep") or {}).get("us_per_frame"))
PY
  done
done
