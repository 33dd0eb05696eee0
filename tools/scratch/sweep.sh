#!/usr/bin/env bash
# usage: bash tools/scratch/sweep.sh "VAR=val VAR2=val" ...   -- one short bench per environment setting, prints kernel us and f/s
for cfg in "$@"; do
  env $cfg timeout -k 10 100 python bench.py --steps 3 --no-cpu-baseline --no-rows 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('%-44s kernel %.2f us  %8.0f f/s  bytes %d frac %.3f' % ('$cfg', 1e3 * r['ms_per_launch'], j['value'], r['bytes_per_launch'], r['frac']))"
done
