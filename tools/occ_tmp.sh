set -uo pipefail
mkdir -p gpurun_out; O=gpurun_out/occ.txt; : > $O
timeout -k 5 600 python3 -m pytest tests/test_gpu_fusion_parity.py -x -q -m gpu -k "tsdf or batch or culling or ragged" > gpurun_out/occ_pytest.txt 2>&1; echo "pytest rc=$?" >> $O; tail -2 gpurun_out/occ_pytest.txt >> $O
for r in 1 2; do
timeout -k 5 200 python3 bench.py --no-cpu-baseline --no-rows --no-single 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], r['ms_per_launch'], r['us_per_frame'])" >> $O
done
cat $O
