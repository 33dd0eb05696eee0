set -uo pipefail
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r03_prof1; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-rows --no-single --steps 4 > $OUT/bench.json 2> $OUT/bench.err
echo rc=$?
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv; grep tl3d $OUT/kernel_stats.csv | cut -c1-60,100-400
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/bench.json | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], json.dumps(j['roofline']))"
