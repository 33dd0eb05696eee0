cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/r04_b10.json 2> gpurun_out/r04_b10.err; echo "bench rc=$?"; tail -3 gpurun_out/r04_b10.err
python3 -c "
import json;j=json.load(open('gpurun_out/r04_b10.json'));print(j['value'],j['roofline']['ms_per_launch'],j['roofline']['frac'], j['roofline']['bound']);print(json.dumps(j['rows'], indent=0)[:3000])"
rm -f gpurun_out/r04_cfg.jsonl
for c in 2 3 4; do timeout -k 5 300 python3 tools/run_config.py --config $c --out gpurun_out/r04_cfg.jsonl > gpurun_out/r04_cfg_$c.log 2>&1; echo "config $c rc=$?"; done
python3 -c "
import json
for l in open('gpurun_out/r04_cfg.jsonl'):
    j=json.loads(l); print(j['config'][:30], j['frames_per_s_whole_pipeline'], j['stage_s'], j.get('grid'), j['stats'].get('sparse'), j.get('chamfer_vs_reference_cpu_path_mm'), j.get('chamfer_vs_reference_cpu_path_first_100_frames_mm'))"
grep -h "Occupancy" gpurun_out/r04_cfg_*.log
