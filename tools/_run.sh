cd $GRAFT_REPO_ROOT
: > gpurun_out/r04f_config_runs.jsonl
for c in 2 3 4 4b; do
timeout -k 10 280 python tools/run_config.py --config $c --out gpurun_out/r04f_config_runs.jsonl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('config $c:', d['frames_per_s_whole_pipeline'], 'f/s; calls', d['first_call_in_process']['reconstruct_s'], d['second_call_in_process']['reconstruct_s'], d['third_call_in_process']['reconstruct_s'], d['stage_s'])"
done
