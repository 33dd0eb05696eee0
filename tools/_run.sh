cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "frame_slabs" 2>&1 | tail -5
for c in 4 3 2; do
timeout -k 10 280 python tools/run_config.py --config $c --out gpurun_out/r04c_config_runs.jsonl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('config $c: first', d['first_call_in_process']['reconstruct_s'], d['first_call_in_process']['stage_s'], 'second', d['second_call_in_process']['reconstruct_s'], d['second_call_in_process']['stage_s'])"
done
