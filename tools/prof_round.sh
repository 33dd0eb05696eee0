#!/usr/bin/env bash
# usage (GPU box, repo root):  bash tools/prof_round.sh r03
# 1. rocprofv3 --kernel-trace --stats of (a) the timed loop alone (`bench.py --no-cpu-baseline --no-rows --no-single`: every
#    tsdf_update_kernel dispatch is a 32-frame launch of the headline workload, so its average duration must agree with
#    roofline.ms_per_launch) and (b) the DEFAULT command (all rows; the update kernel's dispatches split by the phase of bench.py they belong to);
# 2. separate --pmc passes (never combined with tracing; FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950) over ONE step of
#    64 frames (two update launches of 32 frames) of the same workload -> per-dispatch means for every tl3d kernel, pmc_traffic.json;
# 3. FETCH_SIZE calibrated on known byte counts in the kernel's own access patterns (tools/ubench_fetch.hip);
# 4. plain bench lines: default (f32), 16-bit frames, the config-5 shape.
# Writes gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
set -uo pipefail
TAG="${1:-r04}"
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/prof_${TAG}"
rm -rf "$OUT"; mkdir -p "$OUT"
echo "[prof] kernel trace 1 (timed loop only): python3 bench.py --no-cpu-baseline --no-rows --no-single"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1" -- python3 bench.py --no-cpu-baseline --no-rows --no-single > "$OUT/bench_traced.json" 2> "$OUT/bench_traced.err"
echo "rc=$?"
f=$(find "$OUT/trace1" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && grep -E "^\"Name\"|tl3d" "$f" > "$OUT/kernel_stats.csv"
echo "[prof] kernel trace 2 (the default command, rows and the one-frame-per-launch section included): python3 bench.py"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace2" -- python3 bench.py > "$OUT/bench_rows_traced.json" 2> "$OUT/bench_rows_traced.err"
echo "rc=$?"
f=$(find "$OUT/trace2" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && grep -E "^\"Name\"|tl3d" "$f" > "$OUT/rows_kernel_stats.csv"
python3 - "$OUT" <<'PY' > "$OUT/update_by_phase.txt"
# The update kernel's dispatches of the DEFAULT command, split by the phase of bench.py they belong to.  bench.py runs, in this
# order: warm-up + timed steps + the profiled re-run of the timed steps (all 32-frame launches of the headline workload), then the
# one-frame-per-launch section and the rows.  The first (warmup + 2 steps) x frames_per_step / 32 dispatches of the non-counting
# instantiation are therefore the launches roofline.ms_per_launch averages over (its hipEvent pairs cover the re-run).
import csv, glob, json, os, sys
out = sys.argv[1]
b = None
for l in open(os.path.join(out, "bench_rows_traced.json")):
    if l.startswith("{"): b = json.loads(l)
rows = []
for f in glob.glob(os.path.join(out, "trace2", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "tsdf_update_pairs_kernel<false" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
if b and rows:
    per_step = 512 // 32
    n_head = (b["warmup"] + 2 * b["steps"]) * per_step
    head, rest = [d for _, d in rows[:n_head]], [d for _, d in rows[n_head:]]
    print(f"tsdf_update_pairs_kernel<false, ...> dispatches of `python3 bench.py`: {len(rows)}")
    print(f"  first {len(head)} (warm-up, timed region, profiled re-run: 32 frames per launch): mean {sum(head) / len(head):.1f} us  min {min(head):.1f}  max {max(head):.1f}")
    print(f"  last {len(head) - b['warmup'] * per_step - b['steps'] * per_step} of those (the profiled re-run): mean {sum(head[-b['steps'] * per_step:]) / (b['steps'] * per_step):.1f} us;  bench.py's hipEvent figure: {1e3 * b['roofline']['ms_per_launch']:.1f} us")
    if rest: print(f"  the other {len(rest)} (one frame per launch, rows): mean {sum(rest) / len(rest):.1f} us  min {min(rest):.1f}  max {max(rest):.1f}")
PY
cat "$OUT/update_by_phase.txt"
bash tools/pmc_quick.sh "${TAG}" > "$OUT/pmc.log" 2>&1
cp "gpurun_out/pmc_${TAG}/summary.txt" "$OUT/pmc_summary.txt" 2>/dev/null
cp "gpurun_out/pmc_${TAG}/bench1.log" "$OUT/bench_pmc.log" 2>/dev/null
echo "[prof] FETCH_SIZE calibration"
hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o /tmp/ubench_fetch 2>/dev/null
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal" -- /tmp/ubench_fetch > "$OUT/ubench_fetch.txt" 2>&1
python3 - "$OUT" <<'PY' > "$OUT/pmc_calibration.txt"
import csv, glob, os, sys
out = sys.argv[1]
print(open(os.path.join(out, "ubench_fetch.txt")).read())
known = {"read8": 1 << 30, "read16": 1 << 30, "gather4": 4096 * 256 * 64 * 64}
for f in glob.glob(os.path.join(out, "cal", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if r["Counter_Name"] == "FETCH_SIZE" and k in known:
            b = float(r["Counter_Value"]) * 1024.0
            print(f"{k:8s} FETCH_SIZE {b / 1e6:10.1f} MB  known {known[k] / 1e6:10.1f} MB{' (64-B lines)' if k == 'gather4' else ''}  ratio {b / known[k]:.3f}")
PY
cat "$OUT/pmc_calibration.txt"
python3 tools/prof_summary.py "$OUT" "gpurun_out/pmc_${TAG}" > "$OUT/traffic.log" 2>&1; cat "$OUT/traffic.log"
echo "[prof] bench lines"
timeout -k 5 300 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"
timeout -k 5 300 python3 bench.py --depth-format u16 --no-rows --no-cpu-baseline > "$OUT/bench_u16_frames.json" 2>/dev/null; echo "rc=$?"
timeout -k 5 300 python3 bench.py --width 3840 --height 2160 --grid 1024 --voxel 0.002 --resident-frames 256 --frames-per-step 256 --steps 4 --no-rows --no-cpu-baseline > "$OUT/bench_config5_size.json" 2>/dev/null; echo "rc=$?"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
rm -rf "$OUT/trace1" "$OUT/trace2" "$OUT/cal"
du -sh "$OUT"
