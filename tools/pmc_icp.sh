#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/pmc_icp.sh <tag>
# Separate rocprofv3 --pmc passes (never combined with tracing) over the batched registration alone (tools/bench_icp.py: 127 pairs of
# 1080x1920 frames, ten iterations at stride 4, one launch per repetition): per-dispatch means of every counter for icp_batch_kernel,
# and what they say about where its waves spend their time.
set -uo pipefail
TAG="${1:-icp}"
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/pmc_icp_${TAG}"; rm -rf "$OUT"; mkdir -p "$OUT"
ARGS=(--frames 128 --radius 1 --cases "stride 4" --reps 2)
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TA_TOTAL_WAVEFRONTS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -- python3 tools/bench_icp.py "${ARGS[@]}" > "$OUT/bench$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "icp_batch_kernel" not in k and "normals_kernel" not in k and "smooth_depth" not in k: continue
        agg[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k in sorted(agg):
        print(k, file=fh)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"    {c:36s} mean {sum(v)/len(v):16.1f}  max {max(v):16.1f}  (n={len(v)})", file=fh)
        a = {c: sum(v) / len(v) for c, v in agg[k].items()}
        if "icp_batch" in k and "SQ_INSTS_VALU" in a and "GRBM_GUI_ACTIVE" in a:
            # one launch = 127 pairs x 11 passes; 1024 SIMDs; a vector instruction occupies its SIMD's issue port for >= 4 cycles
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0                 # the counter is summed over the 8 XCDs
            pp = 127 * 11
            print(f"    -> vector instructions per pair-pass {a['SQ_INSTS_VALU'] / pp:10.0f}  (x 64 lanes / 129600 samples = {a['SQ_INSTS_VALU'] / pp * 64 / 129600:.1f} per sample)", file=fh)
            print(f"    -> vector issue: {a['SQ_INSTS_VALU'] * 4 / 1024 / cyc:.3f} of the launch at 4 cycles per instruction and SIMD ({a['SQ_INSTS_VALU'] * 4.4 / 1024 / cyc:.3f} at the measured 4.4), launch = {cyc / 1e6:.2f} M cycles per XCD", file=fh)
            if "SQ_WAVE_CYCLES" in a and "SQ_WAIT_INST_ANY" in a:
                print(f"    -> wave-cycles waiting for an instruction's operands (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) {a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES']:.3f}; "
                      f"waiting for anything (SQ_WAIT_ANY) {a.get('SQ_WAIT_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f}; issuing {a.get('SQ_ACTIVE_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f}", file=fh)
            if "FETCH_SIZE" in a:
                print(f"    -> FETCH_SIZE per pair-pass {a['FETCH_SIZE'] * 1024 / pp / 1e6:.2f} MB (algorithmic: 2.59 MB; coalesced reads tally at half on gfx950), "
                      f"L2 hit rate {a.get('TCC_HIT_sum', 0) / max(1.0, a.get('TCC_HIT_sum', 0) + a.get('TCC_MISS_sum', 0)):.3f}", file=fh)
print(open(os.path.join(out, "summary.txt")).read())
PY
grep -h "^radius" "$OUT"/bench1.log | cut -c1-160
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
