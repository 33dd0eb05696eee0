"""pmc_traffic.json for bench.py from the --pmc passes of tools/pmc_quick.sh:  python3 tools/prof_summary.py <out dir> <pmc dir>.

traffic = (FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch of the update kernel PLUS the half of its coalesced record reads that
FETCH_SIZE does not tally on gfx950 (8 B per lane reads report 0.500 of their bytes, like the 16-B case of the guide; scattered
4-B gathers report 0.98 x 64 B per lane; WRITE_SIZE is exact: tools/ubench_fetch.hip, profiles/r03_pmc_calibration.txt).  The
record bytes read equal the record bytes written (every touched record is read and written once per launch), so the correction is
0.5 x WRITE_SIZE.  This corrected figure is what bench.py reports as roofline.traffic."""
import json
import os
import re
import sys

out, pmc = sys.argv[1], sys.argv[2]
vals = {}
cur = None
for line in open(os.path.join(pmc, "summary.txt")):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    m = re.match(r"\s+(\S+)\s+mean\s+([0-9.]+)", line)
    if m and cur and ("tsdf_update_kernel<false" in cur or "tsdf_update_pairs_kernel<false" in cur):
        vals[m.group(1)] = float(m.group(2))
bench = None
for line in open(os.path.join(pmc, "bench1.log")):
    if line.startswith("{"):
        bench = json.loads(line)
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and bench:
    r = bench["roofline"]
    # gfx950: FETCH_SIZE tallies the 128-B requests of a coalesced 8- or 16-B-per-lane read at 64 B (MI355X guide, HBM section; confirmed
    # for 8 B per lane in pmc_calibration.txt) and the 64-B requests of scattered 4-B gathers at face value.  This kernel reads its
    # records 8 B per lane (coalesced; as many as it writes: WRITE_SIZE is exact) and gathers depth 4 B per lane: the records' half
    # that FETCH_SIZE misses is added back.
    rec_read = min(vals["WRITE_SIZE"], 2.0 * vals["FETCH_SIZE"]) * 1024.0
    j = {"grid": bench["config"]["grid"], "width": 1080, "height": 1920, "depth_format": bench["config"]["depth_format"], "free_space_counters": True,
         "kernel": "tsdf_update_pairs_kernel", "frames_per_sweep": int(round(r["frames_per_sweep"])), "fetch_size_kb": round(vals["FETCH_SIZE"], 1),
         "write_size_kb": round(vals["WRITE_SIZE"], 1),
         "hbm_bytes_per_launch": int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 + 0.5 * rec_read),
         "record_read_bytes_added_back": int(0.5 * rec_read),
         "depth_gather_bytes_per_launch": int(vals["FETCH_SIZE"] * 1024.0 - 0.5 * rec_read),
         "algorithmic_bytes_per_launch": r["bytes_per_launch"],
         "l2_hit_rate": round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 3) if "TCC_HIT_sum" in vals else None,
         # what the kernel is actually bound by: 64-B requests that leave the L2, against the chip's measured rate for
         # scattered 4-B reads (tools/ubench_gather.hip: 55.4 G/s), and vector-ALU instructions (4 cycles each on 1024 SIMDs)
         "l2_read_requests_per_launch": int(vals["TCC_EA0_RDREQ_sum"]) if "TCC_EA0_RDREQ_sum" in vals else None,
         "scattered_read_roof_requests_per_s": 55.4e9,
         "valu_instructions_per_launch": int(vals["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in vals else None,
         "note": "FETCH_SIZE + WRITE_SIZE (x 1024) per dispatch from separate --pmc passes, plus the half of the coalesced 8-B-per-lane "
                 "record reads that FETCH_SIZE does not tally on gfx950 (record bytes read = bytes written = WRITE_SIZE); L2-side requests: "
                 "Infinity-Cache hits are included, so this is an upper bound on HBM bytes"}
    json.dump(j, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(j))
else:
    print("no traffic figures found", vals.keys())
