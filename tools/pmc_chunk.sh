# usage (GPU box): bash tools/pmc_chunk.sh "VAR=val ..." ...   -- FETCH_SIZE per frame and L2 hit rate of tsdf_update_kernel per setting (experiments flavour); appends to gpurun_out/r03_pmc_chunk.txt
set -uo pipefail
export TMPDIR=/tmp
cd /root/repo
LIB=$PWD/textureless-3d-reconstruction_amd/libtl3d_exp.so
[[ -f "$LIB" ]] || TL3D_FLAVOUR=experiments bash textureless-3d-reconstruction_amd/csrc/build.sh >/dev/null
ARGS=(--no-cpu-baseline --no-rows --no-single --steps 1 --warmup 0 --frames-per-step 64 --resident-frames 512)
for setting in "$@"; do
  OUT=/tmp/pmcc; rm -rf $OUT
  echo "[pmc_chunk] $setting" >> gpurun_out/r03_pmc_chunk.txt
  env TL3D_LIB=$LIB $setting timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 bench.py "${ARGS[@]}" > /tmp/pmcc.log 2>&1
  python3 - "$setting" >> gpurun_out/r03_pmc_chunk.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
a = defaultdict(list)
for f in glob.glob("/tmp/pmcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tsdf_update_kernel<false" in r["Kernel_Name"]:
            a[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in a.items()}
import json
ls = [l for l in open("/tmp/pmcc.log") if l.startswith("{")]
if not ls or not m:
    print(f"{sys.argv[1]:40s} no result"); sys.exit(0)
fps = json.loads(ls[0])["roofline"]["frames_per_sweep"]
print(f"{sys.argv[1]:40s} FETCH_SIZE {m.get('FETCH_SIZE',0)*1024/1e6/fps:8.1f} MB/frame  (frames/launch {fps})")
PY
done
