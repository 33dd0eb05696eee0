#!/usr/bin/env bash
# usage (GPU box, repo root):  bash tools/sweep_tsdf.sh <out-file> "<VAR=val ...>" ["<VAR=val ...>" ...]
# Runs the headline bench once per environment setting against the EXPERIMENTS flavour of the library (libtl3d_exp.so, built here
# if missing) and prints frames/s + the update kernel's time per frame.  The shipped library ignores these variables.
set -uo pipefail
OUT="$1"; shift
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
LIB="$HERE/textureless-3d-reconstruction_amd/libtl3d_exp.so"
[[ -f "$LIB" ]] || TL3D_FLAVOUR=experiments bash "$HERE/textureless-3d-reconstruction_amd/csrc/build.sh" >/dev/null
: > "$OUT"
for setting in "$@"; do
  line=$(env TL3D_LIB="$LIB" $setting timeout -k 5 120 python3 "$HERE/bench.py" --no-cpu-baseline --no-rows --steps ${STEPS:-4} ${BENCH_ARGS:-} 2>/dev/null)
  python3 - "$setting" "$line" >> "$OUT" <<'PY'
import json, sys
s, line = sys.argv[1], sys.argv[2]
try:
    j = json.loads(line); r = j["roofline"]; sg = r.get("single_frame_per_sweep") or {}
    print(f"{s:60s} {j['value']:9.0f} f/s  pair {r['us_per_frame']:6.2f} us/frame ({r['frac']:.3f})  single {sg.get('us_per_frame', 0):6.2f} ({sg.get('frac', 0):.3f})  all-kernels {1e3 * r['ms_per_frame_all_kernels']:6.2f} us/frame")
except Exception as e:
    print(f"{s:60s} FAILED {e!r} {line[:200]!r}")
PY
done
cat "$OUT"
