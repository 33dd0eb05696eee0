#!/usr/bin/env bash
# Register / scratch / LDS figures of the gfx950 kernels in an object file or library (the notes of the bundled code objects).
#   tools/codeobj.sh textureless-3d-reconstruction_amd/csrc/build/kernels_icp.o [name-filter]
set -euo pipefail
F="$1"; PAT="${2:-.}"
LLVM=/opt/rocm/lib/llvm/bin
TMP="$(mktemp -d)"; trap 'rm -rf "$TMP"' EXIT
"$LLVM/llvm-objcopy" --dump-section .hip_fatbin="$TMP/fb.bin" "$F"
"$LLVM/clang-offload-bundler" --unbundle --type=o --input="$TMP/fb.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/dev.co"
"$LLVM/llvm-readelf" --notes "$TMP/dev.co" | awk -v pat="$PAT" '
  /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_spill_count:/ {vs=$2}
  /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
  /\.wavefront_size:/ { if (name ~ pat) printf "%-100s vgpr %3d agpr %3d sgpr %3d spill %3d scratch %5d lds %6d\n", name, v, a, s, vs, p, g }'
