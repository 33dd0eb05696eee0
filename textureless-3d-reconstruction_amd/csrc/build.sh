#!/usr/bin/env bash
# Builds libtl3d.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : the f32/fp64 sequences are part of the parity contract with the oracle;
#                       FMAs appear only where the source says fmaf().
# TL3D_FLAVOUR=experiments builds libtl3d_exp.so (-DTL3D_EXPERIMENTS: tuning knobs read from the environment, timing
# ablations, in-kernel traces); load it with TL3D_LIB=<path>.  The default library contains none of that.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function)
if [[ "${TL3D_FLAVOUR:-}" == "experiments" ]]; then
  OUT="${HERE}/../libtl3d_exp.so"; BUILD="${HERE}/build_exp"; FLAGS+=(-DTL3D_EXPERIMENTS)
else
  OUT="${HERE}/../libtl3d.so"; BUILD="${HERE}/build"
fi
OBJS=()
PIDS=()
mkdir -p "${BUILD}"
for f in tl3d_api kernels_backproject kernels_centroid kernels_tsdf kernels_icp kernels_extract kernels_sor kernels_grid; do
  src="${HERE}/${f}.hip"; obj="${BUILD}/${f}.o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "${HERE}/tl3d_internal.h" -nt "$obj" || "${HERE}/bp_device.h" -nt "$obj" || "${HERE}/../../include/tl3d.h" -nt "$obj" ]]; then
    rm -f "$obj"                                   # a failed compile must not leave a stale object to link
    "$HIPCC" "${FLAGS[@]}" ${TL3D_EXTRA_FLAGS:-} -c "$src" -o "$obj" &
    PIDS+=($!)
  fi
  OBJS+=("$obj")
done
FAIL=0
for p in "${PIDS[@]:-}"; do [[ -z "$p" ]] || wait "$p" || FAIL=1; done
if [[ "$FAIL" != 0 ]]; then echo "build FAILED" >&2; exit 1; fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
