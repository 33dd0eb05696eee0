set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fusion_parity.py tests/test_gpu_headline_shape.py -m gpu -x -q -k "tsdf or batch or headline or sparse or free_space or culling or ragged or deterministic or reset or 16bit" > gpurun_out/r04_t04.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t04.log
tail -3 gpurun_out/r04_t04.log
grep -q "rc=0" gpurun_out/r04_t04.log || exit 1
export TL3D_LIB=$PWD/textureless-3d-reconstruction_amd/libtl3d_exp.so
for cfg in "TL3D_SINGLE_STREAM=1 TL3D_UPDATE_BLOCKS=2048" "TL3D_SINGLE_STREAM=1 TL3D_UPDATE_BLOCKS=2048 TL3D_NO_ORDER=1"; do
  echo "== $cfg"
  env $cfg TL3D_PAIRS_EXP=16 timeout -k 5 120 python3 bench.py --no-cpu-baseline --no-rows --no-single --steps 1 --warmup 0 --frames-per-step 32 2>&1 >/dev/null | grep "tl3d exp" | tail -2
done > gpurun_out/r04_stamps3.txt 2>&1
cat gpurun_out/r04_stamps3.txt
unset TL3D_LIB
STEPS=4 bash tools/sweep_tsdf.sh gpurun_out/r04_sweep06.txt "TL3D_UPDATE_BLOCKS=1024" "TL3D_UPDATE_BLOCKS=1536" "TL3D_UPDATE_BLOCKS=2048" "TL3D_UPDATE_BLOCKS=3072" "TL3D_UPDATE_BLOCKS=2048 TL3D_NO_ORDER=1" "TL3D_SINGLE_STREAM=1 TL3D_UPDATE_BLOCKS=2048"
