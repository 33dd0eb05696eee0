cd $GRAFT_REPO_ROOT
export TL3D_LIB=$PWD/textureless-3d-reconstruction_amd/libtl3d_exp.so
for t in 2048 1024 512 2048 1024 512; do TL3D_BP_TILE=$t timeout -k 5 100 python tools/bench_bp.py 2>&1 | grep -v amdgpu >> gpurun_out/bp_tiles.txt; done
unset TL3D_LIB
timeout -k 5 100 python tools/bench_bp.py 2>&1 | grep -v amdgpu >> gpurun_out/bp_tiles.txt
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "backproject or dense or cli or pointcloud" 2>&1 | tail -2 >> gpurun_out/bp_tiles.txt
cat gpurun_out/bp_tiles.txt
