cd $GRAFT_REPO_ROOT
TL3D_LIB=$PWD/textureless-3d-reconstruction_amd/libtl3d_w3.so timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "icp or config or pipeline" 2>&1 | tail -60 > gpurun_out/icp_w3_fail.txt
cat gpurun_out/icp_w3_fail.txt | tail -50
