cd $GRAFT_REPO_ROOT
P=$PWD/textureless-3d-reconstruction_amd
for cfg in "exp_s2 32" "exp_s2 16" "exp_s3 32" "exp_s3 16" "exp 32" "exp 16" "exp 8" "exp_s2 8"; do
  set -- $cfg
  echo "== lib $1 members $2" >> gpurun_out/icp_sweep.txt
  TL3D_LIB=$P/libtl3d_$1.so TL3D_ICP_MEMBERS=$2 timeout -k 10 120 python tools/bench_icp.py --frames 256 --radius 1 --cases "stride 4,two-level" 2>&1 | grep "^radius" | cut -c1-140 >> gpurun_out/icp_sweep.txt
done
