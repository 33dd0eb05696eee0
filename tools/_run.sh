cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > gpurun_out/r04_final2_bench.json 2> gpurun_out/r04_final2_bench.err; echo "bench rc=$?"
timeout -k 5 200 python3 tools/bench_icp.py --json gpurun_out/r04d_bench_icp.json > gpurun_out/r04d_bench_icp.txt 2>&1; echo "bench_icp rc=$?"
bash tools/pmc_icp.sh r04d > gpurun_out/pmc_icp_r04d.log 2>&1; echo "pmc_icp rc=$?"
timeout -k 5 100 python tools/bench_bp.py 2>&1 | grep -v amdgpu > gpurun_out/r04d_bench_bp.txt
