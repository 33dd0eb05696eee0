cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_fusion_parity.py -m gpu -x -q -k "centroid" 2>&1 | tail -2
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04_b11.json 2> gpurun_out/r04_b11.err; echo "bench rc=$?"
python3 -c "
import json;j=json.load(open('gpurun_out/r04_b11.json'));print(j['value']);r=j['rows'];print({k:r[k] for k in ['tsdf_plus_centroid_s2_fps','centroid_s2_us_per_frame']})"
