cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in libtl3d_prev.so libtl3d.so; do
  TL3D_LIB=$PWD/textureless-3d-reconstruction_amd/$lib timeout -k 5 120 python3 bench.py --no-cpu-baseline --no-rows --no-single --steps 8 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('$lib', round(j['value']), 'f/s  update', r['us_per_frame'], 'us/frame  frac', r['frac'], ' all', round(1e3*r['ms_per_frame_all_kernels'],2))" >> gpurun_out/r04_order_ab.txt
done; done
cat gpurun_out/r04_order_ab.txt
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "tsdf or batch or headline or sparse or fusion or config5" 2>&1 | tail -2
