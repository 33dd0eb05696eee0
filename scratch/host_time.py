import sys, time, numpy as np
sys.path.insert(0, '.')
import tl3d
from tl3d import synth
W,H = 1080,1920
cam = dict(width=W,height=H,fx=1719.0,fy=1719.0,cx=540.0,cy=960.0)
scene = synth.object_scene(True)
poses = synth.orbit_poses(8,1.0,45.0)
spec = tl3d.GridSpec.cube(512, 0.005, centre=(0,-0.1,0), channels=tl3d.CH_TSDF)
with tl3d.FusionContext(W,H,cam['fx'],cam['fy'],cam['cx'],cam['cy'],n_slots=8,grid=spec) as ctx:
    for i,p in enumerate(poses):
        d,c = synth.render(scene,p,**cam, want_color=False)
        ctx.upload(i,d,None)
    for rep in range(3):
        ctx.sync()
        t0=time.perf_counter()
        for k in range(256):
            ctx.integrate(k%8, poses[k%8])
        t1=time.perf_counter()
        ctx.sync()
        t2=time.perf_counter()
        print(f"enqueue {1e6*(t1-t0)/256:.1f} us/frame, total {1e6*(t2-t0)/256:.1f} us/frame")
    # pure python/ctypes overhead: a trivial call
    t0=time.perf_counter()
    for k in range(20000): ctx._lib.tl3d_version()
    print("ctypes trivial call us:", 1e6*(time.perf_counter()-t0)/20000)
    import ctypes as C
    from tl3d import _cabi as abi
    r,t = poses[0]
    t0=time.perf_counter()
    for k in range(20000): a=abi.ptr(abi.d9(r)); b=abi.ptr(abi.d3(t))
    print("arg marshalling us:", 1e6*(time.perf_counter()-t0)/20000)
