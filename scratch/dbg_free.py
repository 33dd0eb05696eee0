import sys, numpy as np
sys.path.insert(0, '.')
import tl3d
from tl3d import synth
W,H = 540,960
cam = dict(width=W,height=H,fx=859.5,fy=859.5,cx=270.0,cy=480.0)
scene = synth.object_scene(True)
pose = synth.orbit_poses(4,1.0,11.25)[1]
d,c = synth.render(scene,pose,**cam)
print('depth range', d.min(), d.max(), (d==0).mean())
spec = tl3d.GridSpec.cube(256, 0.01, centre=(0,-0.1,0), channels=tl3d.CH_TSDF)
with tl3d.FusionContext(W,H,cam['fx'],cam['fy'],cam['cx'],cam['cy'],n_slots=1,grid=spec) as ctx:
    ctx.upload(0,d,None)
    ctx.set_profile(True,False)
    ctx.integrate(0,pose)
    st = ctx.stats()
    g = ctx.download_grid(tl3d.CH_TSDF)
print(st)
gb = g.reshape(-1,512,2)
touched = (gb[:,:,1]>0).any(1)
allfree = ((gb[:,:,1]==1)&(gb[:,:,0]==32767)).all(1)
print('bricks touched', touched.sum(), 'all-free bricks', allfree.sum(), 'updated voxels', (gb[:,:,1]>0).sum())
