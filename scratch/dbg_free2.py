import sys, numpy as np
sys.path.insert(0, '.')
import tl3d
from tl3d import synth
W,H = 1080,1920
cam = dict(width=W,height=H,fx=1719.0,fy=1719.0,cx=540.0,cy=960.0)
scene = synth.object_scene(True)
pose = synth.orbit_poses(32,1.0,11.25)[1]
d,c = synth.render(scene,pose,**cam)
spec = tl3d.GridSpec.cube(512, 0.005, centre=(0,-0.1,0), channels=tl3d.CH_TSDF)
with tl3d.FusionContext(W,H,cam['fx'],cam['fy'],cam['cx'],cam['cy'],n_slots=1,grid=spec) as ctx:
    ctx.upload(0,d,None)
    ctx.set_profile(True,False)
    ctx.integrate(0,pose)
    st = ctx.stats()
    g = ctx.download_grid(tl3d.CH_TSDF)
print(st)
gb = g.reshape(-1,512,2)
w = gb[:,:,1]; q = gb[:,:,0]
cnt = (w>0).sum(1)
touched = cnt>0
allfree = ((w==1)&(q==32767)).all(1)
print('bricks touched', touched.sum(), 'full', (cnt==512).sum(), 'all-free bricks', allfree.sum(), 'updated voxels', (w>0).sum(), 'free vox', (q==32767).sum())
print('hist', np.histogram(cnt[touched], bins=[1,64,128,256,384,511,512,513])[0])
# where are the all-free bricks in z (camera depth)?
nb = 64
idx = np.nonzero(allfree)[0]
bx = idx % nb; by = (idx//nb)%nb; bz = idx//(nb*nb)
ctr = np.stack([(bx*8+4)*0.005-1.28, (by*8+4)*0.005-1.38, (bz*8+4)*0.005-1.28],1)
zc = (pose[0]@ctr.T).T[:,2] + pose[1].ravel()[2]
print('all-free brick depth hist', np.histogram(zc, bins=[0,0.25,0.5,0.75,1.0,1.25,1.5,2,3])[0])
