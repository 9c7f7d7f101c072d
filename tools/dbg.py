import sys; sys.path.insert(0,'/root/repo')
import numpy as np, nbody_amd as nb
from oracle import oracle as orc
n=1000
m,pos=nb.gen_refcube(n,seed=1)
vel=np.zeros((n,4),np.float32); vel[:,:3]=0.3*(nb.gen_refcube(n,seed=2)[1][:,:3]/125.0-1.0)
with nb.NBodySim(n, law=1) as sim:
    sim.upload(m,pos,vel); sim.step_allpairs(1); p1,v1,a1=sim.download()
rp,rv,ra=orc.integrate_wgsl(pos,vel,np.zeros_like(pos),a1)
print(np.abs(rp-p1).max(), np.abs(rv-v1).max(), (rp!=p1).sum(), (rv!=v1).sum())
i=np.argwhere(rp!=p1)[:3]; print(i, rp[i[:,0]], p1[i[:,0]], pos[i[:,0]], vel[i[:,0]])
