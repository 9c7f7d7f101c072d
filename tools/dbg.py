import sys; sys.path.insert(0,'/root/repo')
import numpy as np, nbody_amd as nb
g=np.load('/root/repo/tests/golden/refcube_n4096_s1.npz')
m,pos,vel=g['masses'],g['pos'],g['vel']; n=len(m)
for name,law in (("cpu_r6",0),("wgsl_bias",1),("plummer",2)):
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m,pos,vel); acc=sim.accel_allpairs()
    ref64=g[f"acc64_{name}"]; e=np.linalg.norm(acc[:,:3]-ref64,axis=1); e_ref=np.linalg.norm(g[f"acc_{name}"][:,:3]-ref64,axis=1)
    s=np.linalg.norm(ref64,axis=1)
    bad=e>1e-5*s+e_ref
    print(name, bad.sum(), (e/s).max(), (e_ref/s).max(), np.median(e/s), np.median(e_ref/s))
    i=np.argmax(e/s - e_ref/s); print(i, e[i]/s[i], e_ref[i]/s[i], acc[i], ref64[i], g[f"acc_{name}"][i])
