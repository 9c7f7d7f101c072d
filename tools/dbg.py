import sys; sys.path.insert(0,'/root/repo')
import numpy as np, nbody_amd as nb
from oracle import oracle as orc
for n in (2,3,5,9,20,64,65,200,1000):
    m,pos=orc.gen_refcube(n,seed=3)
    vel=np.zeros((n,4),np.float32)
    nodes,_,_,_=orc.build_octree(pos,m)
    ref,cnt=orc.bh_accel(2,pos,vel,nodes)
    with nb.NBodySim(n, law=2) as sim:
        sim.upload(m,pos,vel); sim.build_octree(); sim.set_counting(True); acc=sim.accel_bh(); c=sim.counters()
    print(n, len(nodes), (c["bh_node_visits"],c["bh_leaf_interactions"],c["bh_aggregate_interactions"]), (cnt["visits"],cnt["leaf"],cnt["aggregate"]), np.abs(acc-ref).max())
