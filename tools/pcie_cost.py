"""Host-transfer costs at the boundary: nbody_download of the positions into pageable and into
page-locked host memory, a Barnes-Hut step followed by that download (what src/nbody_gpu_bh.rs
does per frame), and nbody_upload."""
import sys, time, ctypes as C; sys.path.insert(0, '.')
import numpy as np, torch, nbody_amd as nb
from nbody_amd.sim import _ptr
for n in (1 << 20, 1 << 24):
    m, p, v = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=2) as sim:
        sim.upload(m, p, v); sim.step_bh(2); sim.sync()
        pageable = np.empty((n, 4), np.float32)
        pinned = torch.empty((n, 4), dtype=torch.float32, pin_memory=True).numpy()
        for name, pos in (("pageable", pageable), ("page-locked", pinned)):
            sim._ck(sim.lib.nbody_download(sim.h, _ptr(pos), None, None))
            t0 = time.perf_counter()
            for _ in range(5): sim._ck(sim.lib.nbody_download(sim.h, _ptr(pos), None, None))
            td = (time.perf_counter() - t0) / 5
            t0 = time.perf_counter()
            for _ in range(5):
                sim.step_bh(1); sim._ck(sim.lib.nbody_download(sim.h, _ptr(pos), None, None))
            ts = (time.perf_counter() - t0) / 5
            print(f"n={n} {name}: download positions {td*1e3:.2f} ms ({16*n/td/1e9:.1f} GB/s), step_bh + download {ts*1e3:.2f} ms", flush=True)
        t0 = time.perf_counter(); sim.upload(m, p, v); tu = time.perf_counter() - t0
        print(f"n={n}: upload (m,pos,vel) {tu*1e3:.2f} ms", flush=True)
