"""Times the all-pairs force kernel for several launch shapes (IPT, slices) and laws."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb

def run(n, law, ipt, slices, steps):
    os.environ["NBODY_AP_IPT"] = str(ipt)
    os.environ["NBODY_AP_SLICES"] = str(slices)
    m, p, v = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.step_allpairs(1)
        sim.sync()
        sim.set_kernel_timing(True)
        t0 = time.time()
        sim.step_allpairs(steps)
        sim.sync()
        wall = (time.time() - t0) / steps
        ms, cnt = sim.kernel_time()
    inter = float(n) * n
    flop = {0: 20, 1: 46, 2: 20}[law]
    print(f"n={n:8d} law={law} ipt={ipt} slices={slices:3d}: kernel {ms:9.3f} ms  wall/step {wall*1e3:9.3f} ms  "
          f"{inter/ms/1e9:8.3f} Ginter/ms-> {inter/(ms*1e-3)/1e12:6.3f} Tint/s  {inter*flop/(ms*1e-3)/1e12:7.2f} TFLOP/s "
          f"({inter*flop/(ms*1e-3)/157.3e12*100:5.1f}% of 157.3)", flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "small"):
        for ipt in (1, 2, 4):
            for sl in (0, 8, 16, 32, 64):
                run(65536, 2, ipt, sl, 20)
    if which in ("all", "big"):
        for ipt in (2, 4, 8):
            for sl in (1, 2, 4):
                run(1 << 20, 2, ipt, sl, 2)
    if which in ("all", "laws"):
        run(1 << 18, 0, 4, 0, 3)
        run(1 << 18, 1, 4, 0, 3)
        run(1 << 18, 1, 2, 0, 3)
        run(1 << 18, 2, 4, 0, 3)
