import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
from ap_explore import run
for nopack in ("1", "0"):
    os.environ["NBODY_AP_NOPACK"] = nopack
    for ipt in (2, 4):
        print("nopack", nopack, end=" ")
        run(1 << 18, 1, ipt, 0, 3)
