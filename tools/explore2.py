import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
from ap_explore import run
for ipt in (2, 4, 8):
    for sl in (1, 2, 4):
        run(1 << 20, 2, ipt, sl, 2)
os.environ["NBODY_AP_NOPACK"] = "1"
run(1 << 20, 2, 4, 2, 2)
del os.environ["NBODY_AP_NOPACK"]
for ipt in (2, 4):
    for sl in (0, 16, 64):
        run(65536, 2, ipt, sl, 20)
