import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
from ap_explore import run
for tpl in (4, 8):
    for seg in (32, 64, 128):
        os.environ["NBODY_AP_SYM_TPL"] = str(tpl); os.environ["NBODY_AP_SEG"] = str(seg)
        print("tpl", tpl, "seg", seg, end="  ")
        run(1 << 20, 2, 0, 0, 2)
del os.environ["NBODY_AP_SEG"]; del os.environ["NBODY_AP_SYM_TPL"]
for n in (16384, 32768, 65536, 131072, 262144, 524288):
    for sym in ("0", "1"):
        os.environ["NBODY_AP_SYM"] = sym
        print("sym", sym, end="  ")
        run(n, 2, 0, 0, 10)
