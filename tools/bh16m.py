import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bh_explore import run
run(1 << 24, 2, steps=2)
