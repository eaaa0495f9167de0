"""One configuration of tools/exp_ddpg_wide.py (for rocprofv3): python3 tools/run_ddpg_wide.py H1 H2 BATCH [wide]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from exp_ddpg_wide import run
h1, h2, B = (int(x) for x in sys.argv[1:4])
run(h1, h2, B, wide=True if len(sys.argv) > 4 else None)
