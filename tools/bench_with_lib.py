"""bench.py against another build of the library: python tools/bench_with_lib.py <lib.so> [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smartstartcontinuous_amd._ffi as F
F.LIB_PATH = os.path.join(ROOT, sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
