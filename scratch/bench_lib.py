"""bench.py against an alternative build of the library: TACORL_SCRATCH_LIB=<path> python scratch/bench_lib.py [bench args]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tacorl_amd import _lib
if os.environ.get("TACORL_SCRATCH_LIB"): _lib.LIB_PATH = os.environ["TACORL_SCRATCH_LIB"]
import bench
bench.main()
