"""bench.py on a variant build of the library (experiments only):  PIVP_BENCH_LIB=<.so> python scripts/r04/bench_with_lib.py <bench.py arguments>"""
import os
import sys
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ['PIVP_BENCH_LIB'])
import bench
bench.main(sys.argv[1:])
