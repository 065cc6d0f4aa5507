import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_simple_amd import api
lib = api.load_library(diag=True); lib.rt_debug_sqrt_mismatches.restype = C.c_longlong
t = time.time(); print("lean sqrt mismatches over 2^32 inputs:", lib.rt_debug_sqrt_mismatches(), "in %.2f s" % (time.time() - t))
