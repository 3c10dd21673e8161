"""Run bench.py in-process with a vitmi_debug_* hook set first: bench_hook.py <hook> <int> [bench args]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import _lib
_lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
getattr(raw, sys.argv[1])(int(sys.argv[2]))
sys.argv = ["bench.py"] + sys.argv[3:]
import bench
bench.main()
