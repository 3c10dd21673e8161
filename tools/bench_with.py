"""Run bench.py with libvitmi debug switches set first (A/B of a kernel option inside the real step).
usage: python tools/bench_with.py hook=value [hook=value ...] -- [bench.py args]
e.g.   python tools/bench_with.py gemm_store_policy=0 gemm_band=0 -- --lean --no-cpu-baseline"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cut = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
hooks, rest = sys.argv[1:cut], sys.argv[cut + 1:]
from vit_torch_amd import _lib  # noqa: E402

raw = ctypes.CDLL(str(_lib.LIB_PATH))
for h in hooks:
    name, val = h.split("=")
    getattr(raw, "vitmi_debug_" + name)(int(val))
sys.argv = [os.path.join(ROOT, "bench.py")] + rest
import bench  # noqa: E402

bench.main()
