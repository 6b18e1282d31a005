"""Dev tool: bench.py on ANOTHER build of the library (same-box A/B of two .so files in one gpurun call):
    python scripts/bench_with_lib.py tmp_libs/base/liblatentaug_hip.so --steps 20 --warmup 5 --no-cpu-baseline
The library path replaces latentaugment_amd._lib.LIB_PATH before anything loads it; symbols the other build lacks are dropped from the
binding table (an older build has no la_noise_normal_f32: the bench workload does not call it)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
import torch  # noqa: E402,F401  (its HIP runtime first, as _lib.load() does)
from latentaugment_amd import _lib  # noqa: E402
_lib.LIB_PATH = lib_path
probe = ctypes.CDLL(lib_path)
for k in list(_lib.SIGNATURES):
    if not hasattr(probe, k):
        del _lib.SIGNATURES[k]
import bench  # noqa: E402
bench.main()
