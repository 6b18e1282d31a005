"""Dev tool: run one layer with the LA_STAMP build and print the per-phase cycle breakdown of the halo kernel."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

os.environ['LATENTAUG_HIP_LIB'] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build_variants', 'lib_stamp.so')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], '--iters', '1'] + sys.argv[1:]
import scripts.bench_layer as bl  # noqa: E402

bl.main()
from latentaugment_amd import _lib  # noqa: E402
lib = _lib._lib
n = 4096 * 48
buf = np.zeros([n], dtype=np.uint64)
lib.la_debug_stamps.argtypes = [C.c_void_p, C.c_int]
lib.la_debug_stamps.restype = C.c_int
assert lib.la_debug_stamps(buf.ctypes.data, n) == 0
st = buf.reshape(4096, 48).astype(np.int64)
d = np.diff(st[:, :46], axis=1)
names = ['setup+issue prologue loads', 'prologue LDS writes', 'prologue barrier'] + [f'c{c} t{t}' if t < 9 else f'c{c} tail' for c in range(4) for t in range(10)] + ['scale', 'epilogue']
# stamps 3 -> 4 is chunk0 tap0, etc.  (s_memtime ticks at 100 MHz)
print('ticks are s_memtime units; median / mean over workgroups')
for k in range(45):
    print(f'{names[k]:30s} {np.median(d[:, k]):8.0f} {d[:, k].mean():9.1f}')
tot = st[:, 45] - st[:, 0]
print('workgroup lifetime median', np.median(tot), 'mean', tot.mean())
