"""Dev tool: s_memtime phase breakdown of the LAST contraction launch of one layer call (LA_STAMP build)."""
import ctypes as C
import os
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
live = st[:, 45] > 0
st = st[live]
print('workgroups stamped', st.shape[0])
pro = st[:, 1] - st[:, 0]
steps = np.diff(st[:, 1:42], axis=1)
steps = np.where(steps > 0, steps, np.nan)
print('prologue median', np.median(pro))
print('step medians', np.nanmedian(steps, axis=0)[:40].round())
print('epilogue median', np.median(st[:, 45] - st[:, 44]), ' lifetime median', np.median(st[:, 45] - st[:, 0]))
