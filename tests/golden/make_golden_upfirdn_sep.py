#!/usr/bin/env python3
"""Golden vectors of upfirdn2d with SEPARABLE (1-D, >= 8 tap) filters and of non-contiguous inputs, by RUNNING THE REFERENCE here.

    python tests/golden/make_golden_upfirdn_sep.py     ->  tests/golden/upfirdn_sep.npz

Executed from the reference (imported, never copied): models/stylegan3/torch_utils/ops/upfirdn2d.py -- setup_filter (:70-114, which
keeps a filter of >= 8 taps one-dimensional), upfirdn2d / upsample2d / downsample2d / filter2d with impl='ref' (:118-211, :277-387),
forward and the gradient with respect to the input (autograd through the reference implementation)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference/models/stylegan3')
from torch_utils.ops import upfirdn2d as ref      # noqa: E402

torch.manual_seed(11)
out, cases = {}, []
taps8 = [1, 7, 21, 35, 35, 21, 7, 1]                      # binomial, 8 taps -> separable by setup_filter's rule
taps12 = list(np.hanning(14)[1:-1])                       # 12 taps
k = 0
for taps, tname in ((taps8, 'binom8'), (taps12, 'hann12')):
    for op, kw in (('upfirdn2d', dict(up=1, down=1, padding=[3, 4, 2, 5])), ('upfirdn2d', dict(up=2, down=1, padding=[5, 4, 5, 4], gain=4)),
                   ('upfirdn2d', dict(up=1, down=2, padding=[4, 3, 4, 3], flip_filter=True)), ('upsample2d', dict(up=2)),
                   ('downsample2d', dict(down=2)), ('filter2d', dict())):
        f = ref.setup_filter(taps)
        assert f.ndim == 1
        x = torch.randn([1, 2, 13, 16]).double().requires_grad_(True)
        y = getattr(ref, op)(x, f, impl="ref", **kw)
        dy = torch.randn(y.shape).double()
        (dx,) = torch.autograd.grad(y, [x], dy)
        name = f's{k}'
        cases.append((name, tname, op, repr(kw)))
        out[f'{name}_taps'] = np.asarray(taps, dtype=np.float64)
        for key, t in (('x', x.detach()), ('dy', dy), ('y', y.detach()), ('dx', dx)):
            out[f'{name}_{key}'] = t.numpy().astype(np.float32 if key in ('x', 'dy') else np.float64)
        k += 1
out['cases'] = np.array([repr(c) for c in cases])
np.savez_compressed(os.path.join(HERE, 'upfirdn_sep.npz'), **out)
print(k, 'cases')
