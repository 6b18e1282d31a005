#!/usr/bin/env python3
"""Golden vectors of the FULL bias_act op (all nine activations, grad 0 / 1 / 2) by RUNNING THE REFERENCE here.

    python tests/golden/make_golden_bias_act.py        ->  tests/golden/bias_act_full.npz

Executed from the reference (imported, never copied): models/stylegan3/torch_utils/ops/bias_act.py -- `bias_act(..., impl='ref')`
(:52-120) with autograd on top of it: the first derivative is what the plugin computes with grad = 1 (bias_act.cpp:32, the dx of
BiasActCudaGrad.forward), the derivative of <dx, d_dx> with respect to x what it computes with grad = 2 (BiasActCudaGrad.backward's
d_x).  The CUDA plugin itself cannot be built here (nvcc absent); the reference's own tests hold no fixture for this op."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference/models/stylegan3')
from torch_utils.ops import bias_act as ref      # noqa: E402

torch.manual_seed(7)
out = {}
cases = []
acts = list(ref.activation_funcs.keys())
k = 0
for act in acts:
    spec = ref.activation_funcs[act]
    for gain, clamp, alpha in ((None, None, None), (0.7, 1.1, 0.3)):
        # (inputs are float32 values; the reference runs on them in float64, so the goldens carry no rounding of their own)
        x = (torch.randn([2, 6, 5, 4]) * 2.0).double()
        b = (torch.randn([6]) * 0.5).double()
        dy = torch.randn([2, 6, 5, 4]).double()
        ddx = torch.randn([2, 6, 5, 4]).double()
        xr = x.clone().requires_grad_(True)
        y = ref.bias_act(xr, b, dim=1, act=act, alpha=alpha, gain=gain, clamp=clamp, impl='ref')
        (dx,) = torch.autograd.grad(y, [xr], dy, create_graph=True)
        d2 = torch.zeros_like(x)
        if dx.requires_grad:
            (d2,) = torch.autograd.grad(dx, [xr], ddx, allow_unused=True)
            d2 = torch.zeros_like(x) if d2 is None else d2
        name = f'c{k}'
        cases.append((name, act, -1.0 if gain is None else gain, -1.0 if clamp is None else clamp, -1.0 if alpha is None else alpha,
                      float(spec.def_alpha), float(spec.def_gain), int(spec.cuda_idx), str(spec.ref), bool(spec.has_2nd_grad)))
        for key, t in (('x', x), ('b', b), ('dy', dy), ('ddx', ddx), ('y', y.detach()), ('dx', dx.detach()), ('d2', d2.detach())):
            out[f'{name}_{key}'] = t.numpy().astype(np.float32 if key in ('x', 'b', 'dy', 'ddx') else np.float64)
        k += 1
out['cases'] = np.array([repr(c) for c in cases])
np.savez_compressed(os.path.join(HERE, 'bias_act_full.npz'), **out)
print(f'{k} cases -> bias_act_full.npz', os.path.getsize(os.path.join(HERE, 'bias_act_full.npz')), 'bytes')
