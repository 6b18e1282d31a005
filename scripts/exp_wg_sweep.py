"""Dev tool: the stride-1 halo layers at batch 1..8 in ONE process (forward and backward), for a rocprofv3 kernel trace: the launch
grid's z is the batch, so `shape_stats.py` lists the kernel time against the number of workgroups -- how a launch's time follows the
rounds of resident workgroups (768 slots at three per CU).    rocprofv3 --kernel-trace -d D -o r -- python3 scripts/exp_wg_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentaugment_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
st = _lib.stream_ptr()
sq2 = float(np.sqrt(2))
shapes = [(256, 128), (128, 256), (64, 512)] if len(sys.argv) < 2 else [tuple(int(v) for v in s.split('x')) for s in sys.argv[1:]]
for res, ch in shapes:
    cin = cout = ch
    w = torch.randn([cout, cin, 3, 3], device=dev)
    wf = torch.empty([9, cin, cout], device=dev); wb = torch.empty([9, cout, cin], device=dev); wsq = torch.empty([cout, cin], device=dev)
    _lib.check(lib.la_pack_conv_weights_f32(_lib.ptr(w), _lib.ptr(wf), _lib.ptr(wb), _lib.ptr(wsq), cout, cin, 9, st))
    wqf = torch.empty([lib.la_modconv_bf16_pack_bytes(cin, cout, 0, 3)], dtype=torch.uint8, device=dev)
    wqb = torch.empty([lib.la_modconv_bf16_pack_bytes(cin, cout, 1, 3)], dtype=torch.uint8, device=dev)
    _lib.check(lib.la_pack_conv_weights_bf16_f32(_lib.ptr(w), _lib.ptr(wqf), cout, cin, 9, 0, 3, st))
    _lib.check(lib.la_pack_conv_weights_bf16_f32(_lib.ptr(w), _lib.ptr(wqb), cout, cin, 9, 1, 3, st))
    bias = torch.randn([cout], device=dev) * 0.1
    noise = torch.randn([res, res], device=dev)
    for B in range(1, 9):
        x = torch.randn([B, cin, res, res], device=dev)
        s = torch.randn([B, cin], device=dev) * 0.5 + 1
        d = torch.rsqrt((s.square() @ wsq.t()) + 1e-8).contiguous()
        gz = torch.randn([B, cout, res, res], device=dev)
        y = torch.empty([B, cout, res, res], device=dev); gx = torch.empty([B, cin, res, res], device=dev)
        dsp = torch.zeros([B, cin, lib.la_modconv_ds_tiles(res)], device=dev)
        skn = int(lib.la_modconv_workspace_bytes(B, cin, cout, res, 0))
        skw = torch.empty([max(skn, 1)], dtype=torch.uint8, device=dev)
        for _ in range(8):
            _lib.check(lib.la_modconv3x3_fwd_f32(_lib.ptr(x), cin * res * res, _lib.ptr(wf), _lib.ptr(wqf), 3, _lib.ptr(s), cin, _lib.ptr(d), cout,
                                                 _lib.ptr(noise), 0, 0.1, _lib.ptr(bias), 3, 0.2, sq2, 256.0, _lib.ptr(y), _lib.ptr(skw), skn, B, cin, cout, res, st))
            _lib.check(lib.la_modconv3x3_bwd_f32(_lib.ptr(gz), _lib.ptr(wb), _lib.ptr(wqb), 3, _lib.ptr(s), cin, _lib.ptr(x), cin * res * res,
                                                 _lib.ptr(gx), _lib.ptr(dsp), _lib.ptr(skw), skn, B, cin, cout, res, st))
        torch.cuda.synchronize()
print('done')
