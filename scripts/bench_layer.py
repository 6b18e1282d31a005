"""Dev tool: time one modulated 3x3 conv layer (forward, direct C-ABI call) on the GPU.

    python scripts/bench_layer.py --res 256 --cin 128 --cout 128 --batch 8 --prec 3 [--up] [--bwd]
Runs on the DEVELOPMENT build of the library (make -C latentaugment_amd/csrc dev): kernel-variant knobs (--ab) and LA_*
environment switches exist there only.  --product times the product build instead (no --ab).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentaugment_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--cin', type=int, default=128)
    ap.add_argument('--cout', type=int, default=128)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--prec', type=int, default=3)
    ap.add_argument('--up', action='store_true')
    ap.add_argument('--bwd', action='store_true')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--ab', type=int, default=-1, help='dev knob id (la_dev_knob_set): time knob=0 against knob=1 in interleaved rounds of this process')
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--va', type=int, default=0, help='knob value of arm A')
    ap.add_argument('--vb', type=int, default=1, help='knob value of arm B')
    ap.add_argument('--product', action='store_true', help='time the product build (no knobs)')
    a = ap.parse_args()
    if not a.product:
        _lib.select_dev_build()
    lib = _lib.load()
    dev = torch.device('cuda:0')
    B, cin, cout, res = a.batch, a.cin, a.cout, a.res
    rin = res // 2 if a.up else res
    st = _lib.stream_ptr()
    x = torch.randn([B, cin, rin, rin], device=dev)
    w = torch.randn([cout, cin, 3, 3], device=dev)
    s = torch.randn([B, cin], device=dev) * 0.5 + 1
    bias = torch.randn([cout], device=dev) * 0.1
    noise = torch.randn([res, res], device=dev)
    gz = torch.randn([B, cout, res, res], device=dev)
    wf = torch.empty([9, cin, cout], device=dev)
    wb = torch.empty([9, cout, cin], device=dev)
    wsq = torch.empty([cout, cin], device=dev)
    _lib.check(lib.la_pack_conv_weights_f32(_lib.ptr(w), _lib.ptr(wf), _lib.ptr(wb), _lib.ptr(wsq), cout, cin, 9, st))
    d = torch.rsqrt((s.square() @ wsq.t()) + 1e-8).contiguous()
    wqf = wqb = None
    if a.prec:
        wqf = torch.empty([lib.la_modconv_bf16_pack_bytes(cin, cout, 0, 3)], dtype=torch.uint8, device=dev)
        wqb = torch.empty([lib.la_modconv_bf16_pack_bytes(cin, cout, 1, 3)], dtype=torch.uint8, device=dev)
        _lib.check(lib.la_pack_conv_weights_bf16_f32(_lib.ptr(w), _lib.ptr(wqf), cout, cin, 9, 0, 3, st))
        _lib.check(lib.la_pack_conv_weights_bf16_f32(_lib.ptr(w), _lib.ptr(wqb), cout, cin, 9, 1, 3, st))
    y = torch.empty([B, cout, res, res], device=dev)
    gx = torch.empty([B, cin, rin, rin], device=dev)
    dsp = torch.zeros([B, cin, lib.la_modconv_ds_tiles(rin)], device=dev)
    f = np.ascontiguousarray(np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32) / 64)
    scratch = torch.empty([B * max(cin, cout) * (res + 1) * (res + 1)], device=dev)
    skn = int(lib.la_modconv_workspace_bytes(B, cin, cout, res, 1 if a.up else 0))
    skw = torch.empty([max(skn, 1)], dtype=torch.uint8, device=dev)
    sq2 = float(np.sqrt(2))

    def run():
        if a.bwd:
            if a.up:
                rc = lib.la_modconv3x3_up2_bwd_f32(_lib.ptr(gz), _lib.ptr(wb), _lib.ptr(wqb), a.prec, _lib.ptr(s), cin, _lib.ptr(x), cin * rin * rin,
                                                   f.ctypes.data, _lib.ptr(scratch), _lib.ptr(gx), _lib.ptr(dsp), _lib.ptr(skw), skn, B, cin, cout, res, st)
            else:
                rc = lib.la_modconv3x3_bwd_f32(_lib.ptr(gz), _lib.ptr(wb), _lib.ptr(wqb), a.prec, _lib.ptr(s), cin, _lib.ptr(x), cin * rin * rin,
                                               _lib.ptr(gx), _lib.ptr(dsp), _lib.ptr(skw), skn, B, cin, cout, res, st)
        elif a.up:
            rc = lib.la_modconv3x3_up2_fwd_f32(_lib.ptr(x), cin * rin * rin, _lib.ptr(wf), _lib.ptr(wqf), a.prec, _lib.ptr(s), cin, _lib.ptr(d), cout,
                                               _lib.ptr(noise), 0, 0.1, _lib.ptr(bias), 3, 0.2, sq2, 256.0, f.ctypes.data, _lib.ptr(scratch),
                                               _lib.ptr(y), _lib.ptr(skw), skn, B, cin, cout, res, st)
        else:
            rc = lib.la_modconv3x3_fwd_f32(_lib.ptr(x), cin * rin * rin, _lib.ptr(wf), _lib.ptr(wqf), a.prec, _lib.ptr(s), cin, _lib.ptr(d), cout,
                                           _lib.ptr(noise), 0, 0.1, _lib.ptr(bias), 3, 0.2, sq2, 256.0, _lib.ptr(y), _lib.ptr(skw), skn, B, cin,
                                           cout, res, st)
        _lib.check(rc, 'layer')

    if a.ab >= 0:
        # interleaved A/B in one process (cdna_hip_programming.md rule 24): median and minimum of the per-round averages, and the
        # largest difference of the two variants' outputs
        outs, times = {}, {a.va: [], a.vb: []}
        for v in (a.va, a.vb):
            _lib.check(lib.la_dev_knob_set(a.ab, v))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            outs[v] = (gx if a.bwd else y).clone()
        dmax = float((outs[a.va] - outs[a.vb]).abs().max()) / float(outs[a.va].abs().max())
        for _ in range(a.rounds):
            for v in (a.va, a.vb):
                _lib.check(lib.la_dev_knob_set(a.ab, v))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / a.iters * 1e3)
        _lib.check(lib.la_dev_knob_set(a.ab, 0))
        m0, m1 = float(np.median(times[a.va])), float(np.median(times[a.vb]))
        print(f'res {res} {cin}->{cout} B{B} up={a.up} bwd={a.bwd} knob {a.ab}: {a.va} -> median {m0:.1f} min {min(times[a.va]):.1f} us | '
              f'{a.vb} -> median {m1:.1f} min {min(times[a.vb]):.1f} us | ratio {m1 / m0:.3f} | max |out0 - out1| / max |out0| = {dmax:.2e}', flush=True)
        return
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    flops = 2.0 * B * res * res * cin * cout * 9
    print(f'{"product" if a.product else "dev build"}: res {res} {cin}->{cout} B{B} prec {a.prec} up={a.up} bwd={a.bwd}: '
          f'{ms * 1e3:.1f} us/call, {flops / ms / 1e9:.1f} TF/s fp32-equivalent', flush=True)


if __name__ == '__main__':
    main()
