"""Dev tool: where a wave of the 128-row (MF 5) halo kernel spends its life.  The development build stamps s_memtime per segment in every wave
(la_conv_bf16.hip, LA_STAMP; dev knob LA_KNOB_HALO_STAMP) -- prologue issue / prologue wait + first stage / tap loops / chunk barriers /
accumulator hand-over / epilogue -- and this script runs ONE stride-1 layer call with the knob on and prints the distribution per segment.
    python scripts/halo_wave_timeline.py --res 256 --ch 128 [--batch 8] [--bwd]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentaugment_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--res', type=int, default=256)
ap.add_argument('--ch', type=int, default=128)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--bwd', action='store_true')
ap.add_argument('--stagger', type=int, default=0, help='dev knob 7: -1 = no phase stagger, 0 = the host estimate, else cycles per wave slot')
ap.add_argument('--ldspad', type=int, default=0, help='extra KB of LDS per workgroup (30: two workgroups per CU, 100: one)')
a = ap.parse_args()
_lib.select_dev_build()
lib = _lib.load()
lib.la_dev_dbg_read.restype = ctypes.c_int
lib.la_dev_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_long]
dev = torch.device('cuda:0')
st = _lib.stream_ptr()
B, cin, cout, res = a.batch, a.ch, a.ch, a.res
sq2 = float(np.sqrt(2))
w = torch.randn([cout, cin, 3, 3], device=dev)
wf = torch.empty([9, cin, cout], device=dev); wb = torch.empty([9, cout, cin], device=dev); wsq = torch.empty([cout, cin], device=dev)
_lib.check(lib.la_pack_conv_weights_f32(_lib.ptr(w), _lib.ptr(wf), _lib.ptr(wb), _lib.ptr(wsq), cout, cin, 9, st))
wqf = torch.empty([lib.la_modconv_bf16_pack_bytes(cin, cout, 0, 3)], dtype=torch.uint8, device=dev)
wqb = torch.empty([lib.la_modconv_bf16_pack_bytes(cin, cout, 1, 3)], dtype=torch.uint8, device=dev)
_lib.check(lib.la_pack_conv_weights_bf16_f32(_lib.ptr(w), _lib.ptr(wqf), cout, cin, 9, 0, 3, st))
_lib.check(lib.la_pack_conv_weights_bf16_f32(_lib.ptr(w), _lib.ptr(wqb), cout, cin, 9, 1, 3, st))
bias = torch.randn([cout], device=dev) * 0.1
noise = torch.randn([res, res], device=dev)
x = torch.randn([B, cin, res, res], device=dev)
s = torch.randn([B, cin], device=dev) * 0.5 + 1
d = torch.rsqrt((s.square() @ wsq.t()) + 1e-8).contiguous()
gz = torch.randn([B, cout, res, res], device=dev)
y = torch.empty([B, cout, res, res], device=dev); gx = torch.empty([B, cin, res, res], device=dev)
dsp = torch.zeros([B, cin, lib.la_modconv_ds_tiles(res)], device=dev)
skn = int(lib.la_modconv_workspace_bytes(B, cin, cout, res, 0))
skw = torch.empty([max(skn, 1)], dtype=torch.uint8, device=dev)


def run():
    if a.bwd:
        _lib.check(lib.la_modconv3x3_bwd_f32(_lib.ptr(gz), _lib.ptr(wb), _lib.ptr(wqb), 3, _lib.ptr(s), cin, _lib.ptr(x), cin * res * res,
                                             _lib.ptr(gx), _lib.ptr(dsp), _lib.ptr(skw), skn, B, cin, cout, res, st))
    else:
        _lib.check(lib.la_modconv3x3_fwd_f32(_lib.ptr(x), cin * res * res, _lib.ptr(wf), _lib.ptr(wqf), 3, _lib.ptr(s), cin, _lib.ptr(d), cout,
                                             _lib.ptr(noise), 0, 0.1, _lib.ptr(bias), 3, 0.2, sq2, 256.0, _lib.ptr(y), _lib.ptr(skw), skn, B, cin, cout, res, st))


for _ in range(3):
    run()
torch.cuda.synchronize()
lib.la_dev_knob_set(5, 1)
lib.la_dev_knob_set(6, a.ldspad)
lib.la_dev_knob_set(7, a.stagger)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); run(); t1.record()
torch.cuda.synchronize()
lib.la_dev_knob_set(5, 0)
lib.la_dev_knob_set(6, 0)
lib.la_dev_knob_set(7, 0)
tiles = (res // 4) * (res // 32)
nw = tiles * ((cout + 127) // 128) * B * 4
buf = np.zeros([nw * 16], dtype=np.uint64)
assert lib.la_dev_dbg_read(buf.ctypes.data, buf.size) == 0
v = buf.reshape(nw, 16).astype(np.float64)
seg = v[:, :9]
names = ['prologue: loads issued, factor table', 'prologue: wait + first stage + barrier', 'tap loops (all chunks)', 'chunk barriers',
         'accumulator hand-over', 'epilogue: tail (partials / scale slot) -> end', 'epilogue: entry -> row tables requested',
         'epilogue: barrier behind the row tables', 'epilogue: value loop (loads, arithmetic, stores)']
end = v[:, 12]; start = end - seg.sum(1)
print(f'{"backward" if a.bwd else "forward"} {cin}->{cout} @{res}^2 batch {B}: {nw // 4} workgroups, layer call {t0.elapsed_time(t1) * 1e3:.1f} us (stamped)')
tot = seg.sum(1)
print(f'wave lifetime: mean {tot.mean():.0f}  median {np.median(tot):.0f}  p10 {np.percentile(tot, 10):.0f}  p90 {np.percentile(tot, 90):.0f} ticks; '
      f'first start to last end {end.max() - start.min():.0f} ticks')
for i, n in enumerate(names):
    c = seg[:, i]
    print(f'  {n:42s} mean {c.mean():8.0f} ({100 * c.mean() / tot.mean():4.1f} %)  median {np.median(c):8.0f}  p10 {np.percentile(c, 10):8.0f}  p90 {np.percentile(c, 90):8.0f}')

# how many waves of a SIMD are in their tap loops at a time: (xcc, se, cu, simd) from HW_ID / XCC_ID, intervals from the stamps
hw = v[:, 13].astype(np.int64); xcc = v[:, 14].astype(np.int64) & 0xf
key = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 8) & 15) << 8) | ((hw >> 4) & 3)
print('wave slots seen (HW_ID & 15):', np.bincount(hw & 15)[:12], ' SIMDs seen:', len(np.unique(key)))
loop0 = start + seg[:, 0] + seg[:, 1]; loop1 = loop0 + seg[:, 2] + seg[:, 3]
hist = np.zeros(8); alive = np.zeros(8)
for k in np.unique(key)[:256]:
    m = key == k
    ev = sorted([(t, 1) for t in loop0[m]] + [(t, -1) for t in loop1[m]])
    n = 0; last = ev[0][0]
    for t, d in ev:
        hist[min(n, 7)] += t - last; last = t; n += d
    ev = sorted([(t, 1) for t in start[m]] + [(t, -1) for t in end[m]])
    n = 0; last = ev[0][0]
    for t, d in ev:
        alive[min(n, 7)] += t - last; last = t; n += d
print('share of the time with k waves of a SIMD in their tap loops, k = 0..4:', np.round(hist[:5] / hist.sum(), 3))
print('share of the time with k waves resident on a SIMD,            k = 0..4:', np.round(alive[:5] / alive.sum(), 3))
