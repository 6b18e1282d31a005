"""Two-stream non-reproducibility, narrowed to the column-planar up-layer path: a 16x16 generator (last up layer = 8 -> 16), engine A's
transposed-conv intermediate zT and layer output after an overlapped pass against a solo pass -- is zT already different, or only the FIR output?"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentaugment_amd import _lib                                          # noqa: E402
_lib.select_dev_build()
from latentaugment_amd import synthetic                                     # noqa: E402
from latentaugment_amd.synthesis import SynthesisEngine                     # noqa: E402

RES = int(os.environ.get('RES', '16'))
dev = torch.device('cuda', 0)
sd, meta = synthetic.make_generator_state_dict(img_resolution=RES, img_channels=2, channel_base=32768, seed=0)
big, _ = synthetic.make_generator_state_dict(img_resolution=256, img_channels=2, channel_base=32768, seed=0)
w0 = synthetic.make_latents(8, seed=1).to(dev)
lib = _lib.load()
lib.la_synth_dev_zt.restype = C.c_void_p
lib.la_synth_dev_zt.argtypes = [C.c_void_p]
other = SynthesisEngine.from_generator(big, dev, 4)          # the disturbing work on the second stream: full 256^2 passes
third = SynthesisEngine.from_generator(sd, dev, 8)
third.forward(w0.repeat(1, third.num_ws, 1), noise_mode='const'); torch.cuda.synchronize()
ea = SynthesisEngine.from_generator(sd, dev, 4)
wa = w0[:4].repeat(1, ea.num_ws, 1).contiguous()
wo = w0[4:].repeat(1, other.num_ws, 1).contiguous()
nl = len(ea.layer_resolutions)
C_last = ea.channels[-1]
h = RES // 2
xhalf = (h + 1 + 3) & ~3
nzt = 4 * C_last * (RES + 1) * 2 * xhalf


def zt():
    p = lib.la_synth_dev_zt(ea._h)
    return ea._view(p, [4, C_last, RES + 1, 2 * xhalf]).clone()


ea.forward(wa, noise_mode='const'); torch.cuda.synchronize()
y_solo, zt_solo = ea.layer_output(nl - 2, 4).clone(), zt()
other.forward(wo, noise_mode='const'); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for rep in range(8):
    with torch.cuda.stream(s2):
        for _ in range(int(os.environ.get('OTHER_PASSES', '12'))):
            other.forward(wo, noise_mode='const')
    with torch.cuda.stream(s1):
        for _ in range(int(os.environ.get('A_PASSES', '20'))):
            ea.forward(wa, noise_mode='const')
    torch.cuda.synchronize()
    y, z = ea.layer_output(nl - 2, 4), zt()
    dz = (z != zt_solo)
    # only written positions of zT matter: even columns [0, h+1), odd columns [xhalf, xhalf + h)
    mask = torch.zeros_like(dz); mask[..., :h + 1] = True; mask[..., xhalf:xhalf + h] = True
    dzw = dz & mask
    dy = (y != y_solo)
    print(f'rep {rep}: zT written positions differing {int(dzw.sum())} (unwritten positions differing {int((dz & ~mask).sum())}), '
          f'up-layer output elements differing {int(dy.sum())}; columns of the output hit: {dy.any(dim=(0, 1, 2)).nonzero().flatten().tolist()}', flush=True)
    if int(dzw.sum()):
        idx = dzw.nonzero()[:5].tolist()
        print('    zT first differing (b, c, row, planar col):', idx, flush=True)
