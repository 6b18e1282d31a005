"""Dev tool: per-ws-slot error of d(img.g)/d(ws) at 1024^2 (HIP per precision, and the CPU fp32 oracle) against the fp64 oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_hip_synthesis import _oracle_grad
from oracle import sg2_networks as nets
from latentaugment_amd import synthetic
from latentaugment_amd.synthesis import SynthesisEngine
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device('cuda', 0)
sd, meta = synthetic.make_generator_state_dict(img_resolution=res, img_channels=2, channel_base=32768, seed=0)
G = nets.Generator(img_resolution=res, img_channels=2, channel_base=32768)
G.load_state_dict(sd, strict=False); G = G.eval().requires_grad_(False)
gen = torch.Generator().manual_seed(2)
nws = meta['num_ws']
ws = torch.randn([1, 1, 512], generator=gen).repeat(1, nws, 1)
# image gradient of the pixel criterion's form (smooth in the image), not white noise
g_img = torch.randn([1, 2, res, res], generator=gen)
torch.set_num_threads(16)
if os.environ.get('SMOOTH'):
    with torch.no_grad():
        g_img = G.synthesis(ws, noise_mode='const').clone()
    cc = int((res * res / 2) ** 0.5); off = round((res - cc) / 2)
    m = torch.zeros_like(g_img); m[:, :, off:off + cc, off:off + cc] = 1
    g_img = g_img * m
img32, d32 = _oracle_grad(G, ws, g_img, torch.float32)
nets.COMPUTE_DTYPE = torch.float64
img64, d64 = _oracle_grad(G, ws, g_img, torch.float64)
nets.COMPUTE_DTYPE = torch.float32
G.float()
e_ref = (d32.double() - d64).abs()
print('slot  |d64|rms   ref32 err rms   ' + '  '.join(sys.argv[2:]))
outs = {}
for prec in sys.argv[2:]:
    eng = SynthesisEngine.from_generator(sd, dev, max_batch=1, precision=prec)
    img = eng.forward(ws.to(dev), noise_mode='const').cpu()
    outs[prec] = (eng.backward(g_img.to(dev)).cpu().double() - d64).abs()
    print(prec, 'img err vs 64:', float((img.double() - img64).abs().max()), 'ref32:', float((img32.double() - img64).abs().max()))
for k in range(nws):
    row = f'{k:3d}  {float(d64[0, k].pow(2).mean().sqrt()):.3e}  {float(e_ref[0, k].pow(2).mean().sqrt()):.3e}   '
    row += '  '.join(f'{float(outs[p][0, k].pow(2).mean().sqrt()):.3e}' for p in sys.argv[2:])
    print(row)
print('sum over slots (W space): ref', float((d32.double().sum(1) - d64.sum(1)).pow(2).mean().sqrt()),
      ' '.join(f'{p} {float(((outs[p] * 0 + 0)).sum())}' for p in sys.argv[2:]))
