"""Dev tool: first synthesis layer whose output diverges from the oracle (per-layer max error)."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import sg2_networks as nets
from latentaugment_amd.synthesis import SynthesisEngine
from latentaugment_amd import _lib
dev = torch.device('cuda', 0)
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'
G = nets.make_generator(img_resolution=64, img_channels=2, channel_base=2048, channel_max=64, seed=3, noise_strength=0.1, w_dim=64, mapping_layers=1)
eng = SynthesisEngine.from_generator(G, dev, max_batch=3, precision=prec)
gen = torch.Generator().manual_seed(9)
ws = torch.randn([3, G.num_ws, 64], generator=gen)
img_r, feats = G.synthesis(ws, noise_mode='const', return_features=True)
img = eng.forward(ws.to(dev), noise_mode='const')
lib = _lib.load()
lib.la_synth_layer_output.restype = C.c_void_p
print('feats', len(feats), 'layers', eng.num_layers)
class _P:
    pass
for k, f in enumerate(feats):
    n = f.numel()
    p = lib.la_synth_layer_output(eng.handle, 2 * k)
    torch.cuda.synchronize()
    o = _P(); o.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (p, False), 'version': 2}
    y = torch.as_tensor(o, device=dev).reshape(f.shape).cpu()
    print('block', k, tuple(f.shape), 'conv1 output max err', float((y - f).abs().max()), 'max ref', float(f.abs().max()))
print('img err', float((img.cpu() - img_r).abs().max()))
