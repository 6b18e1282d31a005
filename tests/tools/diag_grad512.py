"""Diagnostic: d/dws at the config-f 512^2 shape -- HIP (f32, bf16x3) vs the oracle in fp32 and fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latentaugment_amd import synthetic
from latentaugment_amd.synthesis import SynthesisEngine
from oracle import sg2_networks as nets
res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda', 0)
sd, meta = synthetic.make_generator_state_dict(img_resolution=res, img_channels=2, channel_base=32768, seed=0)
G = nets.Generator(img_resolution=res, img_channels=2, channel_base=32768)
G.load_state_dict(sd, strict=False); G = G.eval().requires_grad_(False)
gen = torch.Generator().manual_seed(2)
ws = torch.randn([1, meta['num_ws'], 512], generator=gen)
g_img = torch.randn([1, 2, res, res], generator=gen)
torch.set_num_threads(16)
def oracle(dtype):
    Gd = G.to(dtype)
    wsr = ws.to(dtype).requires_grad_(True)
    img = Gd.synthesis(wsr, noise_mode='const') if dtype == torch.float32 else None
    if dtype != torch.float32:
        # the oracle network casts ws to float32 internally; run the double version through a patched copy
        import oracle.sg2_networks as m
        x = img_ = None
        w_idx = 0
        S = Gd.synthesis
        for r in S.block_resolutions:
            blk = getattr(S, f'b{r}')
            bw = wsr.narrow(1, w_idx, blk.num_conv + blk.num_torgb); w_idx += blk.num_conv
            x, img_ = blk(x, img_, bw, noise_mode='const')
        img = img_
    (d,) = torch.autograd.grad(img, [wsr], g_img.to(dtype))
    return img.detach(), d.detach()
img32, d32 = oracle(torch.float32)
img64, d64 = oracle(torch.float64)
G.float()
sc = float(d64.abs().max())
print('scale max|dws|', sc, 'oracle fp32 vs fp64: max abs', float((d32.double() - d64).abs().max()), 'img', float((img32.double()-img64).abs().max()))
for prec in ('f32', 'bf16x3', 'bf16x2'):
    eng = SynthesisEngine.from_generator(sd, dev, max_batch=1, precision=prec)
    img = eng.forward(ws.to(dev), noise_mode='const')
    d = eng.backward(g_img.to(dev)).cpu()
    print(prec, 'HIP vs fp64: dws max abs', float((d.double() - d64).abs().max()), ' vs fp32 oracle', float((d - d32).abs().max()),
          '| img vs fp64', float((img.cpu().double() - img64).abs().max()))
    del eng
