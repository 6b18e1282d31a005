import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from oracle import sg2_networks as nets
from latentaugment_amd.synthesis import SynthesisEngine
dev = torch.device('cuda:0')
for shrink in (2.0**-6, 2.0**-9, 2.0**-12):
    G = nets.make_generator(img_resolution=64, img_channels=2, channel_base=4096, channel_max=64, seed=3, noise_strength=0.0, w_dim=64, mapping_layers=1)
    with torch.no_grad(): G.synthesis.b4.const.mul_(shrink)
    gen = torch.Generator().manual_seed(9)
    ws = torch.randn([2, G.num_ws, 64], generator=gen)
    g_img = torch.randn([2, 2, 64, 64], generator=gen)
    res = {}
    for dt in (torch.float32, torch.float64):
        nets.COMPUTE_DTYPE = dt
        Gd = G.double() if dt == torch.float64 else G.float()
        wsr = ws.to(dt).requires_grad_(True)
        img_r = Gd.synthesis(wsr, noise_mode='const')
        (d,) = torch.autograd.grad(img_r, [wsr], g_img.to(dt))
        res[dt] = (img_r.detach().double(), d.double())
    nets.COMPUTE_DTYPE = torch.float32; G.float()
    i64, d64 = res[torch.float64]; i32, d32 = res[torch.float32]
    print(f'shrink {shrink:.1e}: oracle fp32 vs fp64: img {float((i32-i64).abs().max()/i64.abs().max()):.2e} grad {float((d32-d64).abs().max()/d64.abs().max()):.2e}')
    for prec in ('f32', 'bf16x3', 'f16x2', 'f16x2-data'):
        eng = SynthesisEngine.from_generator(G, dev, max_batch=2, precision=prec.split('-')[0])
        if prec.endswith('data'):
            eng.set_operand_scale('data')
        img = eng.forward(ws.to(dev), noise_mode='const').cpu().double()
        dws = eng.backward(g_img.to(dev)).cpu().double()
        print(f'   {prec:10s}: img err {float((img-i64).abs().max()/i64.abs().max()):.2e} grad err {float((dws-d64).abs().max()/d64.abs().max()):.2e}')
