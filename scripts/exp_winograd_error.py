"""dev experiment (CPU): layer error of a Winograd F(2x2,3x3) contraction with the f16x2 operand split, against the direct
f16x2 contraction and a plain float32 convolution, all measured against float64."""
import numpy as np, torch
torch.manual_seed(0)
torch.set_num_threads(8)
C, M, H = 128, 128, 64
x = torch.nn.functional.leaky_relu(torch.randn(1, C, H, H, dtype=torch.float64), 0.2) * np.sqrt(2)
s = 1 + 0.3 * torch.randn(1, C, 1, 1, dtype=torch.float64)
xm = (x * s)
w = torch.randn(M, C, 3, 3, dtype=torch.float64)
ref = torch.nn.functional.conv2d(xm, w, padding=1)
def err(y): return float(((y.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
def pow2(v):  # scale so that max in [2^14, 2^15)
    return 2.0 ** (14 - np.floor(np.log2(float(v.abs().max()))))
def split(v32):
    h = v32.half(); l = (v32 - h.float()).half()
    return h.float(), l.float()
x32, w32 = xm.float(), w.float()
print('fp32 direct          ', err(torch.nn.functional.conv2d(x32, w32, padding=1)))
sx, sw = pow2(x32), pow2(w32)
xh, xl = split(x32 * sx); wh, wl = split(w32 * sw)
conv = lambda a, b: torch.nn.functional.conv2d(a, b, padding=1)
y = (conv(xl, wh) + conv(xh, wl) + conv(xh, wh)) / (sx * sw)
print('f16x2 direct         ', err(y))
# Winograd F(2x2,3x3)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
U = torch.einsum('ai,mcij,bj->mcab', G, w, G)               # float64 at pack time
xp = torch.nn.functional.pad(x32, (1, 1, 1, 1))
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                   # [1,C,T,T,4,4]
V = torch.einsum('ai,nctuij,bj->nctuab', Bt.float(), tiles, Bt.float())   # float32 adds
def wino(Uq, Vq, dt):
    Mx = torch.einsum('mcab,nctuab->nmtuab', Uq.to(dt), Vq.to(dt))
    Y = torch.einsum('ia,nmtuab,jb->nmtiuj', At.to(dt), Mx, At.to(dt))
    return Y.reshape(1, M, H, H)
print('winograd fp64 (check)', err(wino(U, torch.einsum('ai,nctuij,bj->nctuab', Bt, xp.double().unfold(2, 4, 2).unfold(3, 4, 2), Bt), torch.float64)))
print('winograd fp32        ', err(wino(U.float(), V, torch.float32)))
su, sv = pow2(U.float()), pow2(V)
Uh, Ul = split(U.float() * su); Vh, Vl = split(V * sv)
Mx = (torch.einsum('mcab,nctuab->nmtuab', Ul, Vh) + torch.einsum('mcab,nctuab->nmtuab', Uh, Vl) + torch.einsum('mcab,nctuab->nmtuab', Uh, Vh)) / (su * sv)
Y = torch.einsum('ia,nmtuab,jb->nmtiuj', At.float(), Mx, At.float()).reshape(1, M, H, H)
print('winograd f16x2       ', err(Y))
