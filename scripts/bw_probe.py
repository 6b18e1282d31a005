"""dev tool: achievable HBM rates of plain torch ops on this box (calibration for the HBM-class kernels)."""
import torch
dev = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (268, 1074):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    # a few more buffers so that consecutive calls do not hit the 256 MB memory-side cache
    xs = [torch.randn(n, device=dev) for _ in range(4)]; ys = [torch.empty_like(x) for _ in range(4)]
    i = [0]
    def copy():
        k = i[0] % 4; i[0] += 1
        ys[k].copy_(xs[k])
    def read():
        k = i[0] % 4; i[0] += 1
        return xs[k].sum()
    def write():
        k = i[0] % 4; i[0] += 1
        ys[k].fill_(1.0)
    us = t(copy); print(f'copy  {mb} MB -> {mb} MB: {us:.1f} us, {2 * n * 4 / us / 1e6:.2f} TB/s (read + write)')
    us = t(read); print(f'read  {mb} MB: {us:.1f} us, {n * 4 / us / 1e6:.2f} TB/s')
    us = t(write); print(f'write {mb} MB: {us:.1f} us, {n * 4 / us / 1e6:.2f} TB/s')
    del x, y, xs, ys
