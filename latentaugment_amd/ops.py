"""Host-side mirror of the reference op layer (models/stylegan3/torch_utils/ops/) over the HIP C ABI.

Same names, argument meaning and error behaviour as the reference's Python wrappers:
  bias_act(x, b, dim, act, alpha, gain, clamp)                      bias_act.py:52-86
  setup_filter / upfirdn2d / filter2d / upsample2d / downsample2d   upfirdn2d.py:70-387
Each op is a torch.autograd.Function whose forward AND backward are HIP launches (bias_act: every activation of the reference's
table with first- and second-order gradients; upfirdn2d: first order, which is all the latent-optimisation path uses).  torch only owns the device memory and the stream.
"""
import math

import numpy as np
import torch

from . import _lib

# activation table of the reference (bias_act.py:20-30): name -> (plugin id, default alpha, default gain, what the backward is formed from,
# whether a second derivative exists)
_ACTS = {
    'linear': (1, 0.0, 1.0, '', False), 'relu': (2, 0.0, math.sqrt(2.0), 'y', False), 'lrelu': (3, 0.2, math.sqrt(2.0), 'y', False),
    'tanh': (4, 0.0, 1.0, 'y', True), 'sigmoid': (5, 0.0, 1.0, 'y', True), 'elu': (6, 0.0, 1.0, 'y', True), 'selu': (7, 0.0, 1.0, 'y', True),
    'softplus': (8, 0.0, 1.0, 'y', True), 'swish': (9, 0.0, math.sqrt(2.0), 'x', True),
}


def _bias_act_launch(x, b, xref, yref, dy, grad, stepb, nb, act, alpha, gain, clamp):
    """One launch of the general op (include/latentaug_hip.h: la_bias_act_ex_f32 = the plugin's bias_act(x, b, xref, yref, dy, grad, ...))."""
    lib = _lib.load()
    out = torch.empty_like(x)
    _lib.check(lib.la_bias_act_ex_f32(_lib.ptr(x), _lib.ptr(b), _lib.ptr(xref), _lib.ptr(yref), _lib.ptr(dy), _lib.ptr(out), x.numel(), stepb, nb,
                                      grad, act, alpha, gain, clamp, _lib.stream_ptr()), 'bias_act')
    return out


class _BiasSum(torch.autograd.Function):
    """db = dx summed over every axis but the bias axis (bias_act.py:187,206), as a launch; its own gradient is a broadcast view."""

    @staticmethod
    def forward(ctx, dx, stepb, nb):
        lib = _lib.load()
        dx = dx.contiguous()
        db = torch.empty([nb], device=dx.device, dtype=torch.float32)
        if dx.numel():
            _lib.check(lib.la_bias_sum_f32(_lib.ptr(dx), _lib.ptr(db), dx.numel(), stepb, nb, _lib.stream_ptr()), 'bias_sum')
        else:
            db.zero_()
        ctx.meta = (tuple(dx.shape), stepb, nb)
        return db

    @staticmethod
    def backward(ctx, d_db):
        shape, stepb, nb = ctx.meta
        lead = int(np.prod(shape)) // (stepb * nb) if stepb * nb else 0
        return d_db.reshape(1, nb, 1).expand(lead, nb, stepb).reshape(shape), None, None


class _BiasAct(torch.autograd.Function):
    """Forward of the op; first- and second-order gradients as HIP launches too (the reference: bias_act.py:130-210)."""

    @staticmethod
    def forward(ctx, x, b, dim, act, alpha, gain, clamp, ref, has2):
        _lib.require_gpu(x)
        x = x.contiguous().float()
        stepb, nb = 1, 1
        if b is not None:
            assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
            b = b.contiguous().float()
            nb = x.shape[dim]
            stepb = int(np.prod(x.shape[dim + 1:])) if dim + 1 < x.ndim else 1
        y = _bias_act_launch(x, b, None, None, None, 0, stepb, nb, act, alpha, gain, clamp)
        # (y is also kept for 'linear' when a clamp is set: the reference's CPU path -- `impl='ref'`, the parity target -- masks the gradient
        #  where the clamp is active for every activation; its CUDA plugin, which saves no y for 'linear', does not)
        ctx.save_for_backward(x if (ref == 'x' or has2) else None, b if (ref == 'x' or has2) else None,
                              y if (ref == 'y' or has2 or clamp >= 0) else None)
        ctx.meta = (stepb, nb, act, alpha, gain, clamp, b is not None, has2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        stepb, nb, act, alpha, gain, clamp, has_b, has2 = ctx.meta
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = _BiasActGrad.apply(dy.contiguous(), x, b, y, ctx.meta)
        if ctx.needs_input_grad[1] and has_b:
            db = _BiasSum.apply(dx, stepb, nb)
        return dx, db, None, None, None, None, None, None, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, meta):
        stepb, nb, act, alpha, gain, clamp, has_b, has2 = meta
        dx = _bias_act_launch(dy, b, x, y, None, 1, stepb, nb, act, alpha, gain, clamp)
        ctx.save_for_backward(dy if has2 else None, x, b, y)
        ctx.meta = meta
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dy, x, b, y = ctx.saved_tensors
        stepb, nb, act, alpha, gain, clamp, has_b, has2 = ctx.meta
        d_dx = d_dx.contiguous()
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, ctx.meta)
        if has2 and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _bias_act_launch(d_dx, b, x, y, dy, 2, stepb, nb, act, alpha, gain, clamp)
            if has_b and ctx.needs_input_grad[2]:
                d_b = _BiasSum.apply(d_x, stepb, nb)
        return d_dy, d_x, d_b, None, None


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='hip'):
    """Fused bias + activation + gain + clamp (reference: bias_act.py:52-86); every activation of the reference's table, first- and
    second-order gradients."""
    assert isinstance(x, torch.Tensor)
    assert clamp is None or clamp >= 0
    if act not in _ACTS:
        raise KeyError(act)      # (the reference indexes its activation table the same way)
    idx, def_alpha, def_gain, ref, has2 = _ACTS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    return _BiasAct.apply(x, b, dim, idx, alpha, gain, clamp, ref, has2)


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """FIR taps for upfirdn2d (reference: upfirdn2d.py:70-114).  Always returned on the HOST: the HIP kernels take
    the (<= 8x8) taps as launch arguments."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if separable:
        assert f.ndim == 1      # (kept 1-D: upfirdn2d then runs one pass per axis, upfirdn2d.py:188-201)
        if f.numel() > 32:
            raise NotImplementedError('separable filters of more than 32 taps')
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return (f * (gain ** (f.ndim / 2))).cpu()


def _parse_scaling(s):
    if isinstance(s, int):
        return s, s
    sx, sy = s
    return int(sx), int(sy)


def _parse_padding(p):
    if isinstance(p, int):
        p = [p, p]
    p = list(p)
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    assert len(p) == 4
    return [int(v) for v in p]


def _launch_upfirdn2d(x, f, upx, upy, dnx, dny, px0, px1, py0, py1, flip, gain):
    lib = _lib.load()
    n, c, h, w = x.shape
    fh, fw = f.shape
    ow = lib.la_upfirdn2d_out_size(w, upx, dnx, px0, px1, fw)
    oh = lib.la_upfirdn2d_out_size(h, upy, dny, py0, py1, fh)
    assert ow >= 1 and oh >= 1
    y = torch.empty([n, c, oh, ow], device=x.device, dtype=torch.float32)
    fh_ = np.ascontiguousarray(f.numpy(), dtype=np.float32)
    _lib.check(lib.la_upfirdn2d_f32(_lib.ptr(x), fh_.ctypes.data, _lib.ptr(y), n, c, h, w, fh, fw, upx, upy, dnx, dny,
                                    px0, px1, py0, py1, int(flip), float(gain), _lib.stream_ptr()), 'upfirdn2d')
    return y


class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, up, down, padding, flip_filter, gain):
        _lib.require_gpu(x)
        x = x.contiguous().float()
        ctx.meta = (f, up, down, padding, flip_filter, gain, x.shape)
        return _launch_upfirdn2d(x, f, *up, *down, *padding, flip_filter, gain)

    @staticmethod
    def backward(ctx, dy):
        # same op with up <-> down, flipped filter and the pads of upfirdn2d.py:255-266
        f, (upx, upy), (dnx, dny), (px0, px1, py0, py1), flip, gain, xs = ctx.meta
        _, _, ih, iw = xs
        _, _, oh, ow = dy.shape
        fh, fw = f.shape
        p = [fw - px0 - 1, iw * upx - ow * dnx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * dny + py0 - upy + 1]
        dx = _launch_upfirdn2d(dy.contiguous(), f, dnx, dny, upx, upy, *p, not flip, gain)
        return dx, None, None, None, None, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='hip'):
    """Pad, upsample, filter, downsample (reference: upfirdn2d.py:118-162).  A 1-D (separable) filter runs as one pass per axis,
    each with the square root of the gain, exactly as the reference's plugin path applies it (upfirdn2d.py:188-201)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32
    (upx, upy), (dnx, dny) = _parse_scaling(up), _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    if f.ndim == 1:
        g = float(gain) ** 0.5
        fc = f.cpu()
        y = _Upfirdn2d.apply(x, fc[None, :], (upx, 1), (dnx, 1), (px0, px1, 0, 0), bool(flip_filter), g)
        return _Upfirdn2d.apply(y, fc[:, None], (1, upy), (1, dny), (0, 0, py0, py1), bool(flip_filter), g)
    return _Upfirdn2d.apply(x, f.cpu(), (upx, upy), (dnx, dny), (px0, px1, py0, py1), bool(flip_filter), float(gain))


def _fshape(f):
    """(fh, fw) of a 2-D or separable 1-D filter (upfirdn2d.py:55-66 _get_filter_size)."""
    return (f.shape[0], f.shape[0]) if f.ndim == 1 else tuple(f.shape)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='hip'):
    """reference: upfirdn2d.py:277-309"""
    px0, px1, py0, py1 = _parse_padding(padding)
    fh, fw = _fshape(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='hip'):
    """reference: upfirdn2d.py:313-348"""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fh, fw = _fshape(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='hip'):
    """reference: upfirdn2d.py:352-387"""
    dnx, dny = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fh, fw = _fshape(f)
    p = [px0 + (fw - dnx + 1) // 2, px1 + (fw - dnx) // 2, py0 + (fh - dny + 1) // 2, py1 + (fh - dny) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def l2_loss_vectorized(X, Y, compute_mean=True):
    """Pairwise squared-L2 in GEMM form (reference: util_latent_aug.py:315-361); forward only."""
    _lib.require_gpu(X)
    lib = _lib.load()
    assert X.ndim == Y.ndim and X.ndim in (2, 3, 4)
    n, m = X.shape[0], Y.shape[0]
    Xf = X.reshape(n, -1).contiguous().float()
    Yf = Y.reshape(m, -1).contiguous().float()
    K = Xf.shape[1]
    assert Yf.shape[1] == K
    D = torch.empty([m, n], device=X.device, dtype=torch.float32)
    mean = torch.empty([1], device=X.device, dtype=torch.float32)
    ws = torch.empty([lib.la_pairwise_l2_workspace_floats(n, m)], device=X.device, dtype=torch.float32)
    _lib.check(lib.la_pairwise_l2_f32(_lib.ptr(Xf), n, _lib.ptr(Yf), m, K, _lib.ptr(D), _lib.ptr(mean), _lib.ptr(ws),
                                      _lib.stream_ptr()), 'pairwise_l2')
    return mean[0] if compute_mean else D
