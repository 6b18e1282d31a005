"""Oracle (TEST INFRASTRUCTURE): torch-CPU restatement of the reference's L0 op layer.

Each function cites the reference lines it restates (paths relative to /root/reference).  The
arithmetic is expressed with stock torch ops so autograd provides the gradients the reference gets
from its own autograd/`_ref` branches on CPU.  Pinned against golden vectors produced by running
the reference ops themselves (tests/golden/l0_ops.npz, made by tests/golden/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)

# name -> (function(x, alpha), default alpha, default gain)
# follows models/stylegan3/torch_utils/ops/bias_act.py:20-30
_ACTS = {
    'linear':   (lambda x, a: x,                          0.0, 1.0),
    'relu':     (lambda x, a: F.relu(x),                  0.0, SQRT2),
    'lrelu':    (lambda x, a: F.leaky_relu(x, a),         0.2, SQRT2),
    'tanh':     (lambda x, a: torch.tanh(x),              0.0, 1.0),
    'sigmoid':  (lambda x, a: torch.sigmoid(x),           0.0, 1.0),
    'elu':      (lambda x, a: F.elu(x),                   0.0, 1.0),
    'selu':     (lambda x, a: F.selu(x),                  0.0, 1.0),
    'softplus': (lambda x, a: F.softplus(x),              0.0, 1.0),
    'swish':    (lambda x, a: torch.sigmoid(x) * x,       0.0, SQRT2),
}


def act_defaults(act):
    """(default alpha, default gain) of an activation; bias_act.py:20-30."""
    _, a, g = _ACTS[act]
    return a, g


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """y = clamp(act(x + b) * gain); restates `_bias_act_ref`, bias_act.py:91-120."""
    fn, def_alpha, def_gain = _ACTS[act]
    alpha = def_alpha if alpha is None else float(alpha)
    gain = def_gain if gain is None else float(gain)
    if b is not None:
        assert b.ndim == 1 and b.shape[0] == x.shape[dim]
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = fn(x, alpha)
    if gain != 1.0:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def setup_filter(taps=(1, 3, 3, 1), normalize=True, flip_filter=False, gain=1.0, separable=None):
    """FIR setup; restates upfirdn2d.py:70-114.  <8 taps => dense 2-D outer product."""
    if taps is None:
        taps = 1
    f = torch.as_tensor(taps, dtype=torch.float32)
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


def _pad4(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = list(padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    assert len(padding) == 4
    return [int(p) for p in padding]


def _xy(v):
    if isinstance(v, int):
        return v, v
    vx, vy = v
    return int(vx), int(vy)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """Zero-insert, pad/crop, FIR, decimate; restates `_upfirdn2d_ref`, upfirdn2d.py:167-211.

    Output size per axis = (in*up + pad0 + pad1 - taps + down) // down  (upfirdn2d.cpp:35-36).
    """
    assert x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    n, c, h, w = x.shape
    upx, upy = _xy(up)
    dnx, dny = _xy(down)
    px0, px1, py0, py1 = _pad4(padding)
    # 1. zero insertion (each sample followed by up-1 zeros)
    if upx > 1 or upy > 1:
        z = x.new_zeros([n, c, h * upy, w * upx])
        z[:, :, ::upy, ::upx] = x
        x = z
    # 2. pad (negative = crop)
    x = F.pad(x, [px0, px1, py0, py1])
    # 3. FIR.  True convolution unless flip_filter; conv2d correlates, hence the flip.
    f = f.to(x.dtype) * (gain ** (f.ndim / 2))
    if not flip_filter:
        f = f.flip(list(range(f.ndim)))
    if f.ndim == 2:
        x = F.conv2d(x, f[None, None].expand(c, 1, -1, -1), groups=c)
    else:
        x = F.conv2d(x, f[None, None, None, :].expand(c, 1, 1, -1), groups=c)
        x = F.conv2d(x, f[None, None, :, None].expand(c, 1, -1, 1), groups=c)
    # 4. decimate
    return x[:, :, ::dny, ::dnx]


def _fsize(f):
    if f is None:
        return 1, 1
    return int(f.shape[-1]), int(f.shape[0])


def filter2d(x, f, padding=0, flip_filter=False, gain=1.0):
    """upfirdn2d.py:277-309."""
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1.0):
    """upfirdn2d.py:313-348: pads ((fw+up-1)//2, (fw-up)//2), gain * up^2."""
    upx, upy = _xy(up)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2,
         py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1.0):
    """upfirdn2d.py:352-387: pads ((fw-down+1)//2, (fw-down)//2)."""
    dnx, dny = _xy(down)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    p = [px0 + (fw - dnx + 1) // 2, px1 + (fw - dnx) // 2,
         py0 + (fh - dny + 1) // 2, py1 + (fh - dny) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d_resample.py:29-41.  flip_weight=True is correlation (what F.conv2d does)."""
    if not flip_weight and (w.shape[2] > 1 or w.shape[3] > 1):
        w = w.flip([2, 3])
    if transpose:
        return F.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups)
    return F.conv2d(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """Convolution with optional FIR up/down-sampling; restates conv2d_resample.py:46-141."""
    cout, cin_g, kh, kw = w.shape
    fw, fh = _fsize(f)
    px0, px1, py0, py1 = _pad4(padding)
    if up > 1:      # :82-86
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:    # :87-91
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2

    if kw == 1 and kh == 1 and down > 1 and up == 1:        # :94-97
        x = upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, groups=groups, flip_weight=flip_weight)
    if kw == 1 and kh == 1 and up > 1 and down == 1:        # :100-103
        x = _conv(x, w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:                                # :106-109
        x = upfirdn2d(x, f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:                                              # :112-129
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * cin_g, cout // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True,
                  flip_weight=(not flip_weight))
        x = upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2,
                      flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:  # :132-134
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)
    # generic fallback :137-141
    x = upfirdn2d(x, f if up > 1 else None, up=up, padding=[px0, px1, py0, py1], gain=up ** 2,
                  flip_filter=flip_filter)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return x


def fma(a, b, c):
    """a*b+c with broadcasting; gradients via autograd equal ops/fma.py:20-58."""
    return a * b + c


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None,
                     demodulate=True, flip_weight=True, fused_modconv=True):
    """SG2 modulated convolution (public NVlabs definition).

    fused: per-sample weights w''[b,o,i,k] = W*s*d through a grouped conv (groups=B) -- the form the
    reference executes in eval mode.  non-fused: x*s -> shared-weight conv -> *d (+noise).  Equal in
    exact arithmetic; the HIP path implements the non-fused form.
    """
    B = x.shape[0]
    cout, cin, kh, kw = weight.shape
    w = None
    dcoefs = None
    if demodulate or fused_modconv:
        w = weight[None] * styles.reshape(B, 1, cin, 1, 1)
    if demodulate:
        dcoefs = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    if not fused_modconv:
        x = x * styles.reshape(B, cin, 1, 1)
        x = conv2d_resample(x, weight, f=resample_filter, up=up, down=down, padding=padding,
                            flip_weight=flip_weight)
        if demodulate and noise is not None:
            x = fma(x, dcoefs.reshape(B, cout, 1, 1), noise)
        elif demodulate:
            x = x * dcoefs.reshape(B, cout, 1, 1)
        elif noise is not None:
            x = x + noise
        return x
    if demodulate:
        w = w * dcoefs.reshape(B, cout, 1, 1, 1)
    x = x.reshape(1, B * cin, *x.shape[2:])
    x = conv2d_resample(x, w.reshape(B * cout, cin, kh, kw), f=resample_filter, up=up, down=down,
                        padding=padding, groups=B, flip_weight=flip_weight)
    x = x.reshape(B, cout, *x.shape[2:])
    if noise is not None:
        x = x + noise
    return x
