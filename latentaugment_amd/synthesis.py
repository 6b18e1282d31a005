"""Host wrapper of the HIP synthesis engine: the `G.synthesis(ws, noise_mode=...)` the reference calls at
augments/utils/util_latent_aug.py:227,488, plus its backward to ws.

`SynthesisEngine.from_generator(G)` accepts any module (or state_dict) that carries the reference's parameter names
(models/stylegan3/legacy.py:171-203: synthesis.b{res}.{const, conv0.*, conv1.*, torgb.*}) -- i.e. the G_ema of a
network pickle -- copies the tensors to the device once and hands their pointers to the C ABI.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

_CONV_KEYS = ('affine.weight', 'affine.bias', 'weight', 'bias', 'noise_const')
_RGB_KEYS = ('affine.weight', 'affine.bias', 'weight', 'bias')
NOISE_MODES = {'none': 0, 'const': 1, 'random': 2}
PRECISIONS = {'f32': 0, 'bf16x3': 1, 'bf16x2': 2, 'f16x2': 3}


def _state_dict_of(G):
    sd = G if isinstance(G, dict) else G.state_dict()
    if any(k.startswith('synthesis.') for k in sd):
        sd = {k[len('synthesis.'):]: v for k, v in sd.items() if k.startswith('synthesis.')}
    return sd


class SynthesisEngine:
    def __init__(self, state_dict, device, max_batch, conv_clamp=256.0, precision='f32'):
        lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.LatentAugHipError('SynthesisEngine needs a ROCm device (no CPU fallback)')
        sd = state_dict
        res_list = sorted({int(k.split('.')[0][1:]) for k in sd if k.startswith('b')})
        assert res_list and res_list[0] == 4, 'expected synthesis blocks b4..bR'
        self.img_resolution = res_list[-1]
        self.block_resolutions = res_list
        self.channels = [int(sd[f'b{r}.conv1.weight'].shape[0]) for r in res_list]
        self.img_channels = int(sd[f'b{res_list[-1]}.torgb.weight'].shape[0])
        self.w_dim = int(sd['b4.conv1.affine.weight'].shape[1])
        self.num_ws = lib.la_synth_num_ws(self.img_resolution)
        self.max_batch = int(max_batch)
        self.conv_clamp = float(-1.0 if conv_clamp is None else conv_clamp)

        def dev(name):
            t = sd[name].detach().to(device=self.device, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t

        self._keep = []
        params = []
        strengths = []
        self.layer_resolutions = []
        for r in res_list:
            if r == 4:
                params.append(dev('b4.const'))
            layers = ('conv1',) if r == 4 else ('conv0', 'conv1')
            for ln in layers:
                for k in _CONV_KEYS:
                    params.append(dev(f'b{r}.{ln}.{k}'))
                strengths.append(float(sd[f'b{r}.{ln}.noise_strength']))
                self.layer_resolutions.append(r)
            for k in _RGB_KEYS:
                t = dev(f'b{r}.torgb.{k}')
                if k == 'weight':
                    t = t.reshape(t.shape[0], t.shape[1]).contiguous()   # [imgc][cin][1][1] -> [imgc][cin]
                    self._keep.append(t)
                params.append(t)
        fkey = f'b{res_list[-1]}.resample_filter'
        fir = sd[fkey].detach().cpu().float().numpy() if fkey in sd else None
        if fir is None:
            f1 = np.array([1, 3, 3, 1], dtype=np.float32)
            fir = np.outer(f1, f1) / 64.0
        assert fir.shape == (4, 4), 'resample filter must be the 4x4 setup_filter([1,3,3,1])'
        self._fir = np.ascontiguousarray(fir, dtype=np.float32)
        self.noise_strengths = strengths
        self.num_layers = len(strengths)

        chan = (C.c_int * len(self.channels))(*self.channels)
        nbytes = lib.la_synth_workspace_bytes(self.img_resolution, self.img_channels, self.w_dim, chan, self.max_batch)
        assert nbytes > 0
        self._workspace = torch.empty([nbytes], dtype=torch.uint8, device=self.device)
        self.workspace_bytes = nbytes
        pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        ns = (C.c_float * len(strengths))(*strengths)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.la_synth_create(self.img_resolution, self.img_channels, self.w_dim, chan, self.conv_clamp, pp,
                                           len(params), ns, len(strengths), self._fir.ctypes.data, 4, 4, self.max_batch,
                                           _lib.ptr(self._workspace), nbytes, _lib.stream_ptr(), C.byref(h)),
                       'la_synth_create')
        self._h = h
        self._lib = lib
        self.set_precision(precision)

    # fp16 operand scales (f16x2 mode): since round 4 every forward contraction's scale is derived from the DATA of the pass by the kernel
    # that produces its input (la_style.hip: la_xscale_bound_kernel; la_common.h: slot rows) -- no a-priori bound, no calibration, nothing
    # to select.  `operand_scale` is kept as a read-only description for log lines.
    operand_scale = 'data (producer-lowered slot rows)'

    def operand_headroom(self, ws):
        """Diagnostic: smallest ratio, over the 3x3 layers after the first and the samples of `ws`, between the data maximum of a
        forward fp16 operand, max|x*s|, and conv_clamp * max|s| -- how far a generator's activations sit below their clamp bound (rounds
        2-3 derived the forward scales from that bound and needed this ratio >= 2^-11; the product no longer depends on it).  One
        forward pass of `ws`; read from the stored layer outputs and styles."""
        ws = ws.to(self.device, torch.float32).contiguous()
        self.forward(ws, noise_mode='const')
        b = ws.shape[0]
        s = self.styles(b).abs()
        cins = [self.channels[0]] + [self.channels[self.block_resolutions.index(r)] for r in self.layer_resolutions[:-1]]
        ratio, off = float('inf'), cins[0]
        for k in range(1, len(self.layer_resolutions)):
            sk = s[:, off:off + cins[k]]
            xm = self.layer_output(k - 1, b).abs().amax(dim=(2, 3))
            r = (xm * sk).amax(dim=1) / (self.conv_clamp * sk.amax(dim=1)).clamp_min(1e-30)
            ratio = min(ratio, float(r.min()))
            off += cins[k]
        return ratio

    def set_precision(self, precision):
        """'f32' exact fp32 MFMA | 'f16x2' scaled split-fp16, 3 MFMAs (fp32-class error) | 'bf16x3' split-bf16, 6 MFMAs
        (fp32-class error) | 'bf16x2' split-bf16, 3 MFMAs (approximate)."""
        _lib.check(self._lib.la_synth_set_precision(self._h, PRECISIONS[precision]), 'la_synth_set_precision')
        self.precision = precision

    @classmethod
    def from_generator(cls, G, device, max_batch, conv_clamp=None, precision='f32'):
        if conv_clamp is None and not isinstance(G, dict):
            try:
                conv_clamp = getattr(G.synthesis, f'b{G.img_resolution}').conv1.conv_clamp
            except AttributeError:
                conv_clamp = 256.0
        return cls(_state_dict_of(G), device, max_batch, conv_clamp=256.0 if conv_clamp is None else conv_clamp,
                   precision=precision)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.la_synth_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def make_noises(self, batch, generator=None, batch_seed=None, rows=None):
        """Unit-variance noise tensors for noise_mode='random': one [B,res,res] per SynthesisLayer, None where the layer's
        noise_strength is 0 (the term vanishes, nothing is drawn).  Without `batch_seed`: torch.randn on the device generator (or
        `generator`).  With `batch_seed` the draw is a pure function of (seed, layer, GLOBAL sample row, element) -- the counter-based
        generator of the library (la_noise_normal_f32: Philox4x32-10 + Box-Muller) -- and `rows = (lo, hi)` generates samples
        lo..hi-1 of a batch of `batch` ONLY: a rank holding a shard draws its rows and nothing else (one launch per layer, no
        transient of the global batch's size; rounds 3-4 drew the whole batch with torch.randn on every rank and kept a slice),
        and the gathered batch does not depend on how it was sharded."""
        lo, hi = (0, batch) if rows is None else rows
        assert 0 <= lo <= hi <= batch
        out = []
        lib = _lib.load()
        for li, (r, ns) in enumerate(zip(self.layer_resolutions, self.noise_strengths)):
            if ns == 0.0:
                out.append(None)
            elif batch_seed is None:
                assert rows is None
                out.append(torch.randn([batch, r, r], device=self.device, generator=generator))
            else:
                t = torch.empty([hi - lo, r, r], device=self.device, dtype=torch.float32)
                with torch.cuda.device(self.device):
                    _lib.check(lib.la_noise_normal_f32(_lib.ptr(t), hi - lo, r * r, int(batch_seed) & 0xFFFFFFFFFFFFFFFF, li, lo,
                                                       _lib.stream_ptr()), 'la_noise_normal_f32')
                out.append(t)
        return out

    def noise_pointer_array(self, noises):
        if noises is None:
            return None
        assert len(noises) == self.num_layers
        for t, r, ns in zip(noises, self.layer_resolutions, self.noise_strengths):
            if t is None:
                assert ns == 0.0, 'a layer with noise_strength != 0 needs its noise tensor'
                continue
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape[-2:] == (r, r)
        return (C.c_void_p * len(noises))(*[t.data_ptr() if t is not None else None for t in noises])

    def forward(self, ws, noise_mode='const', noises=None, out=None):
        """ws [B,num_ws,w_dim] (or [B,1,w_dim] / [B,w_dim]: W space) -> img [B,C,R,R]."""
        _lib.require_gpu(ws)
        ws = ws.contiguous().float()
        if ws.ndim == 2:
            ws = ws[:, None]
        B = ws.shape[0]
        assert ws.shape[2] == self.w_dim and ws.shape[1] in (1, self.num_ws)
        lstride = 0 if ws.shape[1] == 1 else self.w_dim
        mode = NOISE_MODES[noise_mode]
        if mode == 2 and noises is None:
            noises = self.make_noises(B)
        np_arr = self.noise_pointer_array(noises) if mode == 2 else None
        img = out if out is not None else torch.empty([B, self.img_channels, self.img_resolution, self.img_resolution],
                                                      device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.la_synth_forward(self._h, _lib.ptr(ws), ws.shape[1] * self.w_dim, lstride, B, mode, np_arr,
                                                  _lib.ptr(img), _lib.stream_ptr()), 'la_synth_forward')
        self._last_noises = noises   # keep alive until the matching backward
        return img

    def backward(self, g_img):
        """d(loss)/d(img) [B,C,R,R] -> d(loss)/d(ws) [B,num_ws,w_dim] for the last forward."""
        g_img = g_img.contiguous().float()
        B = g_img.shape[0]
        dws = torch.empty([B, self.num_ws, self.w_dim], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.la_synth_backward(self._h, _lib.ptr(g_img), _lib.ptr(dws), _lib.stream_ptr()),
                       'la_synth_backward')
        return dws

    # debugging / test taps -------------------------------------------------------------------
    def _view(self, p, shape):
        n = int(np.prod(shape))
        base = self._workspace.data_ptr()
        off = p - base
        assert 0 <= off and off + 4 * n <= self.workspace_bytes
        return self._workspace[off:off + 4 * n].view(torch.float32).reshape(shape)

    def layer_output(self, k, batch):
        r = self.layer_resolutions[k]
        block = self.block_resolutions.index(r)
        return self._view(self._lib.la_synth_layer_output(self._h, k), [batch, self.channels[block], r, r])

    def styles(self, batch):
        rows = self._lib.la_synth_style_rows(self._h)
        return self._view(self._lib.la_synth_styles(self._h), [batch, rows])

    def style_grads(self, batch):
        rows = self._lib.la_synth_style_rows(self._h)
        return self._view(self._lib.la_synth_style_grads(self._h), [batch, rows])


class _SynthesisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ws, engine, noise_mode, noises):
        ctx.engine = engine
        ctx.ws_shape = ws.shape
        return engine.forward(ws, noise_mode=noise_mode, noises=noises)

    @staticmethod
    def backward(ctx, g_img):
        dws = ctx.engine.backward(g_img)
        shape = ctx.ws_shape
        if len(shape) == 2:
            dws = dws.sum(dim=1)
        elif shape[1] == 1:
            dws = dws.sum(dim=1, keepdim=True)
        return dws, None, None, None


def synthesis(engine, ws, noise_mode='const', noises=None):
    """Differentiable (w.r.t. ws) call of the HIP synthesis engine, autograd-compatible."""
    return _SynthesisFn.apply(ws, engine, noise_mode, noises)


class MappingEngine:
    """G.mapping(z, c=None, truncation_psi=psi) on the HIP path (reference call sites util_latent_aug.py:203,460).

    Built from the `mapping.*` entries of the generator state_dict (legacy.py:172-176)."""

    def __init__(self, G, device, lr_multiplier=0.01):
        self._lib = _lib.load()
        self.device = torch.device(device)
        sd = G if isinstance(G, dict) else G.state_dict()
        sd = {k[len('mapping.'):]: v for k, v in sd.items() if k.startswith('mapping.')}
        n = 0
        while f'fc{n}.weight' in sd:
            n += 1
        if n == 0:
            raise _lib.LatentAugHipError('generator has no mapping.fc* tensors: z input / rand_aug needs the mapping network')
        self.num_layers = n
        self.weights = [sd[f'fc{i}.weight'].detach().to(self.device, torch.float32).contiguous() for i in range(n)]
        self.biases = [sd[f'fc{i}.bias'].detach().to(self.device, torch.float32).contiguous() for i in range(n)]
        self.z_dim = int(self.weights[0].shape[1])
        self.w_dim = int(self.weights[-1].shape[0])
        self.w_avg = sd['w_avg'].detach().to(self.device, torch.float32).contiguous() if 'w_avg' in sd else None
        self.lr_multiplier = float(lr_multiplier)
        self._wp = (C.c_void_p * n)(*[t.data_ptr() for t in self.weights])
        self._bp = (C.c_void_p * n)(*[t.data_ptr() for t in self.biases])

    def forward(self, z, num_ws, truncation_psi=1.0):
        _lib.require_gpu(z)
        z = z.contiguous().float()
        B = z.shape[0]
        assert z.shape[1] == self.z_dim
        tmp = torch.empty([2 * B * max(self.z_dim, self.w_dim)], device=self.device, dtype=torch.float32)
        ws = torch.empty([B, num_ws, self.w_dim], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.la_mapping_forward_f32(_lib.ptr(z), B, self.z_dim, self.w_dim, self.num_layers, self._wp,
                                                        self._bp, self.lr_multiplier, _lib.ptr(self.w_avg),
                                                        float(truncation_psi), num_ws, _lib.ptr(tmp), _lib.ptr(ws),
                                                        _lib.stream_ptr()), 'la_mapping_forward')
        return ws


class DiscriminatorEngine:
    """D(x, c=None) and d(loss_disc)/dx on the HIP path (reference calc_loss_disc, util_latent_aug.py:363-371).

    Built from the discriminator's state_dict (names of legacy.py:271-288: b{res}.{fromrgb,conv0,conv1,skip}.*,
    b4.{conv,fc,out}.*); architecture 'resnet', MinibatchStd group 4."""

    def __init__(self, D, device, max_batch, conv_clamp=256.0, precision='f32', mbstd_group_size=4):
        lib = _lib.load()
        self._lib = lib
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.LatentAugHipError('DiscriminatorEngine needs a ROCm device (no CPU fallback)')
        sd = D if isinstance(D, dict) else D.state_dict()
        res_list = sorted({int(k.split('.')[0][1:]) for k in sd if k.startswith('b') and k.split('.')[0][1:].isdigit()})
        assert res_list[0] == 4 and len(res_list) >= 2, 'expected discriminator blocks b4..bR'
        R = res_list[-1]
        self.img_resolution = R
        self.img_channels = int(sd[f'b{R}.fromrgb.weight'].shape[1])
        # channel table at resolution 4 << k: C[res] = in-channels of b{res}.conv0 ; C[4] = out-channels of b8.conv1
        ch = {r: int(sd[f'b{r}.conv0.weight'].shape[1]) for r in res_list if r > 4}
        ch[4] = int(sd['b4.conv.weight'].shape[0])
        self.channels = [ch[r] for r in res_list]
        self._keep = []

        def dev(name):
            t = sd[name].detach().to(device=self.device, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t

        params = []
        for r in reversed(res_list[1:]):
            if r == R:
                params += [dev(f'b{r}.fromrgb.weight'), dev(f'b{r}.fromrgb.bias')]
            params += [dev(f'b{r}.conv0.weight'), dev(f'b{r}.conv0.bias'), dev(f'b{r}.conv1.weight'), dev(f'b{r}.conv1.bias'),
                       dev(f'b{r}.skip.weight')]
        params += [dev('b4.conv.weight'), dev('b4.conv.bias'), dev('b4.fc.weight'), dev('b4.fc.bias'), dev('b4.out.weight'),
                   dev('b4.out.bias')]
        assert len(params) == lib.la_disc_num_params(R)
        f1 = np.array([1, 3, 3, 1], dtype=np.float32)
        self._fir = np.ascontiguousarray(np.outer(f1, f1) / 64.0, dtype=np.float32)
        chan = (C.c_int * len(self.channels))(*self.channels)
        self.max_batch = int(max_batch)
        nbytes = lib.la_disc_workspace_bytes(R, self.img_channels, chan, self.max_batch)
        assert nbytes > 0
        self._workspace = torch.empty([nbytes], dtype=torch.uint8, device=self.device)
        pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        h = C.c_void_p()
        clamp = float(-1.0 if conv_clamp is None else conv_clamp)
        with torch.cuda.device(self.device):
            _lib.check(lib.la_disc_create(R, self.img_channels, chan, clamp, pp, len(params), self._fir.ctypes.data,
                                          int(mbstd_group_size), self.max_batch, _lib.ptr(self._workspace), nbytes,
                                          _lib.stream_ptr(), C.byref(h)), 'la_disc_create')
        self._h = h
        _lib.check(lib.la_disc_set_precision(h, PRECISIONS[precision]), 'la_disc_set_precision')

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.la_disc_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def forward(self, img):
        """img [B,C,R,R] -> logits [B,1]."""
        _lib.require_gpu(img)
        img = img.contiguous().float()
        B = img.shape[0]
        with torch.cuda.device(self.device):
            _lib.check(self._lib.la_disc_forward(self._h, _lib.ptr(img), B, _lib.stream_ptr()), 'la_disc_forward')
            p = self._lib.la_disc_logits(self._h)
            off = p - self._workspace.data_ptr()
            return self._workspace[off:off + 4 * B].view(torch.float32).clone().reshape(B, 1)

    def backward(self, dlogits):
        """d(loss)/d(logits) [B,1] -> d(loss)/d(img) for the last forward."""
        dl = dlogits.contiguous().float().reshape(-1)
        B = dl.shape[0]
        g = torch.empty([B, self.img_channels, self.img_resolution, self.img_resolution], device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.la_disc_backward(self._h, _lib.ptr(dl), _lib.ptr(g), 0, _lib.stream_ptr()), 'la_disc_backward')
        return g


FEAT_CONV, FEAT_TAP, FEAT_MAXPOOL, FEAT_AVGPOOL = 0, 1, 2, 3


class FeatureEngine:
    """LPIPS-style feature net on the HIP path: `vgg16(x, resize_images=False, return_lpips=True)` of
    util_latent_aug.py:395 and its backward.  `ops` is a list of ('conv', weight, bias) | ('tap', lin) | ('maxpool',) |
    ('avgpool',) in execution order; see `vgg16_lpips_ops` for the VGG16 layout."""

    def __init__(self, ops, device, in_res, max_batch, in_ch=3, precision='f32'):
        lib = _lib.load()
        self._lib = lib
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.LatentAugHipError('FeatureEngine needs a ROCm device (no CPU fallback)')
        self._keep = []
        desc, params = [], []
        c = in_ch
        for op in ops:
            if op[0] == 'conv':
                w = op[1].detach().to(self.device, torch.float32).contiguous()
                b = op[2].detach().to(self.device, torch.float32).contiguous()
                assert w.shape[1] == c and w.shape[2:] == (3, 3)
                desc.append(_lib.FeatOp(FEAT_CONV, c, w.shape[0]))
                c = w.shape[0]
                params += [w, b]
            elif op[0] == 'tap':
                lin = op[1].detach().to(self.device, torch.float32).contiguous()
                assert lin.shape == (c,)
                desc.append(_lib.FeatOp(FEAT_TAP, c, c))
                params.append(lin)
            elif op[0] in ('maxpool', 'avgpool'):
                desc.append(_lib.FeatOp(FEAT_MAXPOOL if op[0] == 'maxpool' else FEAT_AVGPOOL, c, c))
            else:
                raise ValueError(op[0])
        self._keep = params
        # short content hash of the weights (host side, once): identifies which feature net a cached bank was built with
        import hashlib
        hsh = hashlib.sha1()
        for op in ops:
            hsh.update(op[0].encode())
            for t in op[1:]:
                hsh.update(t.detach().to('cpu', torch.float32).contiguous().numpy().tobytes())
        self.weights_digest = hsh.hexdigest()[:10]
        self.in_ch, self.in_res, self.max_batch = in_ch, in_res, int(max_batch)
        arr = (_lib.FeatOp * len(desc))(*desc)
        nbytes = lib.la_feat_workspace_bytes(len(desc), arr, in_ch, in_res, self.max_batch)
        assert nbytes > 0, 'invalid feature-net description'
        self._workspace = torch.empty([nbytes], dtype=torch.uint8, device=self.device)
        pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.la_feat_create(len(desc), arr, pp, len(params), in_ch, in_res, self.max_batch,
                                          _lib.ptr(self._workspace), nbytes, _lib.stream_ptr(), C.byref(h)), 'la_feat_create')
        self._h = h
        self.num_features = lib.la_feat_num_features(h)
        _lib.check(lib.la_feat_set_precision(h, PRECISIONS[precision]), 'la_feat_set_precision')

    @classmethod
    def from_torchscript(cls, src, device, in_res, max_batch, precision='f32', probe_seed=0, rtol=2e-3):
        """Engine for a local TorchScript `vgg16.pt` (reference: util_latent_aug.py:35-43, call :394-395).  Every candidate mapping of
        `vgg16_from_torchscript` is checked ONCE, at load time, against the module's own
        `module(x, resize_images=False, return_lpips=True)` on a small probe batch (the scripted module runs on the host for this
        check only; the criterion itself never calls it); the first one that agrees is kept together with its input affine
        (`engine.pre_scale`, `engine.pre_shift`), otherwise loading fails loudly."""
        cands = vgg16_from_torchscript(src)
        g = torch.Generator().manual_seed(probe_seed)
        probe = torch.rand([2, 1, in_res, in_res], generator=g).repeat(1, 3, 1, 1) * 255.0      # the 0..255 range the script is written for
        with torch.no_grad():
            try:
                want = cands[0].source(probe, resize_images=False, return_lpips=True)
            except (RuntimeError, TypeError):
                want = cands[0].source(probe)
        want = want.reshape(2, -1).to(torch.float32)
        errs = []
        for c in cands:
            eng = cls(c.ops, device, in_res, max_batch=max(max_batch, 2), precision='f32')
            if eng.num_features != want.shape[1]:
                errs.append(float('inf'))
                continue
            x = probe * torch.tensor(c.pre_scale).reshape(1, 3, 1, 1) + torch.tensor(c.pre_shift).reshape(1, 3, 1, 1)      # plumbing
            got = eng.forward(x.to(eng.device)).cpu()
            err = float((got - want).norm() / want.norm().clamp_min(1e-30))
            errs.append(err)
            if err <= rtol:
                if precision != 'f32' or max_batch != eng.max_batch:
                    eng = cls(c.ops, device, in_res, max_batch=max_batch, precision=precision)
                eng.pre_scale, eng.pre_shift, eng.lin_is_sqrt = c.pre_scale, c.pre_shift, c.lin_is_sqrt
                return eng
            del eng
        raise _lib.LatentAugHipError('TorchScript feature net: no mapping of its tensors onto the VGG16-LPIPS op list reproduces the '
                                     f"module's own output (relative errors of the candidates: {errs}); refusing to guess")

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.la_feat_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def forward(self, x):
        _lib.require_gpu(x)
        x = x.contiguous().float()
        N = x.shape[0]
        assert x.shape[1:] == (self.in_ch, self.in_res, self.in_res)
        f = torch.empty([N, self.num_features], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.la_feat_forward(self._h, _lib.ptr(x), N, _lib.ptr(f), _lib.stream_ptr()), 'la_feat_forward')
        self._x = x
        return f

    def backward(self, gfeat):
        gfeat = gfeat.contiguous().float()
        gx = torch.empty_like(self._x)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.la_feat_backward(self._h, _lib.ptr(gfeat), _lib.ptr(gx), _lib.stream_ptr()), 'la_feat_backward')
        return gx


def vgg16_lpips_ops(state_dict, lins):
    """Op list of the LPIPS VGG16 from torchvision-style names (`features.{0,2,5,7,10,12,14,17,19,21,24,26,28}.weight/bias`,
    cf. augments/criteria/lpips/networks.py:87-97) and the five per-channel lin weights; taps after relu1_2, 2_2, 3_3, 4_3, 5_3."""
    conv_ids = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
    tap_after = {2: 0, 7: 1, 14: 2, 21: 3, 28: 4}
    pool_after = {2, 7, 14, 21}
    ops = []
    for i in conv_ids:
        ops.append(('conv', state_dict[f'features.{i}.weight'], state_dict[f'features.{i}.bias']))
        if i in tap_after:
            ops.append(('tap', lins[tap_after[i]]))
        if i in pool_after:
            ops.append(('maxpool',))
    return ops


# ------------------------------------------------------------------------------------------------------------
# A local copy of the reference's default perceptual net: NVIDIA's TorchScript `vgg16.pt` (util_latent_aug.py:35-43 loads it
# with torch.jit.load from a URL; :394-395 calls it as vgg16(x, resize_images=False, return_lpips=True)).
class ScriptedFeatureNet:
    """What `vgg16_from_torchscript` extracts: the FeatureEngine op list plus the module's own input layer as a per-channel
    affine (`pre_scale`, `pre_shift`: x_k * scale_k + shift_k on the three repeated channels)."""

    def __init__(self, ops, pre_scale, pre_shift, lin_is_sqrt, source):
        self.ops, self.pre_scale, self.pre_shift, self.lin_is_sqrt, self.source = ops, pre_scale, pre_shift, lin_is_sqrt, source


def _script_tensors(module):
    sd = module.state_dict()
    return [(k, v.detach().to('cpu', torch.float32)) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()]


def vgg16_from_torchscript(src, map_location='cpu'):
    """Parameter tensors of a TorchScript VGG16-LPIPS module -> candidate `ScriptedFeatureNet`s.

    `src`: path / file object for torch.jit.load, or an already loaded ScriptModule.  The archive holds no Python source we
    could read names from, so the layout is recognised from the tensors themselves, in state_dict order:
      * the 3x3 convolutions: every 4-D [Co, Ci, 3, 3] tensor (13 of them, Ci of the first = 3) with the [Co] vector that follows
        it (or `<name>.bias`);
      * the five LPIPS channel weights: tensors with exactly C = 64, 128, 256, 512, 512 x (width / 64) elements that are not biases;
      * the input layer: 3-element tensors named *mean* / *shift* and *std* / *scale*  ->  (x - mean_k) / std_k.
    Two things cannot be read from the tensors: whether the stored channel weights are the lin weights or their square roots, and
    whether an input layer without buffers is hard-coded in the script.  The function therefore returns the candidates (at most
    four) and `FeatureEngine.from_torchscript` keeps the one that reproduces the module's OWN output on a probe batch; if none
    does, loading fails -- a mapping is never trusted unverified."""
    module = src if isinstance(src, torch.jit.ScriptModule) else torch.jit.load(src, map_location=map_location)
    module = module.eval()
    items = _script_tensors(module)
    convs, used = [], set()
    for idx, (k, v) in enumerate(items):
        if v.ndim == 4 and v.shape[2:] == (3, 3):
            bias = None
            cand = k[:-len('weight')] + 'bias' if k.endswith('weight') else None
            for j, (kj, vj) in enumerate(items):
                if j in used or vj.ndim != 1 or vj.shape[0] != v.shape[0]:
                    continue
                if (cand is not None and kj == cand) or (cand is None and j == idx + 1):
                    bias, _ = vj, used.add(j)
                    break
            if bias is None and idx + 1 < len(items) and items[idx + 1][1].ndim == 1 and items[idx + 1][1].shape[0] == v.shape[0]:
                bias = items[idx + 1][1]
                used.add(idx + 1)
            if bias is None:
                bias = torch.zeros([v.shape[0]])
            used.add(idx)
            convs.append((v, bias))
    if len(convs) != 13 or convs[0][0].shape[1] != 3:
        raise _lib.LatentAugHipError(f'TorchScript feature net: expected the 13 3x3 convolutions of VGG16 (first one on 3 channels), found '
                                     f'{len(convs)}: not a vgg16.pt layout')
    tap_convs = [1, 3, 6, 9, 12]                      # relu1_2, 2_2, 3_3, 4_3, 5_3
    want = [convs[i][0].shape[0] for i in tap_convs]
    lins, k = [], 0
    for idx, (name, v) in enumerate(items):
        if idx in used or k >= 5 or name.endswith('bias'):
            continue
        if v.numel() == want[k] and max(v.shape) == want[k] and float(v.min()) >= 0.0:
            lins.append(v.reshape(-1))
            used.add(idx)
            k += 1
    if len(lins) != 5:
        raise _lib.LatentAugHipError(f'TorchScript feature net: found {len(lins)} of the 5 LPIPS channel-weight tensors {want}')
    mean = std = None
    for idx, (name, v) in enumerate(items):
        if idx in used or v.numel() != 3:
            continue
        low = name.lower()
        if mean is None and ('mean' in low or 'shift' in low):
            mean = v.reshape(3)
        elif std is None and ('std' in low or 'scale' in low):
            std = v.reshape(3)
    pres = [((1.0, 1.0, 1.0), (0.0, 0.0, 0.0))]
    if mean is not None or std is not None:
        m = mean if mean is not None else torch.zeros(3)
        sd_ = std if std is not None else torch.ones(3)
        pres.insert(0, (tuple(float(1.0 / t) for t in sd_), tuple(float(-a / t) for a, t in zip(m, sd_))))
    out = []
    for pre_scale, pre_shift in pres:
        for lin_is_sqrt in (False, True):
            ops = []
            for i, (w, b) in enumerate(convs):
                ops.append(('conv', w, b))
                if i in tap_convs:
                    ln = lins[tap_convs.index(i)]
                    ops.append(('tap', ln.square() if lin_is_sqrt else ln))
                    if i != tap_convs[-1]:
                        ops.append(('maxpool',))
            out.append(ScriptedFeatureNet(ops, pre_scale, pre_shift, lin_is_sqrt, module))
    return out
