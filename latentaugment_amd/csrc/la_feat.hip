// LPIPS-style perceptual feature extractor, forward + backward-to-input, for the criterion
//   calc_loss_lpips_torchscript (augments/utils/util_latent_aug.py:387-409):
//       x = crop[:, mode].repeat(1, 3, 1, 1);  f = vgg16(x, resize_images=False, return_lpips=True);
//       loss_mode = l2_loss_vectorized(f, bank_mode, compute_mean=False).sum() / (n * m) * w_lpips
// The network is a caller-described sequence of { conv3x3 + bias + ReLU | 2x2 max-pool | 2x2 avg-pool | LPIPS tap } ops
// (VGG16: 13 convs, 4 max-pools, taps after conv1_2 / 2_2 / 3_3 / 4_3 / 5_3).  A tap contributes
//       f[n][c][p] * rsqrt(sum_c f^2 + 1e-10) * sqrt(lin[c]) / sqrt(H*W)
// to the output vector, so that squared L2 distance of two vectors is their LPIPS distance.
// Contractions reuse la_conv*.hip (shared weights); pools / taps are small HBM-bound kernels.
#include "la_feat.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "la_conv.h"
#include "la_modconv.h"
#include "la_style.h"

extern "C" int la_bias_act_grad_f32(const float* dy, const float* yref, float* dx, float* db, long n, long stepb, int nb, int act,
                                    float alpha, float gain, float clamp, hipStream_t stream);

#define FEAT_MAX_OPS 48

struct FOp {
    int kind, cin, cout, res_in, res_out;
    const float *w, *bias, *lin;
    float *wf, *wb; void *wqf, *wqb; int mb_;
    float* y;          // output activation [maxN][cout][res_out^2] (conv, pools); taps have none
    long feat_off;     // taps: offset of the slice inside the feature vector
};

struct la_feat {
    int nops, in_ch, in_res, maxN, F, precision;
    FOp op[FEAT_MAX_OPS];
    float *gA, *gB;
    float* pm;           // plane maxima of the gradient entering a backward contraction (la_conv_act_grad_pmax: exact-fp32 / fallback path)
    // fp16 operand scales as slot rows lowered by the producing kernels (la_common.h): xs_f [nops][maxN][LA_XS_FAN] for the input of every
    // forward conv but the first, xs_b for the (masked) gradient entering every backward conv; seam_scr: scratch for the demod-gradient
    // partials the fused activation backward of the contraction epilogues writes (unused here: these convs have no demodulation)
    float *xs_f, *xs_b, *seam_scr;
    void* cws; size_t cws_bytes;
    const float* x_in;   // input of the last forward
    int lastN;
};

static size_t falign(size_t v) { return (v + 63) & ~(size_t)63; }
struct FCarver {
    char* base; size_t off;
    float* take(size_t nfloats) { float* p = base ? (float*)(base + off) : nullptr; off += falign(nfloats * 4); return p; }
};

static int f_describe(la_feat* h, int nops, const la_feat_op* ops, int in_ch, int in_res, int maxN) {
    LA_CHECK_ARG(nops >= 1 && nops <= FEAT_MAX_OPS && ops, "feat: bad op list");
    LA_CHECK_ARG(in_ch >= 1 && in_res >= 2 && maxN >= 1, "feat: bad input shape");
    memset(h, 0, sizeof(*h));
    h->nops = nops; h->in_ch = in_ch; h->in_res = in_res; h->maxN = maxN;
    int c = in_ch, r = in_res;
    long F = 0;
    for (int k = 0; k < nops; ++k) {
        FOp& o = h->op[k];
        o.kind = ops[k].kind; o.cin = c; o.res_in = r;
        switch (o.kind) {
            case LA_FEAT_CONV_RELU:
                LA_CHECK_ARG(ops[k].cin == c && ops[k].cout >= 4 && ops[k].cout % 4 == 0, "feat: conv channels mismatch / not a multiple of 4");
                o.cout = ops[k].cout; o.res_out = r; c = o.cout; break;
            case LA_FEAT_MAXPOOL2: case LA_FEAT_AVGPOOL2:
                LA_CHECK_ARG(r % 2 == 0, "feat: pooling needs an even resolution");
                o.cout = c; o.res_out = r / 2; r /= 2; break;
            case LA_FEAT_TAP:
                o.cout = c; o.res_out = r; o.feat_off = F; F += (long)c * r * r; break;
            default:
                la_set_error("feat: unknown op kind"); return LA_ERR_ARG;
        }
    }
    LA_CHECK_ARG(F > 0 && F < (1L << 31), "feat: no tap op / feature vector too long");
    h->F = (int)F;
    return LA_OK;
}

static size_t f_layout(la_feat* h, void* ws) {
    FCarver c{(char*)ws, 0};
    const size_t mn = h->maxN;
    size_t gmax = mn * h->in_ch * (size_t)h->in_res * h->in_res, cw = 0, pmax = 0;
    for (int k = 0; k < h->nops; ++k) {
        FOp& o = h->op[k];
        const size_t n_out = mn * o.cout * (size_t)o.res_out * o.res_out;
        if (o.kind == LA_FEAT_CONV_RELU) {
            o.mb_ = (o.cin + 3) & ~3;
            o.wf = c.take((size_t)o.cin * o.cout * 9); o.wb = c.take((size_t)o.mb_ * o.cout * 9);
            o.wqf = c.take((la_conv_split_pack_bytes(o.cout, o.cin, 9) + 3) / 4);
            o.wqb = c.take((la_conv_split_pack_bytes(o.mb_, o.cout, 9) + 3) / 4);
            size_t w = la_modconv_workspace_bytes((int)mn, o.mb_ > o.cout ? o.mb_ : o.cout, o.mb_ > o.cout ? o.mb_ : o.cout, o.res_in, 0);
            if (w > cw) cw = w;
            const size_t gin = mn * o.mb_ * (size_t)o.res_in * o.res_in;
            if (gin > gmax) gmax = gin;
            const size_t pmn = mn * o.cout * (size_t)la_conv_act_grad_segments((long)o.res_out * o.res_out);
            if (pmn > pmax) pmax = pmn;
        }
        if (o.kind != LA_FEAT_TAP) { o.y = c.take(n_out); if (n_out > gmax) gmax = n_out; }
    }
    h->gA = c.take(gmax); h->gB = c.take(gmax);
    h->pm = c.take(pmax);
    h->xs_f = c.take((size_t)h->nops * mn * LA_XS_FAN); h->xs_b = c.take((size_t)h->nops * mn * LA_XS_FAN);
    {
        size_t sc = 16;
        for (int k = 0; k < h->nops; ++k)
            if (h->op[k].kind == LA_FEAT_CONV_RELU) {
                const size_t n = mn * (size_t)h->op[k].cout * la_conv_tiles_per_sample(h->op[k].res_in, h->op[k].res_in);
                if (n > sc) sc = n;
            }
        h->seam_scr = c.take(sc);
    }
    h->cws = c.take((cw + 3) / 4); h->cws_bytes = cw;
    return c.off;
}

extern "C" size_t la_feat_workspace_bytes(int nops, const la_feat_op* ops, int in_ch, int in_res, int max_batch) {
    la_feat* h = (la_feat*)malloc(sizeof(la_feat));
    if (!h) return 0;
    size_t need = 0;
    if (f_describe(h, nops, ops, in_ch, in_res, max_batch) == LA_OK) need = f_layout(h, nullptr);
    free(h);
    return need;
}

// params: for each op in order: conv -> weight [cout][cin][3][3], bias [cout]; tap -> lin [C]; pools -> nothing
extern "C" int la_feat_create(int nops, const la_feat_op* ops, const float* const* params, int nparams, int in_ch, int in_res,
                              int max_batch, void* workspace, size_t workspace_bytes, hipStream_t stream, la_feat** out) {
    LA_CHECK_ARG(params && workspace && out, "feat_create: null pointer");
    la_feat* h = (la_feat*)malloc(sizeof(la_feat));
    LA_CHECK_ARG(h, "feat_create: out of host memory");
    int rc = f_describe(h, nops, ops, in_ch, in_res, max_batch);
    if (rc) { free(h); return rc; }
    if (f_layout(h, workspace) > workspace_bytes) { free(h); la_set_error("feat_create: workspace too small"); return LA_ERR_WORKSPACE; }
    int p = 0;
    for (int k = 0; k < nops && !rc; ++k) {
        FOp& o = h->op[k];
        if (o.kind == LA_FEAT_CONV_RELU) {
            if (p + 2 > nparams || !params[p] || !params[p + 1]) { rc = LA_ERR_ARG; la_set_error("feat_create: missing conv tensors"); break; }
            o.w = params[p++]; o.bias = params[p++];
            rc = la_pack_conv_weights(o.w, o.wf, o.wb, nullptr, o.cout, o.cin, 9, stream, 1.f, o.mb_);
            if (!rc) rc = la_pack_conv_weights_bf16(o.w, o.wqf, o.cout, o.cin, 9, 0, 3, stream, 1.f);
            if (!rc) rc = la_pack_conv_weights_bf16(o.w, o.wqb, o.cout, o.cin, 9, 1, 3, stream, 1.f, o.mb_);
        } else if (o.kind == LA_FEAT_TAP) {
            if (p + 1 > nparams || !params[p]) { rc = LA_ERR_ARG; la_set_error("feat_create: missing tap weights"); break; }
            o.lin = params[p++];
        }
    }
    if (!rc && p != nparams) { rc = LA_ERR_ARG; la_set_error("feat_create: parameter list length mismatch"); }
    if (rc) { free(h); return rc; }
    *out = h;
    return LA_OK;
}

extern "C" void la_feat_destroy(la_feat* h) { free(h); }
extern "C" int la_feat_num_features(const la_feat* h) { return h ? h->F : 0; }
extern "C" int la_feat_set_precision(la_feat* h, int precision) {
    LA_CHECK_ARG(h && precision >= 0 && precision <= 3, "feat_set_precision: precision must be 0..3");
    h->precision = precision;
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
__global__ void la_pool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int R, long planes, int is_max) {
    const int Ro = R / 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ro * Ro) return;
    const int ox = (int)(i % Ro), oy = (int)((i / Ro) % Ro);
    const long pl = i / ((long)Ro * Ro);
    const float* p = x + (pl * R + 2 * oy) * R + 2 * ox;
    const float a = p[0], b = p[1], c = p[R], d = p[R + 1];
    y[i] = is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : 0.25f * ((a + b) + (c + d));
}

// max: the gradient goes to the first maximum in scan order (torch semantics); avg: a quarter to each
__global__ void la_pool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, int R,
                                    long planes, int is_max) {
    const int Ro = R / 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ro * Ro) return;
    const int ox = (int)(i % Ro), oy = (int)((i / Ro) % Ro);
    const long pl = i / ((long)Ro * Ro);
    const long base = (pl * R + 2 * oy) * R + 2 * ox;
    const float g = gy[i];
    if (!is_max) {
        const float q = 0.25f * g;
        gx[base] = q; gx[base + 1] = q; gx[base + R] = q; gx[base + R + 1] = q;
        return;
    }
    const float v[4] = {x[base], x[base + 1], x[base + R], x[base + R + 1]};
    int am = 0;
    for (int k = 1; k < 4; ++k) if (v[k] > v[am]) am = k;
    gx[base] = am == 0 ? g : 0.f; gx[base + 1] = am == 1 ? g : 0.f; gx[base + R] = am == 2 ? g : 0.f; gx[base + R + 1] = am == 3 ? g : 0.f;
}

// tap forward: feat[n][off + c*HW + p] = f * rsqrt(sum_c f^2 + 1e-10) * sqrt(lin[c]) / sqrt(HW)
// Workgroup = PL pixel lanes (consecutive pixels: coalesced) x 256 / PL channel groups of one image; every thread walks C / groups
// channels, the channel sums are combined through LDS in a fixed order.  (Round 1 ran one thread per pixel over all C channels:
// 16 threads walking 512 channels three times at the 4x4 tap of VGG16, 555 us.)
__global__ __launch_bounds__(256) void la_tap_fwd_kernel(const float* __restrict__ f, const float* __restrict__ lin, float* __restrict__ feat, int C,
                                                          int HW, long F, long off, int PL) {
    __shared__ float red[256];
    const int pl = threadIdx.x % PL, cg = threadIdx.x / PL, CG = 256 / PL;
    const int p = blockIdx.x * PL + pl;
    const long n = blockIdx.y;
    const bool ok = p < HW;
    const float* fp = f + n * C * HW + p;
    // (eight channel loads in flight per thread, summed in the channel order of the plain loop: at the 4x4 / 8x8 taps a thread walks
    //  32 - 128 channels and the loop was one exposed load latency per channel, 40 - 60 us for 16 workgroups)
    float s = 0.f;
    if (ok) {
        int c = cg;
        for (; c + 7 * CG < C; c += 8 * CG) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fp[(long)(c + k * CG) * HW];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k] * v[k];
        }
        for (; c < C; c += CG) { const float v = fp[(long)c * HW]; s += v * v; }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    float tot = 0.f;
    for (int g = 0; g < CG; ++g) tot += red[g * PL + pl];
    const float r = rsqrtf(tot + 1e-10f) * rsqrtf((float)HW);
    if (!ok) return;
    float* o = feat + n * F + off + p;
    int c = cg;
    for (; c + 7 * CG < C; c += 8 * CG) {
        float v[8], w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = fp[(long)(c + k * CG) * HW]; w[k] = lin[c + k * CG]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) o[(long)(c + k * CG) * HW] = v[k] * r * sqrtf(w[k]);
    }
    for (; c < C; c += CG) o[(long)c * HW] = fp[(long)c * HW] * r * sqrtf(lin[c]);
}

// tap backward: gf[k] (+)= r * (u_k - y_k * sum_c u_c y_c),  u_c = g_c * sqrt(lin_c)/sqrt(HW),  y_c = f_c * r   (same thread layout)
// mask != 0: the tapped activation is the ReLU output of the conv in front of the tap, and the result is the gradient entering that
// conv's backward contraction: the ReLU mask (f > 0) is applied to the SUM (incoming gradient + this tap's) here, and xs_row (slot rows
// [N][LA_XS_FAN], la_common.h) receives the fp16 operand scale of the result -- no separate activation-backward / plane-maxima sweep.
__global__ __launch_bounds__(256) void la_tap_bwd_kernel(const float* __restrict__ f, const float* __restrict__ lin, const float* __restrict__ gfeat,
                                                          float* __restrict__ gf, int C, int HW, long F, long off, int PL, int accumulate, int mask,
                                                          float* __restrict__ xs_row) {
    __shared__ float red[2][256];
    const int pl = threadIdx.x % PL, cg = threadIdx.x / PL, CG = 256 / PL;
    const int p = blockIdx.x * PL + pl;
    const long n = blockIdx.y;
    const bool ok = p < HW;
    const float* fp = f + n * C * HW + p;
    const float* gp = gfeat + n * F + off + p;
    const float a = rsqrtf((float)HW);
    float s = 0.f, d = 0.f;                       // sum f^2 and sum u_c f_c of this thread's channels
    if (ok) {
        int c = cg;
        for (; c + 7 * CG < C; c += 8 * CG) {      // (eight channels' loads in flight, summed in channel order: see la_tap_fwd_kernel)
            float v[8], g[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = fp[(long)(c + k * CG) * HW]; g[k] = gp[(long)(c + k * CG) * HW]; w[k] = lin[c + k * CG]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += v[k] * v[k]; d += g[k] * sqrtf(w[k]) * a * v[k]; }
        }
        for (; c < C; c += CG) {
            const float v = fp[(long)c * HW];
            s += v * v;
            d += gp[(long)c * HW] * sqrtf(lin[c]) * a * v;
        }
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = d;
    __syncthreads();
    float st = 0.f, dt = 0.f;
    for (int g = 0; g < CG; ++g) { st += red[0][g * PL + pl]; dt += red[1][g * PL + pl]; }
    const float r = rsqrtf(st + 1e-10f);
    const float dot = dt * r;                      // sum_c u_c y_c
    float omax = 0.f;
    if (ok) {
        float* op = gf + n * C * HW + p;
        int c = cg;
        for (; c + 7 * CG < C; c += 8 * CG) {
            float fv[8], g[8], w[8], prev[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                fv[k] = fp[(long)(c + k * CG) * HW]; g[k] = gp[(long)(c + k * CG) * HW]; w[k] = lin[c + k * CG];
                prev[k] = accumulate ? op[(long)(c + k * CG) * HW] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float u = g[k] * sqrtf(w[k]) * a, y = fv[k] * r;
                float v = r * (u - y * dot);
                if (accumulate) v += prev[k];
                if (mask && !(fv[k] > 0.f)) v = 0.f;
                op[(long)(c + k * CG) * HW] = v;
                omax = fmaxf(omax, fabsf(v));
            }
        }
        for (; c < C; c += CG) {
            const float fv = fp[(long)c * HW];
            const float u = gp[(long)c * HW] * sqrtf(lin[c]) * a, y = fv * r;
            float v = r * (u - y * dot);
            if (accumulate) v += op[(long)c * HW];
            if (mask && !(fv > 0.f)) v = 0.f;
            op[(long)c * HW] = v;
            omax = fmaxf(omax, fabsf(v));
        }
    }
    if (xs_row) {      // (uniform) this workgroup's maximum lowers a sub-slot of sample n's row
        __syncthreads();
        red[0][threadIdx.x] = omax;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[0][threadIdx.x] = fmaxf(red[0][threadIdx.x], red[0][threadIdx.x + o]);
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            float* row = xs_row + n * LA_XS_FAN + la_xs_sub();
            la_xs_lower(row, la_xs_peek(row), 1.f, red[0][0]);
        }
    }
}

// pixel lanes of a tap workgroup: the largest power of two <= min(64, HW)
static inline int tap_lanes(int HW) { int pl = 1; while (pl * 2 <= HW && pl < 64) pl *= 2; return pl; }

static void fbase(LaConvArgs& a) {
    memset(&a, 0, sizeof(a));
    a.in_sy = a.in_sx = a.out_sy = a.out_sx = 1; a.clamp = -1.f; a.gain = 1.f; a.act = LA_ACT_LINEAR;
}

// slot-row hand-over of the fp16 operand scales (f16x2 mode): xs_in = rows of this launch's input (null: absmax / plane-maxima passes),
// xs_out = rows the epilogue lowers for the contraction that consumes the output (forward: the next conv's; backward: those of the conv
// whose ReLU output `mask_y` is -- the epilogue then also applies that ReLU's mask to the outgoing gradient, LaConvArgs::seam_*)
static int f_conv(la_feat* h, const FOp& o, bool backward, const float* in, float* out, int N, hipStream_t stream, const float* in_pmax = nullptr,
                  int in_nseg = 0, const float* xs_in = nullptr, float* xs_out = nullptr, const float* mask_y = nullptr) {
    LaConvArgs a; fbase(a);
    a.wgt = backward ? o.wb : o.wf;
    a.precision = h->precision; a.wgt_bf16 = backward ? o.wqb : o.wqf;
    a.wgt_bf16_term_elems = la_conv_bf16_pack_elems(backward ? o.mb_ : o.cout, backward ? o.cout : o.cin, 9);
    a.ws = h->cws; a.ws_bytes = h->cws_bytes;
    a.in = in; a.out = out; a.B = N; a.in_pmax = in_pmax; a.in_pmax_nseg = in_nseg;
    a.C = backward ? o.cout : o.cin; a.M = backward ? o.mb_ : o.cout;
    const int res = o.res_in;
    a.in_bstride = (long)a.C * res * res;
    a.Hin = a.Win = a.Hout = a.Wout = a.Gy = a.Gx = res;
    a.ntaps = 9;
    for (int t = 0; t < 9; ++t) {
        a.tap_dy[t] = backward ? 1 - t / 3 : t / 3 - 1; a.tap_dx[t] = backward ? 1 - t % 3 : t % 3 - 1; a.tap_w[t] = t;
    }
    if (xs_in) { a.acc_scale_x = xs_in; a.acc_scale_fan = LA_XS_FAN; a.in_pmax = nullptr; }
    if (backward) {
        a.epi = LA_EPI_BWD;
        if (mask_y) {      // activation backward of the layer below, fused (its saved output is the epilogue's xin)
            a.xin = mask_y; a.xin_bstride = (long)a.M * res * res; a.tiles_per_sample = la_conv_tiles_per_sample(res, res);
            a.seam_ddn_part = h->seam_scr; a.seam_act = LA_ACT_RELU; a.seam_alpha = 0.f; a.seam_gain = 1.f; a.seam_clamp = -1.f;
            a.seam_xs_out = xs_out; a.seam_xs_mult = 1.f;
        }
    } else { a.epi = LA_EPI_FWD; a.bias = o.bias; a.act = LA_ACT_RELU; a.gain = 1.f; a.fwd_xs_out = xs_out; }
    return la_conv_launch(a, stream);
}

extern "C" int la_feat_forward(la_feat* h, const float* x, int N, float* feat_out, hipStream_t stream) {
    LA_CHECK_ARG(h && x && feat_out, "feat_forward: null pointer");
    LA_CHECK_ARG(N >= 1 && N <= h->maxN, "feat_forward: batch exceeds max_batch");
    const float* cur = x;
    int rc;
    // fp16 x2 mode: every conv's epilogue lowers the slot rows of the NEXT conv's operand scale (a pool in between only shrinks the
    // maximum; taps do not touch the activation), so only the first conv -- whose input no kernel of this engine produces -- runs the
    // absmax / scale passes
    const bool slots = h->precision == LA_PREC_F16X2 && !la_dev_env("LA_NO_FEAT_SLOTS");      // (dev knob: the round-3 passes)
    if (slots) LA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->xs_f), (int)LA_XS_INIT, (size_t)h->nops * h->maxN * LA_XS_FAN, stream));
    auto rows = [&](float* base, int op) { return base + (size_t)op * h->maxN * LA_XS_FAN; };
    bool first_conv = true;
    for (int k = 0; k < h->nops; ++k) {
        FOp& o = h->op[k];
        const int HWo = o.res_out * o.res_out;
        if (o.kind == LA_FEAT_CONV_RELU) {
            int nxt = -1;
            for (int q = k + 1; q < h->nops; ++q) if (h->op[q].kind == LA_FEAT_CONV_RELU) { nxt = q; break; }
            if ((rc = f_conv(h, o, false, cur, o.y, N, stream, nullptr, 0, (slots && !first_conv) ? rows(h->xs_f, k) : nullptr,
                             (slots && nxt >= 0) ? rows(h->xs_f, nxt) : nullptr)))
                return rc;
            first_conv = false;
            cur = o.y;
        } else if (o.kind == LA_FEAT_TAP) {
            const int pl = tap_lanes(HWo);
            hipLaunchKernelGGL(la_tap_fwd_kernel, dim3(la_cdiv(HWo, pl), N), dim3(256), 0, stream, cur, o.lin, feat_out, o.cout, HWo,
                               (long)h->F, o.feat_off, pl);
        } else {
            const long planes = (long)N * o.cout;
            hipLaunchKernelGGL(la_pool2_fwd_kernel, dim3(la_cdiv(planes * HWo, 256)), dim3(256), 0, stream, cur, o.y, o.res_in, planes,
                               o.kind == LA_FEAT_MAXPOOL2);
            cur = o.y;
        }
    }
    LA_CHECK_LAUNCH();
    h->x_in = x; h->lastN = N;
    return LA_OK;
}

// gx [N][in_ch][in_res^2] = d(sum gfeat . feat)/dx for the last forward (x must still hold the forward's input)
extern "C" int la_feat_backward(la_feat* h, const float* gfeat, float* gx, hipStream_t stream) {
    LA_CHECK_ARG(h && gfeat && gx && h->lastN >= 1, "feat_backward: null pointer / no forward pass");
    const int N = h->lastN;
    float* g = h->gA;       // gradient w.r.t. the activation that op k produced / tapped
    float* other = h->gB;
    bool have = false;
    int rc;
    // fp16 x2 mode: the ReLU mask of conv k and the operand scale of the masked gradient come from the kernel that WRITES that gradient
    // last -- the tap behind conv k (la_tap_bwd_kernel, mask fused), or the backward contraction of conv k+1 when it follows directly
    // (its epilogue applies the mask of its xin = y_k, LaConvArgs::seam_*) -- instead of an activation-backward sweep with plane maxima
    // and a scale-reduction launch per conv.  `masked`: g already carries op k's mask and its slot rows are final.
    const bool slots = h->precision == LA_PREC_F16X2 && !la_dev_env("LA_NO_FEAT_SLOTS");      // (dev knob: the round-3 passes)
    if (slots) LA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->xs_b), (int)LA_XS_INIT, (size_t)h->nops * h->maxN * LA_XS_FAN, stream));
    auto rows = [&](float* base, int op) { return base + (size_t)op * h->maxN * LA_XS_FAN; };
    bool masked = false;
    for (int k = h->nops - 1; k >= 0; --k) {
        FOp& o = h->op[k];
        // activation feeding op k
        const float* act_in = h->x_in;
        for (int q = k - 1; q >= 0; --q) if (h->op[q].kind != LA_FEAT_TAP) { act_in = h->op[q].y; break; }
        const int HWo = o.res_out * o.res_out;
        if (o.kind == LA_FEAT_TAP) {
            const int pl = tap_lanes(HWo);
            // (the tap's input is the output of the conv right in front of it: this launch is the last writer of that conv's gradient)
            const bool fuse = slots && k > 0 && h->op[k - 1].kind == LA_FEAT_CONV_RELU;
            hipLaunchKernelGGL(la_tap_bwd_kernel, dim3(la_cdiv(HWo, pl), N), dim3(256), 0, stream, act_in, o.lin, gfeat, g, o.cout, HWo,
                               (long)h->F, o.feat_off, pl, have ? 1 : 0, fuse ? 1 : 0, fuse ? rows(h->xs_b, k - 1) : (float*)nullptr);
            have = true;
            masked = fuse;
        } else if (o.kind == LA_FEAT_CONV_RELU) {
            LA_CHECK_ARG(have, "feat_backward: the op list must end with a tap");
            float* dst = (k == 0 && o.mb_ == o.cin) ? gx : other;
            // the conv below, if it feeds this one directly: this contraction's epilogue masks its gradient and lowers its slot rows
            const bool below = slots && k > 0 && h->op[k - 1].kind == LA_FEAT_CONV_RELU && o.mb_ == o.cin && dst != gx;
            if (masked) {
                if ((rc = f_conv(h, o, true, g, dst, N, stream, nullptr, 0, rows(h->xs_b, k), below ? rows(h->xs_b, k - 1) : nullptr,
                                 below ? h->op[k - 1].y : nullptr)))
                    return rc;
            } else {
                // ReLU mask of this layer on the incoming gradient, with the plane maxima the fp16 operand scale of its backward
                // contraction needs (one sweep instead of a mask pass + an absmax pass): exact-fp32 / bf16 modes, op lists without
                // a tap or a conv right behind this conv
                if ((rc = la_conv_act_grad_pmax(g, o.y, g, h->pm, N, o.cout, HWo, LA_ACT_RELU, 0.f, 1.f, -1.f, stream))) return rc;
                if ((rc = f_conv(h, o, true, g, dst, N, stream, h->pm, la_conv_act_grad_segments(HWo), nullptr, below ? rows(h->xs_b, k - 1) : nullptr,
                                 below ? h->op[k - 1].y : nullptr)))
                    return rc;
            }
            masked = below;
            if (k == 0 && dst != gx) {
                // padded backward channels (cin not a multiple of 4): copy the real ones out
                const long HWi = (long)o.res_in * o.res_in;
                // (one strided copy: a copy per sample was 16 launches of 5 us each in every step of preset E)
                LA_HIP(hipMemcpy2DAsync(gx, sizeof(float) * o.cin * HWi, dst, sizeof(float) * o.mb_ * HWi, sizeof(float) * o.cin * HWi, (size_t)N,
                                        hipMemcpyDeviceToDevice, stream));
            }
            float* t = g; g = other; other = t;
            if (o.mb_ != o.cin && k != 0) { la_set_error("feat_backward: only the first conv may have cin % 4 != 0"); return LA_ERR_ARG; }
        } else {
            LA_CHECK_ARG(have, "feat_backward: the op list must end with a tap");
            const long planes = (long)N * o.cout;
            hipLaunchKernelGGL(la_pool2_bwd_kernel, dim3(la_cdiv(planes * HWo, 256)), dim3(256), 0, stream, act_in, g, other, o.res_in,
                               planes, o.kind == LA_FEAT_MAXPOOL2);
            float* t = g; g = other; other = t;
            masked = false;
        }
    }
    LA_CHECK_LAUNCH();
    if (h->op[0].kind != LA_FEAT_CONV_RELU) {
        // the gradient ended in g (no first conv wrote gx)
        const long n_in = (long)N * h->in_ch * h->in_res * h->in_res;
        LA_HIP(hipMemcpyAsync(gx, g, sizeof(float) * n_in, hipMemcpyDeviceToDevice, stream));
    }
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// criterion glue: crop (+ repeat to 3 channels, affine preprocess) and its adjoint
//   xc[(c*B + b)][k][y][x] = img[b][c][y0+y][x0+x] * scale + shift      k = 0..rep-1
// (pos != null: the window position {y0, x0} is read from device memory, so that a captured launch follows the position the
//  host draws per forward)
struct LaPreAffine { float scale[4], shift[4]; };      // per repeated channel k (k < rep <= 4)
__global__ void la_crop_repeat_kernel(const float* __restrict__ img, float* __restrict__ xc, int B, int imgc, int R, int S, int y0,
                                      int x0, int rep, LaPreAffine pre, long total, const int* __restrict__ pos) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (pos) { y0 = pos[0]; x0 = pos[1]; }
    const int x = (int)(i % S), y = (int)((i / S) % S);
    const long rest = i / ((long)S * S);
    const long n = rest / rep;
    const int k = (int)(rest - n * rep);
    const int c = (int)(n / B), b = (int)(n - (long)c * B);
    xc[i] = img[(((long)b * imgc + c) * R + y0 + y) * R + x0 + x] * pre.scale[k] + pre.shift[k];
}

// g_img[b][c][y0+y][x0+x] += sum_k scale[k] * gxc[(c*B+b)][k][y][x]
__global__ void la_crop_repeat_bwd_kernel(const float* __restrict__ gxc, float* __restrict__ g_img, int B, int imgc, int R, int S,
                                          int y0, int x0, int rep, LaPreAffine pre, long total, const int* __restrict__ pos) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (pos) { y0 = pos[0]; x0 = pos[1]; }
    const int x = (int)(i % S), y = (int)((i / S) % S);
    const long n = i / ((long)S * S);
    const int c = (int)(n / B), b = (int)(n - (long)c * B);
    float acc = 0.f;
    for (int k = 0; k < rep; ++k) acc += gxc[((n * rep + k) * S + y) * S + x] * pre.scale[k];
    g_img[(((long)b * imgc + c) * R + y0 + y) * R + x0 + x] += acc;
}

// scale / shift: one value per repeated channel ([rep], rep <= 4) -- e.g. the (x - mean_k) / std_k of an ImageNet-style input layer
int la_crop_repeat_ex3(const float* img, float* xc, int B, int imgc, int R, int S, int y0, int x0, const int* pos_dev, int rep,
                       const float* scale, const float* shift, hipStream_t stream) {
    LA_CHECK_ARG(img && xc && y0 >= 0 && x0 >= 0 && y0 + S <= R && x0 + S <= R && rep >= 1 && rep <= 4 && scale && shift, "crop_repeat: bad arguments");
    LaPreAffine pre;
    for (int k = 0; k < 4; ++k) { pre.scale[k] = scale[k < rep ? k : rep - 1]; pre.shift[k] = shift[k < rep ? k : rep - 1]; }
    const long total = (long)B * imgc * rep * S * S;
    hipLaunchKernelGGL(la_crop_repeat_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, stream, img, xc, B, imgc, R, S, y0, x0, rep, pre,
                       total, pos_dev);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
int la_crop_repeat_ex(const float* img, float* xc, int B, int imgc, int R, int S, int y0, int x0, const int* pos_dev, int rep, float scale,
                      float shift, hipStream_t stream) {
    const float sc[4] = {scale, scale, scale, scale}, sh[4] = {shift, shift, shift, shift};
    LA_CHECK_ARG(rep >= 1 && rep <= 4, "crop_repeat: rep must be 1..4");
    return la_crop_repeat_ex3(img, xc, B, imgc, R, S, y0, x0, pos_dev, rep, sc, sh, stream);
}
extern "C" int la_crop_repeat_f32(const float* img, float* xc, int B, int imgc, int R, int S, int y0, int x0, int rep, float scale,
                                  float shift, hipStream_t stream) {
    return la_crop_repeat_ex(img, xc, B, imgc, R, S, y0, x0, nullptr, rep, scale, shift, stream);
}

int la_crop_repeat_grad_ex3(const float* gxc, float* g_img, int B, int imgc, int R, int S, int y0, int x0, const int* pos_dev, int rep,
                            const float* scale, hipStream_t stream) {
    LA_CHECK_ARG(gxc && g_img && y0 >= 0 && x0 >= 0 && y0 + S <= R && x0 + S <= R && rep >= 1 && rep <= 4 && scale, "crop_repeat_grad: bad arguments");
    LaPreAffine pre;
    for (int k = 0; k < 4; ++k) { pre.scale[k] = scale[k < rep ? k : rep - 1]; pre.shift[k] = 0.f; }
    const long total = (long)B * imgc * S * S;
    hipLaunchKernelGGL(la_crop_repeat_bwd_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, stream, gxc, g_img, B, imgc, R, S, y0, x0, rep,
                       pre, total, pos_dev);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
int la_crop_repeat_grad_ex(const float* gxc, float* g_img, int B, int imgc, int R, int S, int y0, int x0, const int* pos_dev, int rep,
                           float scale, hipStream_t stream) {
    const float sc[4] = {scale, scale, scale, scale};
    LA_CHECK_ARG(rep >= 1 && rep <= 4, "crop_repeat_grad: rep must be 1..4");
    return la_crop_repeat_grad_ex3(gxc, g_img, B, imgc, R, S, y0, x0, pos_dev, rep, sc, stream);
}
extern "C" int la_crop_repeat_grad_f32(const float* gxc, float* g_img, int B, int imgc, int R, int S, int y0, int x0, int rep,
                                       float scale, hipStream_t stream) {
    return la_crop_repeat_grad_ex(gxc, g_img, B, imgc, R, S, y0, x0, nullptr, rep, scale, stream);
}
