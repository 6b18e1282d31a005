// StyleGAN2 discriminator (architecture 'resnet', no conditioning) forward + backward-to-image, for the criterion
//   loss_disc = softplus(-D(x, c=None)).mean() * w_disc            (augments/utils/util_latent_aug.py:363-371)
// D is frozen (:481): only d(loss)/d(image) is needed.  Layer definitions: SURVEY Appendix A; parameter names
// models/stylegan3/legacy.py:271-288; resampling algebra conv2d_resample.py:87-109 (Appendix B).
//
// Every 3x3 / 1x1 contraction reuses the implicit-GEMM kernels of la_conv*.hip (shared weights: no modulation);
// FIR stages reuse la_upfirdn2d.hip; what is new here: fromrgb (K = img_channels, HBM-bound), MinibatchStd, the FC tail.
#include "la_disc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "la_conv.h"
#include "la_modconv.h"
#include "la_style.h"
#include "la_upfirdn2d.h"

extern "C" int la_bias_act_grad_f32(const float* dy, const float* yref, float* dx, float* db, long n, long stepb, int nb, int act,
                                    float alpha, float gain, float clamp, hipStream_t stream);
extern "C" int la_fc_f32(const float* x, const float* W, const float* bias, float* y, int B, int in, int out, float lr_mul, int act,
                         float alpha, float gain, hipStream_t stream);

#define DMAX_BLOCKS 12

struct DConv {
    int cin, cout, k;
    int mb_;                     // backward output channels, padded to a multiple of 4 (= cin for every layer but b4.conv)
    const float *w, *bias;
    float *wf, *wb;
    void *wqf, *wqb;
};

struct DBlock {
    int res, cin, cout;          // cin = tmp_channels = C[res], cout = C[res/2]
    const float *frgb_w, *frgb_b;
    float *frgb_wt;              // [imgc][cin] = W^T * gain (for the backward 1x1)
    DConv conv0, conv1, skip;
    float *xin, *y0, *x1, *ysk, *sum;
};

struct la_disc {
    int R, imgc, nblocks, maxB, C4;
    float clamp;
    DBlock blk[DMAX_BLOCKS];
    DConv econv;                  // b4.conv: (C4 + 1) -> C4
    const float *fc_w, *fc_b, *out_w, *out_b;
    float *mb, *yc, *fc, *logits, *dlogits, *g_fc, *g_flat, *gA, *gB, *scrA, *scrB;
    float* pm;              // plane maxima of the gradient entering a backward contraction (la_conv_act_grad_pmax)
    float fir[16];
    void* cws; size_t cws_bytes;
    float* xs_b;            // backward slot rows [2 * nblocks][maxB][LA_XS_FAN] (la_common.h): fp16 operand scales of the gradient entering a block
                            // (lowered by the up-2 FIR that produces it) and of the masked gradient entering conv0's backward (lowered by the FIR adjoint)
    float* xs_f;            // forward slot rows [2 * nblocks + 1][maxB][LA_XS_FAN]: the input of block k (fromrgb output / residual sum of the block
                            // above; also bounds its FIR-down copy for the skip conv, and for k = nblocks the MinibatchStd output) and conv0's output y0
                            // (also bounds the pre-filtered copy conv1 reads), each lowered by the kernel that produces the tensor
    float* xs_fwd;          // [maxB] constant fp16 operand scale of every FORWARD contraction input: with a clamp all of them are bounded
                            // (clamped layer outputs, residual sums <= 2 * clamp / sqrt(2), FIR outputs <= their input, |std| <= max|x|)
    int precision, lastB, mbstd_group;
};

static inline int dz_xhalf(int res) { return (res / 2 + 1 + 3) & ~3; }      // column-planar rows of a (res+1)-wide intermediate: odd columns start here
static size_t alup(size_t v) { return (v + 63) & ~(size_t)63; }
struct DCarver {
    char* base; size_t off;
    float* take(size_t nfloats) { float* p = base ? (float*)(base + off) : nullptr; off += alup(nfloats * 4); return p; }
};

static void dconv_layout(DCarver& c, DConv& L) {
    L.mb_ = (L.cin + 3) & ~3;
    const size_t kk = (size_t)L.k * L.k;
    L.wf = c.take((size_t)L.cin * L.cout * kk); L.wb = c.take((size_t)L.mb_ * L.cout * kk);
    L.wqf = c.take((la_conv_split_pack_bytes(L.cout, L.cin, (int)kk) + 3) / 4);
    L.wqb = c.take((la_conv_split_pack_bytes(L.mb_, L.cout, (int)kk) + 3) / 4);
}

static int d_describe(la_disc* h, int R, int imgc, const int* channels, int maxB) {
    LA_CHECK_ARG(R >= 8 && (R & (R - 1)) == 0, "disc: resolution must be a power of two >= 8");
    LA_CHECK_ARG(imgc >= 1 && imgc <= 4 && maxB >= 1, "disc: bad img_channels / batch");
    memset(h, 0, sizeof(*h));
    h->R = R; h->imgc = imgc; h->maxB = maxB;
    int nb = 0;
    for (int r = R; r > 4; r >>= 1) ++nb;
    LA_CHECK_ARG(nb <= DMAX_BLOCKS, "disc: too many blocks");
    h->nblocks = nb;
    // channels[k] = C at resolution 4 << k (same table as the generator)
    int nres = 0;
    for (int r = 4; r <= R; r <<= 1) ++nres;
    for (int k = 0; k < nb; ++k) {
        DBlock& b = h->blk[k];
        b.res = R >> k;
        int idx = 0;
        for (int r = 4; r < b.res; r <<= 1) ++idx;
        b.cin = channels[idx]; b.cout = channels[idx - 1];
        LA_CHECK_ARG(b.cin % 4 == 0 && b.cout % 4 == 0, "disc: channel counts must be multiples of 4");
        b.conv0 = DConv{b.cin, b.cin, 3}; b.conv1 = DConv{b.cin, b.cout, 3}; b.skip = DConv{b.cin, b.cout, 1};
    }
    (void)nres;
    h->C4 = channels[0];
    h->econv = DConv{h->C4 + 1, h->C4, 3};
    return LA_OK;
}

static size_t d_layout(la_disc* h, void* ws) {
    DCarver c{(char*)ws, 0};
    const size_t mb = h->maxB;
    size_t gmax = 0, smax = 0, cw = 0, pmax = 0;
    for (int k = 0; k < h->nblocks; ++k) {
        DBlock& b = h->blk[k];
        const size_t hw = (size_t)b.res * b.res, hq = hw / 4;
        { const size_t p0 = mb * b.cin * la_conv_act_grad_segments((long)hw), p1 = mb * b.cout * la_conv_act_grad_segments((long)hq);
          if (p0 > pmax) pmax = p0; if (p1 > pmax) pmax = p1; }
        dconv_layout(c, b.conv0); dconv_layout(c, b.conv1); dconv_layout(c, b.skip);
        if (k == 0) { b.xin = c.take(mb * b.cin * hw); b.frgb_wt = c.take((size_t)h->imgc * b.cin); }
        b.y0 = c.take(mb * b.cin * hw);
        b.x1 = c.take(mb * b.cout * hq); b.ysk = c.take(mb * b.cout * hq); b.sum = c.take(mb * b.cout * hq);
        if (k + 1 < h->nblocks) h->blk[k + 1].xin = b.sum;
        if (mb * b.cin * hw > gmax) gmax = mb * b.cin * hw;
        // (backward: the transposed conv's (res+1)-row intermediate in column-planar rows of pitch 2 * dz_xhalf(res), as the generator's up layers)
        const size_t s1 = mb * b.cin * (size_t)(b.res + 1) * (size_t)(2 * dz_xhalf(b.res));
        if (s1 > smax) smax = s1;
        size_t w;
        w = la_modconv_workspace_bytes((int)mb, b.cin, b.cin, b.res, 0); if (w > cw) cw = w;
        w = la_modconv_workspace_bytes((int)mb, b.cout, b.cin, b.res, 1); if (w > cw) cw = w;       // stride-2 pair (either direction)
        w = la_modconv_workspace_bytes((int)mb, b.cin, b.cout, b.res, 1); if (w > cw) cw = w;
        w = la_modconv_workspace_bytes((int)mb, b.cin, b.cout, b.res / 2, 0); if (w > cw) cw = w;   // 1x1 at res/2
        w = la_modconv_workspace_bytes((int)mb, b.cout, b.cin, b.res / 2, 0); if (w > cw) cw = w;
    }
    dconv_layout(c, h->econv);
    { size_t w = la_modconv_workspace_bytes((int)mb, h->C4 + 1, h->C4 + 1, 4, 0); if (w > cw) cw = w; }
    h->mb = c.take(mb * (h->C4 + 1) * 16); h->yc = c.take(mb * h->C4 * 16); h->fc = c.take(mb * h->C4);
    h->xs_fwd = c.take(mb * LA_XS_FAN);
    h->xs_b = c.take((size_t)2 * DMAX_BLOCKS * mb * LA_XS_FAN);
    h->xs_f = c.take((size_t)(2 * DMAX_BLOCKS + 1) * mb * LA_XS_FAN);
    h->logits = c.take(mb); h->dlogits = c.take(mb); h->g_fc = c.take(mb * h->C4); h->g_flat = c.take(mb * (h->C4 + 1) * 16);
    h->gA = c.take(gmax); h->gB = c.take(gmax); h->scrA = c.take(smax); h->scrB = c.take(gmax);
    h->pm = c.take(pmax);
    h->cws = c.take((cw + 3) / 4); h->cws_bytes = cw;
    return c.off;
}

extern "C" int la_disc_num_params(int img_resolution) {
    int nb = 0;
    for (int r = img_resolution; r > 4; r >>= 1) ++nb;
    return 2 + nb * 5 + 6;
}

extern "C" size_t la_disc_workspace_bytes(int img_resolution, int img_channels, const int* channels, int max_batch) {
    la_disc* h = (la_disc*)malloc(sizeof(la_disc));
    if (!h) return 0;
    size_t need = 0;
    if (d_describe(h, img_resolution, img_channels, channels, max_batch) == LA_OK) need = d_layout(h, nullptr);
    free(h);
    return need;
}

__global__ void la_transpose_scale_kernel(const float* __restrict__ w, float* __restrict__ wt, int M, int C, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * C) return;
    const int m = i / C, c = i - m * C;
    wt[(long)c * M + m] = w[i] * scale;
}

static int dconv_pack(DConv& L, hipStream_t stream) {
    const float g = 1.0f / sqrtf((float)(L.cin * L.k * L.k));      // Conv2dLayer.weight_gain
    int rc = la_pack_conv_weights(L.w, L.wf, L.wb, nullptr, L.cout, L.cin, L.k * L.k, stream, g, L.mb_);
    if (!rc) rc = la_pack_conv_weights_bf16(L.w, L.wqf, L.cout, L.cin, L.k * L.k, 0, 3, stream, g);
    if (!rc) rc = la_pack_conv_weights_bf16(L.w, L.wqb, L.cout, L.cin, L.k * L.k, 1, 3, stream, g, L.mb_);
    return rc;
}

// params (device tensors, names of legacy.py:271-288), resolution R first:
//   bR: fromrgb.weight, fromrgb.bias, conv0.weight, conv0.bias, conv1.weight, conv1.bias, skip.weight
//   b(R/2)..b8: conv0.weight, conv0.bias, conv1.weight, conv1.bias, skip.weight
//   b4: conv.weight, conv.bias, fc.weight, fc.bias, out.weight, out.bias
extern "C" int la_disc_create(int img_resolution, int img_channels, const int* channels, float conv_clamp,
                              const float* const* params, int nparams, const float* fir_host, int mbstd_group_size,
                              int max_batch, void* workspace, size_t workspace_bytes, hipStream_t stream, la_disc** out) {
    LA_CHECK_ARG(params && fir_host && workspace && out, "disc_create: null pointer");
    la_disc* h = (la_disc*)malloc(sizeof(la_disc));
    LA_CHECK_ARG(h, "disc_create: out of host memory");
    int rc = d_describe(h, img_resolution, img_channels, channels, max_batch);
    if (rc) { free(h); return rc; }
    if (nparams != la_disc_num_params(img_resolution)) { free(h); la_set_error("disc_create: parameter list length mismatch"); return LA_ERR_ARG; }
    for (int i = 0; i < nparams; ++i) if (!params[i]) { free(h); la_set_error("disc_create: null parameter tensor"); return LA_ERR_ARG; }
    if (d_layout(h, workspace) > workspace_bytes) { free(h); la_set_error("disc_create: workspace too small"); return LA_ERR_WORKSPACE; }
    h->clamp = conv_clamp; h->mbstd_group = mbstd_group_size > 0 ? mbstd_group_size : 4;
    memcpy(h->fir, fir_host, sizeof(float) * 16);
    int p = 0;
    for (int k = 0; k < h->nblocks && !rc; ++k) {
        DBlock& b = h->blk[k];
        if (k == 0) {
            b.frgb_w = params[p++]; b.frgb_b = params[p++];
            hipLaunchKernelGGL(la_transpose_scale_kernel, dim3(la_cdiv((long)b.cin * h->imgc, 256)), dim3(256), 0, stream, b.frgb_w,
                               b.frgb_wt, b.cin, h->imgc, 1.0f / sqrtf((float)h->imgc));
        }
        b.conv0.w = params[p++]; b.conv0.bias = params[p++];
        b.conv1.w = params[p++]; b.conv1.bias = params[p++];
        b.skip.w = params[p++]; b.skip.bias = nullptr;
        rc = dconv_pack(b.conv0, stream);
        if (!rc) rc = dconv_pack(b.conv1, stream);
        if (!rc) rc = dconv_pack(b.skip, stream);
    }
    if (!rc) {
        h->econv.w = params[p++]; h->econv.bias = params[p++];
        h->fc_w = params[p++]; h->fc_b = params[p++]; h->out_w = params[p++]; h->out_b = params[p++];
        rc = dconv_pack(h->econv, stream);
    }
    if (rc) { free(h); return rc; }
    {   // bound 4 * clamp covers every forward input (see xs_fwd); the scale is the power of two that puts the bound in [2^14, 2^15)
        int e = 0;
        frexpf(4.f * (conv_clamp > 0.f ? conv_clamp : 1.f), &e);
        const int nb = max_batch < 256 ? max_batch : 256;
        float* hx = (float*)malloc(sizeof(float) * (size_t)nb * LA_XS_FAN);      // (slot rows: every sub-slot holds the constant)
        if (!hx) { free(h); la_set_error("disc_create: out of host memory"); return LA_ERR_ARG; }
        for (size_t i = 0; i < (size_t)nb * LA_XS_FAN; ++i) hx[i] = ldexpf(1.f, 15 - e);
        const hipError_t ce = hipMemcpyAsync(h->xs_fwd, hx, sizeof(float) * (size_t)nb * LA_XS_FAN, hipMemcpyHostToDevice, stream);
        const hipError_t se = hipStreamSynchronize(stream);
        free(hx);
        if (ce != hipSuccess || se != hipSuccess) { free(h); la_set_error("disc_create: copying the operand scales failed"); return LA_ERR_HIP; }
    }
    *out = h;
    return LA_OK;
}

extern "C" void la_disc_destroy(la_disc* h) { free(h); }
extern "C" int la_disc_set_precision(la_disc* h, int precision) {
    LA_CHECK_ARG(h && precision >= 0 && precision <= 3, "disc_set_precision: precision must be 0..3");
    h->precision = precision;
    return LA_OK;
}
extern "C" const float* la_disc_logits(const la_disc* h) { return h ? h->logits : nullptr; }

// ------------------------------------------------------------------------------------------------------------
// fromrgb: y[b][m][p] = act(sum_c W[m][c]*g * img[b][c][p] + bias[m]); one thread per 4 pixels, loops over m (HBM-bound on y)
template <int IMGC>
__global__ __launch_bounds__(256) void la_fromrgb_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int M,
                                                            long HW, float wgain, float alpha, float gain, float clamp, float* __restrict__ xs_rows) {
    const int b = blockIdx.y;
    const long p4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    float ymax = 0.f;
    if (p4 < HW) {
    float4 x[IMGC];
#pragma unroll
    for (int c = 0; c < IMGC; ++c) x[c] = *reinterpret_cast<const float4*>(img + ((long)b * IMGC + c) * HW + p4);
    for (int m = 0; m < M; ++m) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < IMGC; ++c) {
            const float wv = w[m * IMGC + c] * wgain;
            v.x += wv * x[c].x; v.y += wv * x[c].y; v.z += wv * x[c].z; v.w += wv * x[c].w;
        }
        const float bv = bias[m];
        v.x = la_act_fwd(v.x + bv, LA_ACT_LRELU, alpha, gain, clamp); v.y = la_act_fwd(v.y + bv, LA_ACT_LRELU, alpha, gain, clamp);
        v.z = la_act_fwd(v.z + bv, LA_ACT_LRELU, alpha, gain, clamp); v.w = la_act_fwd(v.w + bv, LA_ACT_LRELU, alpha, gain, clamp);
        *reinterpret_cast<float4*>(y + ((long)b * M + m) * HW + p4) = v;
        ymax = fmaxf(ymax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    }
    if (xs_rows) {      // fp16 operand scale of y for the contractions that read it: every wave lowers a sub-slot of the sample's row (la_common.h)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
        if ((threadIdx.x & 63) == 0) {
            float* row = xs_rows + (long)b * LA_XS_FAN + la_xs_sub((int)(threadIdx.x >> 6));
            la_xs_lower(row, la_xs_peek(row), 1.f, ymax);
        }
    }
}

// MinibatchStd (group G, 1 statistic channel): forward writes [N][C+1][HW]; sample n belongs to slot n % (N/G).
// xs_rows (fp16 slot rows of the epilogue conv's input, or null): the statistic channel is a GROUP value -- bounded by the group's largest
// member, not by sample n's own maximum, which is all that the producer of `x` lowered n's row with -- so every member's row is
// lowered with it here (a sample much smaller than its group-mates would otherwise overflow its std-channel operand in fp16).
__global__ __launch_bounds__(256) void la_mbstd_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int G,
                                                          int C, int HW, float* __restrict__ xs_rows) {
    __shared__ float red[4];
    const int slot = blockIdx.x, M = N / G;
    const int CHW = C * HW;
    // (only N / G workgroups run: four elements per thread and pass with all their group loads in flight, accumulated in the order of
    //  the plain loop -- it was one exposed load latency per element and group member, 62 us)
    float acc = 0.f;
    auto one = [&](const float (&xv)[8]) {      // (fixed trip counts: the arrays stay in registers)
        float mean = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) if (g < G) mean += xv[g];
        mean /= (float)G;
        float var = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) if (g < G) { const float d = xv[g] - mean; var += d * d; }
        return sqrtf(var / (float)G + 1e-8f);
    };
    if (G <= 8) {
        int e = threadIdx.x;
        for (; e + 3 * (int)blockDim.x < CHW; e += 4 * blockDim.x) {
            float xv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int g = 0; g < 8; ++g) xv[u][g] = g < G ? x[(long)(g * M + slot) * CHW + e + u * (int)blockDim.x] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += one(xv[u]);
        }
        for (; e < CHW; e += blockDim.x) {
            float xv[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) xv[g] = g < G ? x[(long)(g * M + slot) * CHW + e] : 0.f;
            acc += one(xv);
        }
    } else
    for (int e = threadIdx.x; e < CHW; e += blockDim.x) {
        float mean = 0.f;
        for (int g = 0; g < G; ++g) mean += x[(long)(g * M + slot) * CHW + e];
        mean /= (float)G;
        float var = 0.f;
        for (int g = 0; g < G; ++g) { const float d = x[(long)(g * M + slot) * CHW + e] - mean; var += d * d; }
        acc += sqrtf(var / (float)G + 1e-8f);
    }
    const float stat = la_block_sum_256(acc, red) / (float)CHW;
    if (xs_rows && threadIdx.x < (unsigned)G) {
        float* row = xs_rows + (long)((int)threadIdx.x * M + slot) * LA_XS_FAN + la_xs_sub((int)threadIdx.x);
        la_xs_lower(row, la_xs_peek(row), 1.f, stat);
    }
    for (int g = 0; g < G; ++g) {
        const long n = g * M + slot;
        int e = threadIdx.x;
        for (; e + 3 * (int)blockDim.x < CHW; e += 4 * blockDim.x) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = x[n * CHW + e + u * (int)blockDim.x];
#pragma unroll
            for (int u = 0; u < 4; ++u) y[n * (CHW + HW) + e + u * (int)blockDim.x] = v[u];
        }
        for (; e < CHW; e += blockDim.x) y[n * (CHW + HW) + e] = x[n * CHW + e];
        for (int q = threadIdx.x; q < HW; q += blockDim.x) y[n * (CHW + HW) + CHW + q] = stat;
    }
}

// backward: gx = gy[:, :C] + d(stat)/dx * sum(gy[:, C]) over the slot's members and pixels
__global__ __launch_bounds__(256) void la_mbstd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          float* __restrict__ gx, int N, int G, int C, int HW, int CP) {
    __shared__ float red[4];
    const int slot = blockIdx.x, M = N / G;
    const int CHW = C * HW;
    float gs = 0.f;
    for (int e = threadIdx.x; e < G * HW; e += blockDim.x) {
        const int g = e / HW, p = e - g * HW;
        gs += gy[(long)(g * M + slot) * ((long)CP * HW) + CHW + p];
    }
    const float gstat = la_block_sum_256(gs, red);
    // (as in the forward kernel: two elements per thread and pass, their 2 x 2G loads in flight, same arithmetic per element)
    auto elem = [&](int e, const float (&xv)[8], const float (&gv)[8]) {
        float mean = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) if (g < G) mean += xv[g];
        mean /= (float)G;
        float var = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) if (g < G) { const float d = xv[g] - mean; var += d * d; }
        const float sd = sqrtf(var / (float)G + 1e-8f);
        const float k = gstat / ((float)CHW * (float)G * sd);
#pragma unroll
        for (int g = 0; g < 8; ++g) if (g < G) gx[(long)(g * M + slot) * CHW + e] = gv[g] + k * (xv[g] - mean);
    };
    if (G <= 8) {
        int e = threadIdx.x;
        for (; e + (int)blockDim.x < CHW; e += 2 * blockDim.x) {
            float xv[2][8], gv[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const long n = g * M + slot;
                    xv[u][g] = g < G ? x[n * CHW + e + u * (int)blockDim.x] : 0.f;
                    gv[u][g] = g < G ? gy[n * ((long)CP * HW) + e + u * (int)blockDim.x] : 0.f;
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) elem(e + u * (int)blockDim.x, xv[u], gv[u]);
        }
        for (; e < CHW; e += blockDim.x) {
            float xv[8], gv[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const long n = g * M + slot;
                xv[g] = g < G ? x[n * CHW + e] : 0.f;
                gv[g] = g < G ? gy[n * ((long)CP * HW) + e] : 0.f;
            }
            elem(e, xv, gv);
        }
        return;
    }
    for (int e = threadIdx.x; e < CHW; e += blockDim.x) {
        float mean = 0.f;
        for (int g = 0; g < G; ++g) mean += x[(long)(g * M + slot) * CHW + e];
        mean /= (float)G;
        float var = 0.f;
        for (int g = 0; g < G; ++g) { const float d = x[(long)(g * M + slot) * CHW + e] - mean; var += d * d; }
        const float sd = sqrtf(var / (float)G + 1e-8f);
        const float k = gstat / ((float)CHW * (float)G * sd);
        for (int g = 0; g < G; ++g) {
            const long n = g * M + slot;
            gx[n * CHW + e] = gy[n * ((long)CP * HW) + e] + k * (x[n * CHW + e] - mean);
        }
    }
}

// gx[b][i] = wgain * sum_o g[b][o] * W[o][i]          (FullyConnectedLayer backward-data)
// Workgroup = FC_IL consecutive i (coalesced rows of W) x FC_NG groups of output rows; every weight is read once for up to FCB samples,
// four rows in flight; the row groups are combined through LDS in a fixed order.  (Round 1: one thread per i and sample walking all
// `out` rows alone: 112 us on the 8192 x 512 epilogue layer.)
#define FCB 8
#define FC_IL 32      // consecutive inputs per workgroup (one 128-byte segment of every weight row)
#define FC_NG 8       // groups of output rows per workgroup (256 / FC_IL): 256 workgroups x 8 groups on the 8192-wide layer (round 2: 64 x 4,
                      // i.e. half the workgroups and twice the rows per thread: 40 us for 16.8 MB)
__global__ __launch_bounds__(256) void la_fc_bwd_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                       float* __restrict__ gx, int B, int in, int out, float wgain) {
    __shared__ float comb[FC_NG][FCB][FC_IL];
    const int il = threadIdx.x % FC_IL, og = threadIdx.x / FC_IL;
    const int i = blockIdx.x * FC_IL + il;
    const int b0 = blockIdx.y * FCB;
    const int per = (out + FC_NG - 1) / FC_NG;
    const int o0 = og * per, o1 = o0 + per < out ? o0 + per : out;
    float acc[FCB];
#pragma unroll
    for (int q = 0; q < FCB; ++q) acc[q] = 0.f;
    if (i < in) {
        int o = o0;
        for (; o + 3 < o1; o += 4) {
            const float w0 = W[(long)o * in + i], w1 = W[(long)(o + 1) * in + i], w2 = W[(long)(o + 2) * in + i], w3 = W[(long)(o + 3) * in + i];
#pragma unroll
            for (int q = 0; q < FCB; ++q)
                if (b0 + q < B) {
                    const float* gb = g + (long)(b0 + q) * out + o;
                    acc[q] += (gb[0] * w0 + gb[1] * w1) + (gb[2] * w2 + gb[3] * w3);
                }
        }
        for (; o < o1; ++o) {
            const float w0 = W[(long)o * in + i];
#pragma unroll
            for (int q = 0; q < FCB; ++q)
                if (b0 + q < B) acc[q] += g[(long)(b0 + q) * out + o] * w0;
        }
    }
#pragma unroll
    for (int q = 0; q < FCB; ++q) comb[og][q][il] = acc[q];
    __syncthreads();
    if (og == 0 && i < in) {
#pragma unroll
        for (int q = 0; q < FCB; ++q)
            if (b0 + q < B) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < FC_NG; ++k) t += comb[k][q][il];
                gx[(long)(b0 + q) * in + i] = t * wgain;
            }
    }
}

// loss = mean softplus(-logit) * w ; dlogit = -sigmoid(-logit) * w / n            (util_latent_aug.py:367-369)
__global__ void la_disc_loss_kernel(const float* __restrict__ logits, float* __restrict__ dlogits, float* __restrict__ loss,
                                    int B, float w, float nb) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float z = -logits[b];
            acc += z > 20.f ? z : log1pf(expf(z));          // torch softplus (beta 1, threshold 20)
            dlogits[b] = -(1.f / (1.f + expf(-z))) * w / nb;
        }
        if (loss) loss[0] = acc / nb * w;
    }
}

static void cbase(LaConvArgs& a) {
    memset(&a, 0, sizeof(a));
    a.in_sy = a.in_sx = a.out_sy = a.out_sx = 1; a.clamp = -1.f; a.gain = 1.f; a.act = LA_ACT_LINEAR;
}

static void set_w(LaConvArgs& a, la_disc* h, const DConv& L, bool backward) {
    a.wgt = backward ? L.wb : L.wf;
    a.precision = h->precision; a.wgt_bf16 = backward ? L.wqb : L.wqf;
    a.wgt_bf16_term_elems = la_conv_bf16_pack_elems(backward ? L.mb_ : L.cout, backward ? L.cout : L.cin, L.k * L.k);
    a.ws = h->cws; a.ws_bytes = h->cws_bytes;
}

// dense conv (k = 3 pad 1, or k = 1) at one resolution, forward (with bias/act epilogue) or backward-data (plain)
static int conv_same(la_disc* h, const DConv& L, bool backward, const float* in, float* out, int B, int res, int act, float gain,
                     float clamp, const float* addend, float* out2, hipStream_t stream, const float* in_pmax = nullptr, int in_nseg = 0,
                     const float* xs_rows = nullptr, float in_gain = 1.f, float* xs_out = nullptr) {
    LaConvArgs a; cbase(a);
    set_w(a, h, L, backward);
    a.in = in; a.out = out; a.B = B; a.in_pmax = in_pmax; a.in_pmax_nseg = in_nseg;
    if (xs_rows) { a.acc_scale_x = xs_rows; a.acc_scale_fan = LA_XS_FAN; a.in_pmax = nullptr; }      // slot rows left by the producer of `in`
    a.in_gain = in_gain;
    a.C = backward ? L.cout : L.cin; a.M = backward ? L.mb_ : L.cout;      // (padded channels of a backward come out as zeros)
    a.in_bstride = (long)a.C * res * res;
    a.Hin = a.Win = a.Hout = a.Wout = a.Gy = a.Gx = res;
    a.ntaps = L.k * L.k;
    for (int t = 0; t < a.ntaps; ++t) {
        const int ky = t / L.k, kx = t % L.k, pad = L.k / 2;
        a.tap_dy[t] = backward ? pad - ky : ky - pad; a.tap_dx[t] = backward ? pad - kx : kx - pad; a.tap_w[t] = t;
    }
    if (backward) { a.epi = LA_EPI_BWD; }
    else {
        a.epi = LA_EPI_FWD; a.bias = L.bias; a.act = act; a.alpha = 0.2f; a.gain = gain; a.clamp = clamp; a.addend = addend; a.out2 = out2;
        a.fwd_xs_out = xs_out;
        // (no slot rows given: the constant a-priori scale -- every forward input of D is bounded by 4 * conv_clamp)
        if (!xs_rows && h->precision == LA_PREC_F16X2 && h->clamp > 0.f && h->maxB <= 256) { a.acc_scale_x = h->xs_fwd; a.acc_scale_fan = LA_XS_FAN; }
    }
    return la_conv_launch(a, stream);
}

extern "C" int la_disc_forward(la_disc* h, const float* img, int B, hipStream_t stream) {
    LA_CHECK_ARG(h && img, "disc_forward: null pointer");
    LA_CHECK_ARG(B >= 1 && B <= h->maxB, "disc_forward: batch exceeds max_batch");
    const int G = h->mbstd_group < B ? h->mbstd_group : B;
    LA_CHECK_ARG(B % G == 0, "disc_forward: batch must be divisible by the MinibatchStd group size (as in the reference)");
    const float sq2 = sqrtf(2.f), rs2 = sqrtf(0.5f);
    int rc;
    // fp16 x2 mode: every contraction input's operand scale comes from the kernel that produced the tensor (slot rows, la_common.h):
    // rows_x(k) = input of block k (k = nblocks: input of the epilogue conv), rows_y(k) = conv0's output of block k.  A FIR with taps
    // summing to 1 (the skip path's FIR-down, conv1's pre-filter) and MinibatchStd do not raise the maximum: their outputs share the rows
    // of their input.
    const bool slots = h->precision == LA_PREC_F16X2 && !la_dev_env("LA_NO_DISC_FUSE");
    if (slots) LA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->xs_f), (int)LA_XS_INIT, (size_t)(2 * h->nblocks + 1) * h->maxB * LA_XS_FAN, stream));
    auto rows_x = [&](int k) { return slots ? h->xs_f + (size_t)k * h->maxB * LA_XS_FAN : nullptr; };
    auto rows_y = [&](int k) { return slots ? h->xs_f + (size_t)(h->nblocks + 1 + k) * h->maxB * LA_XS_FAN : nullptr; };
    for (int k = 0; k < h->nblocks; ++k) {
        DBlock& b = h->blk[k];
        const int res = b.res, hq = res / 2;
        const long HW = (long)res * res;
        if (k == 0) {
            dim3 grid(la_cdiv(HW / 4, 256), B);
#define FRGB(N) hipLaunchKernelGGL(la_fromrgb_fwd_kernel<N>, grid, dim3(256), 0, stream, img, b.frgb_w, b.frgb_b, b.xin, b.cin, HW, 1.0f / sqrtf((float)h->imgc), 0.2f, sq2, h->clamp, rows_x(0))
            switch (h->imgc) { case 1: FRGB(1); break; case 2: FRGB(2); break; case 3: FRGB(3); break; default: FRGB(4); }
#undef FRGB
            LA_CHECK_LAUNCH();
        }
        // skip: FIR (pad 1,1,1,1) + decimate 2, then 1x1 conv, linear * sqrt(1/2)      (conv2d_resample.py:94-97)
        if ((rc = la_upfirdn2d_ex(b.xin, h->scrB, B, b.cin, res, res, h->fir, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1, 0, 1.f, nullptr, stream))) return rc;
        if ((rc = conv_same(h, b.skip, false, h->scrB, b.ysk, B, hq, LA_ACT_LINEAR, rs2, -1.f, nullptr, nullptr, stream, nullptr, 0, rows_x(k)))) return rc;
        // conv0
        if ((rc = conv_same(h, b.conv0, false, b.xin, b.y0, B, res, LA_ACT_LRELU, sq2, h->clamp, nullptr, nullptr, stream, nullptr, 0, rows_x(k), 1.f, rows_y(k)))) return rc;
        // conv1: FIR pad (2,2,2,2) -> (res+1)^2, stride-2 conv, lrelu * sqrt2 * sqrt(1/2), clamp * sqrt(1/2); + skip  (:106-109)
        // fp16 x2 mode with the a-priori operand scale: the FIR writes its result ALREADY as the contraction's packed operand (scale,
        // fp16 split, channel interleave: la_fir4x4_adj_pack_kernel with the forward taps) -- one pass instead of the scalar FIR into a
        // (res+1)^2 fp32 scratch + the pre-split copy of that scratch
        const size_t qbytes = (size_t)B * la_cdiv(b.cin, 32) * 32 * (res + 1) * (res + 1) * 4;
        const bool fir_pack = slots && res % 4 == 0 && h->cws_bytes > qbytes + 1024;
        if (!fir_pack && (rc = la_upfirdn2d_ex(b.y0, h->scrA, B, b.cin, res, res, h->fir, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, 0, 1.f, nullptr, stream))) return rc;
        {
            LaConvArgs a; cbase(a);
            set_w(a, h, b.conv1, false);
            if (fir_pack) {
                unsigned* q = reinterpret_cast<unsigned*>(h->cws);
                if ((rc = la_fir4x4_adjoint_pack_f16(b.y0, q, rows_y(k), LA_XS_FAN, B, b.cin, res, res, h->fir, 1.f, stream, 1))) return rc;
                const size_t used = (qbytes + 255) & ~(size_t)255;
                a.in_q = q; a.ws = static_cast<char*>(h->cws) + used; a.ws_bytes = h->cws_bytes - used;
            }
            a.in = h->scrA; a.in_bstride = (long)b.cin * (res + 1) * (res + 1); a.out = b.x1;
            a.B = B; a.C = b.cin; a.M = b.cout; a.Hin = a.Win = res + 1; a.Hout = a.Wout = a.Gy = a.Gx = hq;
            a.in_sy = a.in_sx = 2; a.ntaps = 9;
            for (int t = 0; t < 9; ++t) { a.tap_dy[t] = t / 3; a.tap_dx[t] = t % 3; a.tap_w[t] = t; }
            a.epi = LA_EPI_FWD; a.bias = b.conv1.bias; a.act = LA_ACT_LRELU; a.alpha = 0.2f; a.gain = sq2 * rs2;
            a.clamp = h->clamp >= 0.f ? h->clamp * rs2 : -1.f;
            a.addend = b.ysk; a.out2 = b.sum;
            a.fwd_xs_out = rows_x(k + 1);      // (the residual sum out2 is what the next block reads)
            if (slots) { a.acc_scale_x = rows_y(k); a.acc_scale_fan = LA_XS_FAN; }
            else if (h->precision == LA_PREC_F16X2 && h->clamp > 0.f && h->maxB <= 256) { a.acc_scale_x = h->xs_fwd; a.acc_scale_fan = LA_XS_FAN; }
            if ((rc = la_conv_launch(a, stream))) return rc;
        }
    }
    const DBlock& last = h->blk[h->nblocks - 1];
    hipLaunchKernelGGL(la_mbstd_fwd_kernel, dim3(B / G), dim3(256), 0, stream, last.sum, h->mb, B, G, h->C4, 16, rows_x(h->nblocks));
    LA_CHECK_LAUNCH();
    if ((rc = conv_same(h, h->econv, false, h->mb, h->yc, B, 4, LA_ACT_LRELU, sq2, h->clamp, nullptr, nullptr, stream, nullptr, 0, rows_x(h->nblocks)))) return rc;
    if ((rc = la_fc_f32(h->yc, h->fc_w, h->fc_b, h->fc, B, h->C4 * 16, h->C4, 1.f, LA_ACT_LRELU, 0.2f, sq2, stream))) return rc;
    if ((rc = la_fc_f32(h->fc, h->out_w, h->out_b, h->logits, B, h->C4, 1, 1.f, LA_ACT_LINEAR, 0.f, 1.f, stream))) return rc;
    h->lastB = B;
    return LA_OK;
}

// loss_out[0] = softplus(-logits).mean() * w_disc (norm_batch = n of the mean; 0 = B); dlogits kept for la_disc_backward
extern "C" int la_disc_loss(la_disc* h, float w_disc, int norm_batch, float* loss_out, hipStream_t stream) {
    LA_CHECK_ARG(h && h->lastB >= 1, "disc_loss: no forward pass");
    hipLaunchKernelGGL(la_disc_loss_kernel, dim3(1), dim3(64), 0, stream, h->logits, h->dlogits, loss_out, h->lastB, w_disc,
                       (float)(norm_batch > 0 ? norm_batch : h->lastB));
    LA_CHECK_LAUNCH();
    return LA_OK;
}

__global__ void la_outfc_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ w, float* __restrict__ g, int B,
                                    int n, float wgain) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n) return;
    g[i] = dlogits[i / n] * w[i % n] * wgain;
}

// d(loss)/d(img) from dlogits (set by la_disc_loss, or given).  accumulate != 0: g_img += result.
extern "C" int la_disc_backward(la_disc* h, const float* dlogits, float* g_img, int accumulate, hipStream_t stream) {
    LA_CHECK_ARG(h && g_img && h->lastB >= 1, "disc_backward: null pointer / no forward pass");
    const int B = h->lastB;
    const int G = h->mbstd_group < B ? h->mbstd_group : B;
    const float sq2 = sqrtf(2.f), rs2 = sqrtf(0.5f);
    const float* dl = dlogits ? dlogits : h->dlogits;
    int rc;
    const int C4 = h->C4;
    hipLaunchKernelGGL(la_outfc_bwd_kernel, dim3(la_cdiv((long)B * C4, 256)), dim3(256), 0, stream, dl, h->out_w, h->g_fc, B, C4,
                       1.0f / sqrtf((float)C4));
    if ((rc = la_bias_act_grad_f32(h->g_fc, h->fc, h->g_fc, nullptr, (long)B * C4, 1, 1, LA_ACT_LRELU, 0.2f, sq2, -1.f, stream))) return rc;
    hipLaunchKernelGGL(la_fc_bwd_kernel, dim3(la_cdiv(C4 * 16, FC_IL), la_cdiv(B, FCB)), dim3(256), 0, stream, h->g_fc, h->fc_w, h->g_flat, B, C4 * 16,
                       C4, 1.0f / sqrtf((float)(C4 * 16)));
    LA_CHECK_LAUNCH();
    if ((rc = la_bias_act_grad_f32(h->g_flat, h->yc, h->g_flat, nullptr, (long)B * C4 * 16, 1, 1, LA_ACT_LRELU, 0.2f, sq2, h->clamp, stream))) return rc;
    // b4.conv backward-data: [B][C4][16] -> [B][C4+1][16]
    if ((rc = conv_same(h, h->econv, true, h->g_flat, h->scrB, B, 4, 0, 0.f, 0.f, nullptr, nullptr, stream))) return rc;
    const DBlock& last = h->blk[h->nblocks - 1];
    hipLaunchKernelGGL(la_mbstd_bwd_kernel, dim3(B / G), dim3(256), 0, stream, last.sum, h->scrB, h->gA, B, G, C4, 16, h->econv.mb_);
    LA_CHECK_LAUNCH();
    float* g_sum = h->gA;       // gradient w.r.t. the current block's output (sum)
    float* other = h->gB;
    // fp16 x2 mode: no activation-backward sweeps.  The gradient entering a block comes out of the up-2 FIR of the block above, which
    // lowers its operand-scale slot rows; the two contractions that read it take their element-wise factor -- act'(x1) of conv1,
    // sqrt(1/2) of the skip branch -- inside their pre-split copy (LaConvArgs::in_mask_y / in_gain); the FIR adjoint behind the
    // transposed conv applies act'(y0) and lowers the rows of conv0's backward contraction.  (Round 3: three sweeps of
    // la_act_grad_pmax_kernel + three scale reductions per block.)  The last block's gradient comes from MinibatchStd: sweeps as before.
    const bool fuse = h->precision == LA_PREC_F16X2 && !la_dev_env("LA_NO_DISC_FUSE");
    if (fuse) LA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->xs_b), (int)LA_XS_INIT, (size_t)2 * h->nblocks * h->maxB * LA_XS_FAN, stream));
    auto rows_g = [&](int k) { return h->xs_b + (size_t)(2 * k) * h->maxB * LA_XS_FAN; };
    auto rows_c0 = [&](int k) { return h->xs_b + (size_t)(2 * k + 1) * h->maxB * LA_XS_FAN; };
    auto zx_of = [&](int res) { return (fuse && h->precision != LA_PREC_F32 && res % 4 == 0) ? dz_xhalf(res) : 0; };
    for (int k = h->nblocks - 1; k >= 0; --k) {
        DBlock& b = h->blk[k];
        const int res = b.res, hq = res / 2;
        const bool gs = fuse && k < h->nblocks - 1;      // g_sum came out of the up-2 FIR of block k + 1 with its slot rows
        // ---- conv1 branch: act' (gain 1, clamp*sqrt(1/2)) -> transposed stride-2 conv -> FIR adjoint (pad 1)
        const int nsq = la_conv_act_grad_segments((long)hq * hq), nsf = la_conv_act_grad_segments((long)res * res);
        if (!gs && (rc = la_conv_act_grad_pmax(g_sum, b.x1, h->scrB, h->pm, B, b.cout, (long)hq * hq, LA_ACT_LRELU, 0.2f, sq2 * rs2,
                                               h->clamp >= 0.f ? h->clamp * rs2 : -1.f, stream))) return rc;
        {
            LaConvArgs a; cbase(a);
            set_w(a, h, b.conv1, true);
            a.in = gs ? g_sum : h->scrB; a.in_bstride = (long)b.cout * hq * hq; a.out = h->scrA;
            a.B = B; a.C = b.cout; a.M = b.cin; a.Hin = a.Win = hq; a.Hout = a.Wout = res + 1;
            a.out_sy = a.out_sx = 2; a.epi = LA_EPI_RAW;
            // fused tail: the intermediate in column-planar rows (even columns | odd columns: every phase stores contiguous runs), read by
            // the planar vector FIR kernel (la_fir4x4_s1p_kernel) -- the scalar kernel on dense rows took 291 us at 8 x 128 x 256^2
            const int zx = zx_of(res), zp = 2 * zx;
            if (zx) { a.out_pitch = zp; a.out_plane = (long)zp * (res + 1); a.out_sx = 1; a.Wout = zp; }
            if (gs) {
                a.in_mask_y = b.x1; a.in_mask_act = LA_ACT_LRELU; a.in_mask_alpha = 0.2f; a.in_mask_gain = sq2 * rs2;
                a.in_mask_clamp = h->clamp >= 0.f ? h->clamp * rs2 : -1.f;
                a.acc_scale_x = rows_g(k); a.acc_scale_fan = LA_XS_FAN;
            } else { a.in_pmax = h->pm; a.in_pmax_nseg = nsq; }
            if (h->precision != LA_PREC_F32 && (rc = la_conv_prepare_input(a, stream))) return rc;   // split once for the four phases
            const bool merged = h->precision != LA_PREC_F32;      // 16-bit kernels: the four phases in ONE launch (as the generator's up layers)
            int np = 0;
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px) {
                    a.out_oy = py; a.out_ox = zx ? px * zx : px; a.Gy = py ? hq : hq + 1; a.Gx = px ? hq : hq + 1;
                    int nt = 0;
                    for (int ky = py; ky < 3; ky += 2)
                        for (int kx = px; kx < 3; kx += 2) { a.tap_dy[nt] = -(ky / 2); a.tap_dx[nt] = -(kx / 2); a.tap_w[nt] = ky * 3 + kx; ++nt; }
                    a.ntaps = nt;
                    if (merged) {
                        LaConvArgs::Phase& P = a.ph[np++];
                        P.Gy = a.Gy; P.Gx = a.Gx; P.out_oy = py; P.out_ox = a.out_ox; P.ntaps = nt;
                        for (int t = 0; t < nt; ++t) { P.tap_dy[t] = a.tap_dy[t]; P.tap_dx[t] = a.tap_dx[t]; P.tap_w[t] = a.tap_w[t]; }
                        continue;
                    }
                    if ((rc = la_conv_launch(a, stream))) return rc;
                }
            if (merged) {
                a.nphase = np;
                a.out_oy = a.out_ox = 0; a.Gy = a.Gx = hq + 1; a.ntaps = 4;      // launch-wide fields = the largest phase (checks only)
                if ((rc = la_conv_launch(a, stream))) return rc;
            }
        }
        if (fuse) {
            // FIR adjoint + act'(y0) + the slot rows of conv0's backward contraction in one kernel
            LaFirTail tail{b.y0, LA_ACT_LRELU, 0.2f, sq2, h->clamp, rows_c0(k)};
            if (zx_of(res)) { tail.in_pitch = 2 * zx_of(res); tail.in_plane = (long)tail.in_pitch * (res + 1); tail.in_xhalf = zx_of(res); }
            if ((rc = la_upfirdn2d_ex(h->scrA, other, B, b.cin, res + 1, res + 1, h->fir, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1.f, nullptr, stream, nullptr, &tail))) return rc;
            if ((rc = conv_same(h, b.conv0, true, other, h->scrA, B, res, 0, 0.f, 0.f, nullptr, nullptr, stream, nullptr, 0, rows_c0(k)))) return rc;
        } else {
            if ((rc = la_upfirdn2d_ex(h->scrA, other, B, b.cin, res + 1, res + 1, h->fir, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1.f, nullptr, stream))) return rc;
            // ---- conv0: act' then backward-data
            if ((rc = la_conv_act_grad_pmax(other, b.y0, other, h->pm, B, b.cin, (long)res * res, LA_ACT_LRELU, 0.2f, sq2, h->clamp, stream))) return rc;
            if ((rc = conv_same(h, b.conv0, true, other, h->scrA, B, res, 0, 0.f, 0.f, nullptr, nullptr, stream, h->pm, nsf))) return rc;   // scrA >= B*cin*res^2
        }
        // ---- skip branch: * sqrt(1/2) -> 1x1 adjoint -> FIR-down adjoint (up 2, pad (2,1,2,1), flipped), added to the conv branch
        if (gs) {
            // (the pre-split copy of g_sum carries the factor; the contraction then overwrites the g_sum buffer: [B][cin][hq^2])
            if ((rc = conv_same(h, b.skip, true, g_sum, g_sum, B, hq, 0, 0.f, 0.f, nullptr, nullptr, stream, nullptr, 0, rows_g(k), rs2))) return rc;
        } else {
            if ((rc = la_conv_act_grad_pmax(g_sum, b.ysk, h->scrB, h->pm, B, b.cout, (long)hq * hq, LA_ACT_LINEAR, 0.f, rs2, -1.f, stream))) return rc;
            if ((rc = conv_same(h, b.skip, true, h->scrB, g_sum, B, hq, 0, 0.f, 0.f, nullptr, nullptr, stream, h->pm, nsq))) return rc;       // reuse g_sum buffer: [B][cin][hq^2]
        }
        {
            LaFirTail tail{nullptr, 0, 0.f, 0.f, 0.f, (fuse && k > 0) ? rows_g(k - 1) : nullptr};
            if ((rc = la_upfirdn2d_ex(g_sum, other, B, b.cin, hq, hq, h->fir, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1, 1, 1.f, h->scrA, stream, nullptr, &tail))) return rc;
        }
        // `other` now holds d/d(xin) of this block
        float* t = g_sum; g_sum = other; other = t;
    }
    // fromrgb backward: act' then the 1x1 adjoint onto the image channels (streams the gradient once)
    DBlock& b0 = h->blk[0];
    const long n0 = (long)B * b0.cin * h->R * h->R;
    if (fuse && (long)h->R * h->R > 4096) {      // act' inside the 1x1's stream of the gradient
        const LaTorgbMask mk{b0.xin, LA_ACT_LRELU, 0.2f, sq2, h->clamp};
        return la_torgb_forward(g_sum, b0.frgb_wt, nullptr, 0, nullptr, accumulate ? g_img : nullptr, nullptr, g_img, B, b0.cin, h->imgc, h->R,
                                h->R, -1.f, stream, &mk);
    }
    if ((rc = la_bias_act_grad_f32(g_sum, b0.xin, g_sum, nullptr, n0, 1, 1, LA_ACT_LRELU, 0.2f, sq2, h->clamp, stream))) return rc;
    return la_torgb_forward(g_sum, b0.frgb_wt, nullptr, 0, nullptr, accumulate ? g_img : nullptr, nullptr, g_img, B, b0.cin, h->imgc, h->R,
                            h->R, -1.f, stream);
}
