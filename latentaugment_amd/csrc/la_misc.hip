// Error reporting + the standalone elementwise ops of the C ABI (bias_act, Adam, small vector helpers).
#include "la_common.h"

#include <string.h>

static thread_local char g_err[256] = "";
void la_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* la_last_error(void) { return g_err; }
extern "C" int la_abi_version(void) { return 1; }

// ------------------------------------------------------------------------------------------------------------
// bias_act forward:  y = clamp(act(x + b[(i / stepb) % nb]) * gain)            (bias_act.cu:23-147, grad = 0)
// bias_act backward: dx = dy * act'(y-referenced) ; db[c] = sum dx              (bias_act.cu grad = 1; bias_act.py:155-177)
__global__ __launch_bounds__(256) void la_bias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                             float* __restrict__ y, long n, long stepb, int nb, int act,
                                                             float alpha, float gain, float clamp) {
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n && (stepb % 4 == 0 || !b)) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            const float bv = b ? b[(i / stepb) % nb] : 0.f;
            v.x = la_act_fwd(v.x + bv, act, alpha, gain, clamp); v.y = la_act_fwd(v.y + bv, act, alpha, gain, clamp);
            v.z = la_act_fwd(v.z + bv, act, alpha, gain, clamp); v.w = la_act_fwd(v.w + bv, act, alpha, gain, clamp);
            *reinterpret_cast<float4*>(y + i) = v;
        } else {
            for (long k = i; k < n && k < i + 4; ++k) {
                const float bv = b ? b[(k / stepb) % nb] : 0.f;
                y[k] = la_act_fwd(x[k] + bv, act, alpha, gain, clamp);
            }
        }
    }
}

__global__ __launch_bounds__(256) void la_bias_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yref,
                                                             float* __restrict__ dx, long n, int act, float alpha,
                                                             float gain, float clamp) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dx[i] = dy[i] * la_act_bwd_from_y(yref[i], act, alpha, gain, clamp);
}

// db[c] = sum over everything but dim: element i belongs to channel (i / stepb) % nb.  One block per channel.
__global__ __launch_bounds__(256) void la_bias_grad_kernel(const float* __restrict__ dx, float* __restrict__ db, long n,
                                                          long stepb, int nb) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const long outer = n / (stepb * nb);
    float acc = 0.f;
    for (long o = 0; o < outer; ++o) {
        const float* p = dx + (o * nb + c) * stepb;
        for (long k = threadIdx.x; k < stepb; k += blockDim.x) acc += p[k];
    }
    const float t = la_block_sum_256(acc, red);
    if (threadIdx.x == 0) db[c] = t;
}

static int act_ok(int act) { return act == LA_ACT_LINEAR || act == LA_ACT_RELU || act == LA_ACT_LRELU; }

extern "C" int la_bias_act_f32(const float* x, const float* b, float* y, long n, long stepb, int nb, int act, float alpha,
                               float gain, float clamp, hipStream_t stream) {
    if (n == 0) return LA_OK;   // empty tensors are legal (and have null data pointers)
    LA_CHECK_ARG(x && y && n > 0, "bias_act: null pointer");
    LA_CHECK_ARG(act_ok(act), "bias_act: only linear(1)/relu(2)/lrelu(3) are implemented on this path");
    LA_CHECK_ARG(!b || (stepb >= 1 && nb >= 1 && n % (stepb * nb) == 0), "bias_act: bias does not tile the tensor");
    if (n == 0) return LA_OK;
    if (!b) { stepb = 4; nb = 1; }
    long blocks = la_cdiv(la_cdiv(n, 4), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(la_bias_act_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, b, y, n, stepb, nb, act,
                       alpha, gain, clamp);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

extern "C" int la_bias_act_grad_f32(const float* dy, const float* yref, float* dx, float* db, long n, long stepb, int nb,
                                    int act, float alpha, float gain, float clamp, hipStream_t stream) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(dy && yref && dx && n > 0, "bias_act_grad: null pointer");
    LA_CHECK_ARG(act_ok(act), "bias_act_grad: only linear(1)/relu(2)/lrelu(3) are implemented on this path");
    if (n == 0) return LA_OK;
    long blocks = la_cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(la_bias_act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dy, yref, dx, n, act, alpha,
                       gain, clamp);
    LA_CHECK_LAUNCH();
    if (db) {
        LA_CHECK_ARG(stepb >= 1 && nb >= 1 && n % (stepb * nb) == 0, "bias_act_grad: bias does not tile the tensor");
        hipLaunchKernelGGL(la_bias_grad_kernel, dim3(nb), dim3(256), 0, stream, dx, db, n, stepb, nb);
        LA_CHECK_LAUNCH();
    }
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// The general bias_act op: the plugin entry point  bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)  of the reference
// (bias_act.cpp:32; activation table bias_act.py:20-30: linear, relu, lrelu, tanh, sigmoid, elu, selu, softplus, swish = ids 1..9) with
// all three values of `grad` -- the SG2 path above only ever needs linear / relu / lrelu with grad <= 1 and has its own fused forms.
//   grad 0:  out = clamp(f(x + b) * gain)                         (dy, if given, multiplies before the clamp, as the plugin does)
//   grad 1:  out = x * f'  * gain * dy,  zero where |yref| >= clamp      (x = the incoming gradient, f' from yref / gain or xref + b)
//   grad 2:  out = x * f'' * gain * dy,  zero where |yref| >= clamp      (x = the gradient of the gradient)
// Each activation is one small function object: value, first and second derivative, the derivatives written in what the reference
// saves for it -- the OUTPUT (bias_act.py `ref='y'`) or, for swish, the INPUT (`ref='x'`).
#define LA_SELU_SCALE 1.0507009873554804934193349852946f
#define LA_SELU_ALPHA 1.6732632423543772848170429916717f
struct LaActV { float f, d1, d2; };
// value at v (grad 0), or derivatives from the saved reference r (grad >= 1: r = yref / gain, or xref + b for swish)
__device__ __forceinline__ float la_actfull_value(int act, float v, float alpha) {
    switch (act) {
        case 2: return v > 0.f ? v : 0.f;
        case 3: return v > 0.f ? v : v * alpha;
        case 4: return tanhf(v);
        case 5: return 1.f / (1.f + expf(-v));
        case 6: return v >= 0.f ? v : expm1f(v);
        case 7: return v >= 0.f ? LA_SELU_SCALE * v : (LA_SELU_SCALE * LA_SELU_ALPHA) * expm1f(v);
        case 8: return v > 80.f ? v : log1pf(expf(v));
        case 9: return v / (1.f + expf(-v));
        default: return v;
    }
}
__device__ __forceinline__ void la_actfull_derivs(int act, float r, float alpha, float& d1, float& d2) {
    d2 = 0.f;
    switch (act) {
        case 2: d1 = r > 0.f ? 1.f : 0.f; break;
        case 3: d1 = r > 0.f ? 1.f : alpha; break;
        case 4: d1 = 1.f - r * r; d2 = d1 * (-2.f * r); break;
        case 5: d1 = r * (1.f - r); d2 = d1 * (1.f - 2.f * r); break;
        case 6: d1 = r >= 0.f ? 1.f : r + 1.f; d2 = r >= 0.f ? 0.f : r + 1.f; break;
        case 7: d1 = r >= 0.f ? LA_SELU_SCALE : r + LA_SELU_SCALE * LA_SELU_ALPHA; d2 = r >= 0.f ? 0.f : r + LA_SELU_SCALE * LA_SELU_ALPHA; break;
        case 8: { const float e = expf(-r); d1 = 1.f - e; d2 = e * (1.f - e); break; }
        case 9: {      // r = the pre-activation: sigma = 1 / (1 + e^-r);  f' = sigma (1 + r (1 - sigma));  f'' = sigma (1 - sigma) (2 + r (1 - 2 sigma))
            const float sg = 1.f / (1.f + expf(-r));
            d1 = sg * (1.f + r * (1.f - sg));
            d2 = sg * (1.f - sg) * (2.f + r * (1.f - 2.f * sg));
            break;
        }
        default: d1 = 1.f; break;
    }
}
__global__ __launch_bounds__(256) void la_bias_act_full_kernel(const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ xref,
                                                              const float* __restrict__ yref, const float* __restrict__ dy, float* __restrict__ out, long n,
                                                              long stepb, int nb, int grad, int act, float alpha, float gain, float clamp) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float bv = b ? b[(i / stepb) % nb] : 0.f;
        const float g = dy ? dy[i] : 1.f;
        float v;
        if (grad == 0) {
            v = la_actfull_value(act, x[i] + bv, alpha) * gain * g;
            if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
        } else {
            float yr = yref ? yref[i] : 0.f;
            float r = gain != 0.f ? yr / gain : 0.f;
            if (act == 9) {      // swish keeps its input: the reference value and the clamp test are re-derived from it
                r = (xref ? xref[i] : 0.f) + bv;
                yr = la_actfull_value(9, r, alpha) * gain;
            }
            float d1, d2;
            la_actfull_derivs(act, r, alpha, d1, d2);
            v = x[i] * (grad == 1 ? d1 : d2) * gain * g;
            if (clamp >= 0.f && !(yr > -clamp && yr < clamp)) v = 0.f;
        }
        out[i] = v;
    }
}

extern "C" int la_bias_act_ex_f32(const float* x, const float* b, const float* xref, const float* yref, const float* dy, float* out, long n,
                                  long stepb, int nb, int grad, int act, float alpha, float gain, float clamp, hipStream_t stream) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && out && n > 0, "bias_act_ex: null pointer");
    LA_CHECK_ARG(act >= 1 && act <= 9, "bias_act_ex: activation id must be 1..9 (bias_act.py:20-30)");
    LA_CHECK_ARG(grad >= 0 && grad <= 2, "bias_act_ex: grad must be 0, 1 or 2");
    LA_CHECK_ARG(grad == 0 || act == 9 || act == 1 || yref, "bias_act_ex: grad >= 1 needs the saved output yref");
    LA_CHECK_ARG(grad == 0 || act != 9 || xref, "bias_act_ex: swish derives its gradients from the saved input xref");
    LA_CHECK_ARG(!b || (stepb >= 1 && nb >= 1 && n % (stepb * nb) == 0), "bias_act_ex: bias does not tile the tensor");
    if (!b) { stepb = 1; nb = 1; }
    long blocks = la_cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(la_bias_act_full_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, b, xref, yref, dy, out, n, stepb, nb, grad, act,
                       alpha, gain, clamp);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// db[c] = sum of dx over every axis but the bias axis (bias_act.py:187, :206): element i belongs to channel (i / stepb) % nb
extern "C" int la_bias_sum_f32(const float* dx, float* db, long n, long stepb, int nb, hipStream_t stream) {
    LA_CHECK_ARG(dx && db && n > 0 && stepb >= 1 && nb >= 1 && n % (stepb * nb) == 0, "bias_sum: bad arguments");
    hipLaunchKernelGGL(la_bias_grad_kernel, dim3(nb), dim3(256), 0, stream, dx, db, n, stepb, nb);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics as used at util_latent_aug.py:213,276): one fused elementwise update.
__global__ void la_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                               float bc2_sqrt, float gscale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gv = g[i] * gscale;
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv; v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mv / denom);
}

// Loop-engine variant: the bias corrections of step t = *ctr + 1 come from a device table {1 - b1^t, sqrt(1 - b2^t)} (filled by
// the host with the same powf as la_adam_step_f32), so that one captured launch serves every step; la_step_advance bumps the
// counter at the end of a step.
__global__ void la_adam_tab_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                   long n, float lr, float b1, float b2, float eps, const float2* __restrict__ tab,
                                   const int* __restrict__ ctr) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 bc = tab[*ctr];
    const float gv = g[i];
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv; v[i] = vv;
    const float denom = sqrtf(vv) / bc.y + eps;
    p[i] = p[i] - (lr / bc.x) * (mv / denom);
}
__global__ void la_step_advance_kernel(int* ctr) { if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += 1; }

int la_adam_step_tab(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                     const float* tab, const int* ctr, hipStream_t stream) {
    if (n == 0) return LA_OK;
    hipLaunchKernelGGL(la_adam_tab_kernel, dim3(la_cdiv(n, 256)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       reinterpret_cast<const float2*>(tab), ctr);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
int la_step_advance(int* ctr, hipStream_t stream) {
    hipLaunchKernelGGL(la_step_advance_kernel, dim3(1), dim3(64), 0, stream, ctr);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
// Tail of an optimisation step in ONE launch (was three: la_latent_combine, la_adam_tab_kernel, la_step_advance -- each a 5 us link of
// the step's serial chain): dw = sum over the ws slots of dws + the latent criterion's gradient (the arithmetic and its order are
// la_latent_combine_kernel's), the Adam update with the bias corrections of step *ctr + 1 (la_adam_tab_kernel's), and the step
// counter: every workgroup draws a ticket AFTER its threads have read *ctr; the one that draws the last ticket resets it and bumps
// the counter -- nobody is left to read the old value.
__global__ __launch_bounds__(256) void la_step_tail_kernel(const float* __restrict__ dws, const float* __restrict__ colsumW, float* __restrict__ dw,
                                                          float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, int num_ws,
                                                          int wdim, float lat2, float mrows, long total, float lr, float b1, float b2,
                                                          float eps, const float2* __restrict__ tab, int* __restrict__ ctr,
                                                          int* __restrict__ ticket) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const float2 bc = tab[*ctr];
    if (i < total) {
        const long b = i / wdim;
        const int j = (int)(i - b * wdim);
        float acc = 0.f, cs = 0.f;
        int l = 0;
        for (; l + 7 < num_ws; l += 8) {      // (eight slots' loads in flight, added in slot order)
            float a[8], c[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] = dws ? dws[(b * num_ws + l + k) * wdim + j] : 0.f;
                c[k] = colsumW ? colsumW[(long)(l + k) * wdim + j] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { if (dws) acc += a[k]; if (colsumW) cs += c[k]; }
        }
        for (; l < num_ws; ++l) {
            if (dws) acc += dws[(b * num_ws + l) * wdim + j];
            if (colsumW) cs += colsumW[(long)l * wdim + j];
        }
        const float pv = p[i];
        if (colsumW) acc += lat2 * ((float)num_ws * mrows * pv - cs);
        dw[i] = acc;
        const float mv = b1 * m[i] + (1.f - b1) * acc;
        const float vv = b2 * v[i] + (1.f - b2) * acc * acc;
        m[i] = mv; v[i] = vv;
        const float denom = sqrtf(vv) / bc.y + eps;
        p[i] = pv - (lr / bc.x) * (mv / denom);
    }
    __syncthreads();      // every thread of the workgroup has read *ctr
    if (threadIdx.x == 0) {
        if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) { *ticket = 0; *ctr += 1; }
    }
}

int la_step_tail(const float* dws, const float* colsumW, float* dw, float* p, float* m, float* v, int B, int num_ws, int wdim, float lat2,
                 float mrows, float lr, float beta1, float beta2, float eps, const float* tab, int* ctr, int* ticket, hipStream_t stream) {
    const long total = (long)B * wdim;
    if (total == 0) return LA_OK;
    hipLaunchKernelGGL(la_step_tail_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, stream, dws, colsumW, dw, p, m, v, num_ws, wdim, lat2,
                       mrows, total, lr, beta1, beta2, eps, reinterpret_cast<const float2*>(tab), ctr, ticket);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

void la_adam_fill_table(float* tab_host, int steps, float beta1, float beta2) {
    for (int t = 1; t <= steps; ++t) {
        tab_host[2 * (t - 1)] = 1.f - powf(beta1, (float)t);
        tab_host[2 * (t - 1) + 1] = sqrtf(1.f - powf(beta2, (float)t));
    }
}

extern "C" int la_adam_step_f32(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1,
                                float beta2, float eps, hipStream_t stream) {
    LA_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "adam: bad arguments");
    if (n == 0) return LA_OK;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(la_adam_kernel, dim3(la_cdiv(n, 256)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       bc1, bc2s, 1.f);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Counter-based unit normals for the explicit noise tensors of noise_mode='random' (the reference draws torch.randn per layer inside
// G.synthesis, util_latent_aug.py:308 / SURVEY 3.4 defect g; a draw cannot be bit-equal to another library's stream, it only has to
// be N(0, 1) and reproducible).  Element e of GLOBAL sample row r of layer l under `seed` is a pure function of (seed, l, r, e):
// Philox4x32-10 (Salmon et al., SC'11) with counter (e / 4, r, l, 0) and key (seed low, seed high) gives four 32-bit words, two
// Box-Muller pairs turn them into elements 4 (e / 4) .. 4 (e / 4) + 3.  A rank that holds rows [row0, row0 + rows) of a batch
// therefore generates exactly its rows -- nothing of the other ranks' -- and the gathered batch does not depend on the sharding.
__device__ __forceinline__ void la_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__global__ __launch_bounds__(256) void la_noise_normal_kernel(float* __restrict__ out, long rows, long row_elems, unsigned k0, unsigned k1,
                                                             unsigned layer, long row0) {
    const long q4 = (row_elems + 3) >> 2;                      // 4-element groups per row
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * q4) return;
    const long r = g / q4, q = g - r * q4;
    unsigned x[4];
    la_philox4x32_10((unsigned)q, (unsigned)(row0 + r), layer, (unsigned)((unsigned long long)q >> 32), k0, k1, x);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(x[2 * h] >> 8) + 0.5f) * (1.f / 16777216.f);      // (0, 1): 24 bits, exactly representable
        const float u2 = ((float)(x[2 * h + 1] >> 8) + 0.5f) * (1.f / 16777216.f);
        const float rad = sqrtf(-2.f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
    }
    float* o = out + r * row_elems + 4 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (4 * q + k < row_elems) o[k] = z[k];
}

extern "C" int la_noise_normal_f32(float* out, long rows, long row_elems, unsigned long long seed, unsigned layer, long row0, hipStream_t stream) {
    LA_CHECK_ARG((out || rows == 0) && rows >= 0 && row_elems >= 1 && row0 >= 0, "noise_normal: bad arguments");
    LA_CHECK_ARG(row0 + rows <= 0xffffffffl, "noise_normal: row index exceeds 32 bits");
    if (rows == 0) return LA_OK;
    const long n = rows * ((row_elems + 3) >> 2);
    hipLaunchKernelGGL(la_noise_normal_kernel, dim3((unsigned)la_cdiv(n, 256)), dim3(256), 0, stream, out, rows, row_elems, (unsigned)seed,
                       (unsigned)(seed >> 32), layer, row0);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
