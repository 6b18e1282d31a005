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
