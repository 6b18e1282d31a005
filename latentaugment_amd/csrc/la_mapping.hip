// Mapping network forward (rand_aug mode): G.mapping(z, c=None, truncation_psi) as called at
// augments/utils/util_latent_aug.py:203,460.  Public SG2 definition (names legacy.py:172-176):
//   x = z * rsqrt(mean(z^2) + 1e-8);  8 x { x = lrelu(x @ (W*lr_mul/sqrt(in))^T + b*lr_mul) * sqrt(2) };
//   ws = broadcast(x, num_ws);  psi != 1:  ws = w_avg + psi * (ws - w_avg).
// Latency-bound (8 dependent 512x512 GEMVs on a [B,512] activation); one wave per output feature.
#include "la_common.h"

#define MB 8

__global__ __launch_bounds__(256) void la_normalize_2nd_moment_kernel(const float* __restrict__ z, float* __restrict__ x,
                                                                     int dim) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float sq = 0.f;
    for (int j = threadIdx.x; j < dim; j += blockDim.x) { const float v = z[(long)b * dim + j]; sq += v * v; }
    const float t = la_block_sum_256(sq, red);
    const float r = rsqrtf(t / (float)dim + 1e-8f);
    for (int j = threadIdx.x; j < dim; j += blockDim.x) x[(long)b * dim + j] = z[(long)b * dim + j] * r;
}

// y[b][o] = act(dot(x[b], W[o]) * wgain + bias[o] * bgain) * gain       (FullyConnectedLayer + bias_act)
__global__ __launch_bounds__(256) void la_fc_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ y, int B, int in,
                                                   int out, float wgain, float bgain, int act, float alpha, float gain) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= out) return;
    const float* wrow = W + (long)o * in;
    const float bv = bias ? bias[o] * bgain : 0.f;
    const bool vec = (in & 3) == 0 && ((((size_t)W) | ((size_t)x)) & 15) == 0;      // rows of 16-byte aligned float4s
    for (int b0 = 0; b0 < B; b0 += MB) {
        float acc[MB];
#pragma unroll
        for (int q = 0; q < MB; ++q) acc[q] = 0.f;
        if (vec) {      // (the discriminator's 8192-wide epilogue row: 16-byte loads, two weight groups in flight)
            const float4* w4 = reinterpret_cast<const float4*>(wrow);
            const int n4 = in >> 2;
            for (int j = lane; j < n4; j += 128) {
                const bool two = j + 64 < n4;
                const float4 wa = w4[j], wb = two ? w4[j + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < MB; ++q)
                    if (b0 + q < B) {
                        const float4* x4 = reinterpret_cast<const float4*>(x + (long)(b0 + q) * in);
                        const float4 xa = x4[j], xb = two ? x4[j + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
                        acc[q] += (wa.x * xa.x + wa.y * xa.y) + (wa.z * xa.z + wa.w * xa.w) + (wb.x * xb.x + wb.y * xb.y) + (wb.z * xb.z + wb.w * xb.w);
                    }
            }
        } else
        for (int j = lane; j < in; j += 64) {
            const float wv = wrow[j];
#pragma unroll
            for (int q = 0; q < MB; ++q)
                if (b0 + q < B) acc[q] += wv * x[(long)(b0 + q) * in + j];
        }
#pragma unroll
        for (int q = 0; q < MB; ++q) {
            const float v = la_wave_sum(acc[q]);
            if (lane == 0 && b0 + q < B) y[(long)(b0 + q) * out + o] = la_act_fwd(v * wgain + bv, act, alpha, gain, -1.f);
        }
    }
}

__global__ void la_broadcast_truncate_kernel(const float* __restrict__ x, const float* __restrict__ w_avg,
                                             float* __restrict__ ws, int num_ws, int dim, float psi, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % dim);
    const long b = i / ((long)dim * num_ws);
    float v = x[b * dim + j];
    if (psi != 1.f && w_avg) v = w_avg[j] + psi * (v - w_avg[j]);     // torch.lerp(w_avg, x, psi)
    ws[i] = v;
}

// Wide rows (the discriminator's 8192 -> 512 epilogue layer: 16.8 MB of weights, B = 8): ONE output row per workgroup, its four waves
// each a quarter of the row (two weight float4 and 2 x MB input float4 in flight per lane and step), quarters combined through LDS in a
// fixed order.  With one wave per row (la_fc_kernel) 512 waves had 2 MB of loads in flight on the whole chip: 52 us = 0.3 TB/s.
__global__ __launch_bounds__(256) void la_fc_wide_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B, int in,
                                                        int out, float wgain, float bgain, int act, float alpha, float gain) {
    __shared__ float part[4][MB];
    const int o = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4* w4 = reinterpret_cast<const float4*>(W + (long)o * in);
    const int n4 = in >> 2, per = (n4 + 3) / 4;
    const int j0 = wave * per, j1 = j0 + per < n4 ? j0 + per : n4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b0 = 0; b0 < B; b0 += MB) {
        float acc[MB];
#pragma unroll
        for (int q = 0; q < MB; ++q) acc[q] = 0.f;
        for (int j = j0 + lane; j < j1; j += 128) {
            const bool two = j + 64 < j1;
            const float4 wa = w4[j], wb = two ? w4[j + 64] : z4;
#pragma unroll
            for (int q = 0; q < MB; ++q)
                if (b0 + q < B) {
                    const float4* x4 = reinterpret_cast<const float4*>(x + (long)(b0 + q) * in);
                    const float4 xa = x4[j], xb = two ? x4[j + 64] : z4;
                    acc[q] += (wa.x * xa.x + wa.y * xa.y) + (wa.z * xa.z + wa.w * xa.w) + (wb.x * xb.x + wb.y * xb.y) + (wb.z * xb.z + wb.w * xb.w);
                }
        }
#pragma unroll
        for (int q = 0; q < MB; ++q) {
            const float v = la_wave_sum(acc[q]);
            if (lane == 0) part[wave][q] = v;
        }
        __syncthreads();
        if (threadIdx.x < MB && b0 + (int)threadIdx.x < B) {
            const int q = threadIdx.x;
            const float v = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
            y[(long)(b0 + q) * out + o] = la_act_fwd(v * wgain + (bias ? bias[o] * bgain : 0.f), act, alpha, gain, -1.f);
        }
        __syncthreads();
    }
}

extern "C" int la_fc_f32(const float* x, const float* W, const float* bias, float* y, int B, int in, int out, float lr_mul,
                         int act, float alpha, float gain, hipStream_t stream) {
    LA_CHECK_ARG(x && W && y && B >= 1 && in >= 1 && out >= 1, "fc: bad arguments");
    if (in >= 2048 && (in & 3) == 0 && ((((size_t)W) | ((size_t)x)) & 15) == 0)
        hipLaunchKernelGGL(la_fc_wide_kernel, dim3(out), dim3(256), 0, stream, x, W, bias, y, B, in, out, lr_mul / sqrtf((float)in), lr_mul,
                           act, alpha, gain);
    else
    hipLaunchKernelGGL(la_fc_kernel, dim3(la_cdiv(out, 4)), dim3(256), 0, stream, x, W, bias, y, B, in, out,
                       lr_mul / sqrtf((float)in), lr_mul, act, alpha, gain);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// weights[i] [w_dim][in_i], biases[i] [w_dim]; tmp: 2 * B * max(z_dim, w_dim) floats; ws_out [B][num_ws][w_dim]
extern "C" int la_mapping_forward_f32(const float* z, int B, int z_dim, int w_dim, int num_layers,
                                      const float* const* weights, const float* const* biases, float lr_mul,
                                      const float* w_avg, float truncation_psi, int num_ws, float* tmp, float* ws_out,
                                      hipStream_t stream) {
    LA_CHECK_ARG(z && weights && biases && tmp && ws_out, "mapping: null pointer");
    LA_CHECK_ARG(B >= 1 && z_dim >= 1 && w_dim >= 1 && num_layers >= 0 && num_ws >= 1, "mapping: bad shape");
    const int mx = z_dim > w_dim ? z_dim : w_dim;
    float* a = tmp;
    float* b = tmp + (long)B * mx;
    hipLaunchKernelGGL(la_normalize_2nd_moment_kernel, dim3(B), dim3(256), 0, stream, z, a, z_dim);
    int in = z_dim;
    for (int l = 0; l < num_layers; ++l) {
        LA_CHECK_ARG(weights[l] && biases[l], "mapping: null layer tensor");
        int rc = la_fc_f32(a, weights[l], biases[l], b, B, in, w_dim, lr_mul, LA_ACT_LRELU, 0.2f, sqrtf(2.f), stream);
        if (rc) return rc;
        float* t = a; a = b; b = t;
        in = w_dim;
    }
    const long total = (long)B * num_ws * w_dim;
    hipLaunchKernelGGL(la_broadcast_truncate_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, stream, a, w_avg, ws_out, num_ws,
                       in, truncation_psi, total);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
