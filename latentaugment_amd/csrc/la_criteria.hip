// Diversity criteria of the latent-optimisation loop: pairwise squared-L2 against real-data banks.
// Restates l2_loss_vectorized (augments/utils/util_latent_aug.py:315-361) and its callers calc_loss_latent (:427-433)
// and calc_loss_pix (:373-385) in GEMM form:  D[m][n] = |Y_m|^2 + |X_n|^2 - 2 <Y_m, X_n>.
// Every kernel here is a single streaming pass over a bank: HBM-bound.
#include "la_criteria.h"

#define NCH 8   // query rows handled per pass over the bank

// yx[m][n] = <Y_m, X_n>,  yy[m] = |Y_m|^2  -- skinny GEMM, HBM-bound on the bank.
// Block = RB bank rows x one K slice; the NCH query rows are re-used across the RB bank rows from registers (one X load
// feeds RB FMAs), each thread keeps RB x NCH accumulators, K slices are summed by la_bank_dot_finish_kernel in a fixed
// order (deterministic).  X element (n, k) = X[n*ldx + k % xmod].  VEC = 4 needs K % 4 == 0 and 16-byte aligned rows.
#define RB 8
#define KSPLIT 16     // workspace slices; a launch uses 4 (short rows) or all 16 (rows > 16384 elements, e.g. the pixel bank)
template <int VEC>
__global__ __launch_bounds__(256) void la_bank_dot_kernel(const float* __restrict__ Y, long m, long K, const float* __restrict__ X,
                                                         int n, int n0, long ldx, long xmod, float* __restrict__ part_yx,
                                                         float* __restrict__ part_yy, int ksplit) {
    const long r0 = (long)blockIdx.x * RB;
    const int ks = blockIdx.y;
    const long kper = ((K + ksplit - 1) / ksplit + VEC * 256 - 1) / (VEC * 256) * (VEC * 256);
    const long kbeg = ks * kper, kend = (kbeg + kper < K) ? kbeg + kper : K;
    float acc[RB][NCH], sq[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) { sq[r] = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) acc[r][q] = 0.f; }
    for (long k = kbeg + (long)threadIdx.x * VEC; k < kend; k += 256 * VEC) {
        float xv[NCH][VEC];
        const long kx = (xmod > 0) ? k % xmod : k;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            if (n0 + q < n) {
                if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(X + (long)(n0 + q) * ldx + kx); xv[q][0] = t.x; xv[q][1] = t.y; xv[q][2] = t.z; xv[q][3] = t.w; }
                else xv[q][0] = X[(long)(n0 + q) * ldx + kx];
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) xv[q][e] = 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (r0 + r >= m) break;
            float yv[VEC];
            if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(Y + (r0 + r) * K + k); yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w; }
            else yv[0] = Y[(r0 + r) * K + k];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                sq[r] += yv[e] * yv[e];
#pragma unroll
                for (int q = 0; q < NCH; ++q) acc[r][q] += yv[e] * xv[q][e];
            }
        }
    }
    // block sums of the RB*NCH + RB accumulators: shuffle tree inside each wave, then one pass through LDS (fixed order)
    __shared__ float wsum[4][RB * NCH + RB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            float v = acc[r][q];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) wsum[wv][r * NCH + q] = v;
        }
        float v = sq[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) wsum[wv][RB * NCH + r] = v;
    }
    __syncthreads();
    if (threadIdx.x < RB * NCH + RB) {
        const int i = threadIdx.x;
        const float t = (wsum[0][i] + wsum[1][i]) + (wsum[2][i] + wsum[3][i]);
        if (i < RB * NCH) {
            const int r = i / NCH, q = i - r * NCH;
            if (r0 + r < m && n0 + q < n) part_yx[((long)ks * m + r0 + r) * n + n0 + q] = t;
        } else {
            const int r = i - RB * NCH;
            if (r0 + r < m && n0 == 0 && part_yy) part_yy[(long)ks * m + r0 + r] = t;
        }
    }
}

__global__ void la_bank_dot_finish_kernel(const float* __restrict__ part_yx, const float* __restrict__ part_yy,
                                          float* __restrict__ yx, float* __restrict__ yy, long m, int n, int ksplit) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m * n) {
        float v = 0.f;
        for (int k = 0; k < ksplit; ++k) v += part_yx[(long)k * m * n + i];
        yx[i] = v;
    }
    if (yy && i < m) {
        float v = 0.f;
        for (int k = 0; k < ksplit; ++k) v += part_yy[(long)k * m + i];
        yy[i] = v;
    }
}

// xx[n] = |X_n|^2 (K elements, with the same modulo addressing); 32 slices per row, then a fixed-order sum
#define SQ_SLICES 32
__global__ __launch_bounds__(256) void la_rows_sqnorm_kernel(const float* __restrict__ X, long K, long ldx, long xmod,
                                                            float* __restrict__ part) {
    __shared__ float red[4];
    const long n = blockIdx.x;
    const long per = (K + SQ_SLICES - 1) / SQ_SLICES;
    const long k0 = blockIdx.y * per, k1 = (k0 + per < K) ? k0 + per : K;
    float sq = 0.f;
    for (long k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
        const float v = X[n * ldx + ((xmod > 0) ? k % xmod : k)];
        sq += v * v;
    }
    const float t = la_block_sum_256(sq, red);
    if (threadIdx.x == 0) part[n * SQ_SLICES + blockIdx.y] = t;
}
__global__ void la_rows_sqnorm_finish_kernel(const float* __restrict__ part, float* __restrict__ xx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int k = 0; k < SQ_SLICES; ++k) v += part[(long)i * SQ_SLICES + k];
    xx[i] = v;
}

// xx_ws: n * (SQ_SLICES + 1) floats; result in xx_ws[0 .. n)
static int rows_sqnorm(const float* X, long K, long ldx, long xmod, float* xx_ws, int n, hipStream_t stream) {
    hipLaunchKernelGGL(la_rows_sqnorm_kernel, dim3(n, SQ_SLICES), dim3(256), 0, stream, X, K, ldx, xmod, xx_ws + n);
    hipLaunchKernelGGL(la_rows_sqnorm_finish_kernel, dim3(la_cdiv(n, 256)), dim3(256), 0, stream, xx_ws + n, xx_ws, n);
    return LA_OK;
}

// D[m][n] = yy[m] + xx[n] - 2 yx[m][n]   (in place over yx)
__global__ void la_l2_finish_kernel(float* __restrict__ yx, const float* __restrict__ yy, const float* __restrict__ xx,
                                    long m, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * n) return;
    const long mi = i / n;
    const int ni = (int)(i - mi * n);
    yx[i] = (yy[mi] + xx[ni]) - 2.f * yx[i];
}

// out[slot] (+)= scale * sum(D)          single workgroup, deterministic
__global__ __launch_bounds__(256) void la_sum_scale_kernel(const float* __restrict__ D, long count, float scale,
                                                          float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = threadIdx.x; i < count; i += blockDim.x) acc += D[i];
    const float t = la_block_sum_256(acc, red);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + t * scale;
}

// colsum[k] = sum_m Y[m][k].  Block = 64 columns x 4 row groups (wave w takes rows w, w+4, ...), combined through LDS.
__global__ __launch_bounds__(256) void la_bank_colsum_kernel(const float* __restrict__ Y, long m, long K,
                                                            float* __restrict__ colsum) {
    __shared__ float comb[4][64];
    const int kl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long k = (long)blockIdx.x * 64 + kl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four rows in flight per thread (the scan is a plain HBM stream)
    if (k < K) {
        long r = w;
        for (; r + 12 < m; r += 16) {
            const float v0 = Y[r * K + k], v1 = Y[(r + 4) * K + k], v2 = Y[(r + 8) * K + k], v3 = Y[(r + 12) * K + k];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; r < m; r += 4) a0 += Y[r * K + k];
    }
    comb[w][kl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && k < K) colsum[k] = (comb[0][kl] + comb[1][kl]) + (comb[2][kl] + comb[3][kl]);
}

// yx: (KSPLIT + 1) * m * n floats (result in the first m*n); yy (may be NULL): (KSPLIT + 1) * m floats
int la_bank_dot(const float* Y, long m, long K, const float* X, int n, long ldx, long xmod, float* yx, float* yy,
                hipStream_t stream) {
    LA_CHECK_ARG(Y && X && yx && m >= 1 && K >= 1 && n >= 1, "bank_dot: bad arguments");
    float* pyx = yx + m * n;
    float* pyy = yy ? yy + m : nullptr;
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && (xmod % 4 == 0) && (((size_t)Y | (size_t)X) & 15) == 0;
    // long rows with few bank rows (pixel bank: 256 x 32761) need more K slices to cover the chip
    const int ksplit = (K > 16384 && la_cdiv(m, RB) * 4 < 512) ? KSPLIT : 4;
    dim3 grid(la_cdiv(m, RB), ksplit);
    for (int n0 = 0; n0 < n; n0 += NCH) {
        if (vec) hipLaunchKernelGGL(la_bank_dot_kernel<4>, grid, dim3(256), 0, stream, Y, m, K, X, n, n0, ldx, xmod, pyx, pyy, ksplit);
        else hipLaunchKernelGGL(la_bank_dot_kernel<1>, grid, dim3(256), 0, stream, Y, m, K, X, n, n0, ldx, xmod, pyx, pyy, ksplit);
    }
    hipLaunchKernelGGL(la_bank_dot_finish_kernel, dim3(la_cdiv(m * n > m ? m * n : m, 256)), dim3(256), 0, stream, pyx, pyy, yx, yy, m, n, ksplit);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

int la_bank_colsum(const float* Y, long m, long K, float* colsum, hipStream_t stream) {
    LA_CHECK_ARG(Y && colsum && m >= 1 && K >= 1, "bank_colsum: bad arguments");
    hipLaunchKernelGGL(la_bank_colsum_kernel, dim3(la_cdiv(K, 64)), dim3(256), 0, stream, Y, m, K, colsum);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

int la_l2_mean_from_bank(const float* Y, long m, long K, const float* X, int n, long ldx, long xmod, float* yx_ws,
                         float* yy_ws, float* xx_ws, float scale, float* out, int accumulate, hipStream_t stream) {
    int rc = la_bank_dot(Y, m, K, X, n, ldx, xmod, yx_ws, yy_ws, stream);
    if (rc) return rc;
    rows_sqnorm(X, K, ldx, xmod, xx_ws, n, stream);
    hipLaunchKernelGGL(la_l2_finish_kernel, dim3(la_cdiv(m * n, 256)), dim3(256), 0, stream, yx_ws, yy_ws, xx_ws, m, n);
    hipLaunchKernelGGL(la_sum_scale_kernel, dim3(1), dim3(256), 0, stream, yx_ws, m * n, scale, out, accumulate);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// Public op: the reference's l2_loss_vectorized on flattened rows.  D [m][n] always written; mean optional.
extern "C" long la_pairwise_l2_workspace_floats(int n, long m) {
    return (KSPLIT + 1) * m * n + (KSPLIT + 1) * m + (long)(SQ_SLICES + 1) * n;
}

extern "C" int la_pairwise_l2_f32(const float* X, int n, const float* Y, long m, long K, float* D, float* mean_out,
                                  float* workspace /* la_pairwise_l2_workspace_floats(n, m) */, hipStream_t stream) {
    LA_CHECK_ARG(X && Y && D && workspace, "pairwise_l2: null pointer");
    LA_CHECK_ARG(n >= 1 && m >= 1 && K >= 1, "pairwise_l2: empty input");
    // workspace: yx (KSPLIT+1)*m*n | yy (KSPLIT+1)*m | xx (SQ_SLICES+1)*n
    float* yx = workspace;
    float* yy = yx + (KSPLIT + 1) * m * n;
    float* xx = yy + (KSPLIT + 1) * m;
    int rc = la_bank_dot(Y, m, K, X, n, K, 0, yx, yy, stream);
    if (rc) return rc;
    rows_sqnorm(X, K, K, 0, xx, n, stream);
    hipLaunchKernelGGL(la_l2_finish_kernel, dim3(la_cdiv(m * n, 256)), dim3(256), 0, stream, yx, yy, xx, m, n);
    LA_HIP(hipMemcpyAsync(D, yx, sizeof(float) * m * n, hipMemcpyDeviceToDevice, stream));
    if (mean_out)
        hipLaunchKernelGGL(la_sum_scale_kernel, dim3(1), dim3(256), 0, stream, D, m * n,
                           1.0f / ((float)m * (float)n) / (float)K, mean_out, 0);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// centre crop of planes: src [P][R][R] -> dst [P][cc][cc] (torchvision CenterCrop offset: round((R-cc)/2))
__global__ void la_center_crop_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int cc, int off,
                                      long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long p = i / ((long)cc * cc);
    const int r = (int)(i - p * cc * cc);
    const int y = r / cc, x = r - y * cc;
    dst[i] = src[(p * R + y + off) * R + x + off];
}

extern "C" int la_center_crop_f32(const float* src, float* dst, long planes, int R, int cc, int off,
                                  hipStream_t stream) {
    LA_CHECK_ARG(src && dst && planes >= 1 && cc >= 1 && off >= 0 && off + cc <= R, "center_crop: bad arguments");
    const long total = planes * cc * cc;
    hipLaunchKernelGGL(la_center_crop_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, stream, src, dst, R, cc, off, total);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// pixel-criterion gradient scattered into the image gradient:
//   inside the crop:  g_img[b][c][y][x] = sign * 2*coef * (m * img - colsum_c[(y-off)*cc + (x-off)]) ;  0 elsewhere
// colsum layout [imgc][cc*cc].  coef = w_pix / (n_modes * m * n * cc*cc).
__global__ void la_pix_grad_kernel(const float* __restrict__ img, const float* __restrict__ colsum, float* __restrict__ g,
                                   int imgc, int R, int cc, int off, float coef2, float mrows, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % R);
    const int y = (int)((i / R) % R);
    const int c = (int)((i / ((long)R * R)) % imgc);
    float v = 0.f;
    if (x >= off && x < off + cc && y >= off && y < off + cc)
        v = coef2 * (mrows * img[i] - colsum[((long)c * cc + (y - off)) * cc + (x - off)]);
    g[i] = v;
}

int la_pix_grad(const float* img, const float* colsum, float* g, int B, int imgc, int R, int cc, int off, float coef2,
                float mrows, hipStream_t stream) {
    const long total = (long)B * imgc * R * R;
    hipLaunchKernelGGL(la_pix_grad_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, stream, img, colsum, g, imgc, R, cc,
                       off, coef2, mrows, total);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// latent-space gradient assembly for W-space optimisation (ws = w repeated num_ws times, util_latent_aug.py:493-494):
//   dw[b][j] = sum_l dws[b][l][j]  +  lat2 * (num_ws * m * w[b][j] - sum_l colsumW[l][j])
// lat2 = sign * 2 * w_latent / (m * n * num_ws * wdim); colsumW may be null (no latent criterion).
__global__ void la_latent_combine_kernel(const float* __restrict__ dws, const float* __restrict__ w,
                                         const float* __restrict__ colsumW, float* __restrict__ dw, int num_ws, int wdim,
                                         float lat2, float mrows, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
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
    if (colsumW) acc += lat2 * ((float)num_ws * mrows * w[i] - cs);
    dw[i] = acc;
}

int la_latent_combine(const float* dws, const float* w, const float* colsumW, float* dw, int B, int num_ws, int wdim,
                      float lat2, float mrows, hipStream_t stream) {
    const long total = (long)B * wdim;
    hipLaunchKernelGGL(la_latent_combine_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, stream, dws, w, colsumW, dw,
                       num_ws, wdim, lat2, mrows, total);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// w_aug[b][l][j] = alpha * w_opt[b][j] + (1 - alpha) * w0[b][j]   (hard_aug: alpha = 1)   util_latent_aug.py:438-454
__global__ void la_broadcast_mix_kernel(const float* __restrict__ w_opt, const float* __restrict__ w0,
                                        float* __restrict__ w_aug, int num_ws, int wdim, float alpha, int soft,
                                        long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % wdim);
    const long b = i / ((long)wdim * num_ws);
    const float a = w_opt[b * wdim + j];
    w_aug[i] = soft ? (alpha * a) + ((1.f - alpha) * w0[b * wdim + j]) : a;
}

int la_broadcast_mix(const float* w_opt, const float* w0, float* w_aug, int B, int num_ws, int wdim, float alpha,
                     int soft, hipStream_t stream) {
    const long total = (long)B * num_ws * wdim;
    hipLaunchKernelGGL(la_broadcast_mix_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, stream, w_opt, w0, w_aug, num_ws,
                       wdim, alpha, soft, total);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
