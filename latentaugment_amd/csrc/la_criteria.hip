// Diversity criteria of the latent-optimisation loop: pairwise squared-L2 against real-data banks.
// Restates l2_loss_vectorized (augments/utils/util_latent_aug.py:315-361) and its callers calc_loss_latent (:427-433)
// and calc_loss_pix (:373-385) in GEMM form:  D[m][n] = |Y_m|^2 + |X_n|^2 - 2 <Y_m, X_n>.
// Every kernel here is a single streaming pass over a bank: HBM-bound.
#include "la_criteria.h"

#define NCH 8   // query rows handled per pass over the bank

// yx[m][n] = <Y_m, X_n>,  yy[m] = |Y_m|^2.   One workgroup per bank row; X element (n,k) = X[n*ldx + k % xmod].
__global__ __launch_bounds__(256) void la_bank_dot_kernel(const float* __restrict__ Y, long K, const float* __restrict__ X,
                                                         int n, long ldx, long xmod, float* __restrict__ yx,
                                                         float* __restrict__ yy) {
    __shared__ float red[4];
    const long m = blockIdx.x;
    const float* yrow = Y + m * K;
    for (int n0 = 0; n0 < n; n0 += NCH) {
        float acc[NCH], sq = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) acc[q] = 0.f;
        for (long k = threadIdx.x; k < K; k += blockDim.x) {
            const float yv = yrow[k];
            sq += yv * yv;
            const long kx = (xmod > 0) ? k % xmod : k;
#pragma unroll
            for (int q = 0; q < NCH; ++q)
                if (n0 + q < n) acc[q] += yv * X[(long)(n0 + q) * ldx + kx];
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const float t = la_block_sum_256(acc[q], red);
            if (threadIdx.x == 0 && n0 + q < n) yx[m * n + n0 + q] = t;
        }
        if (n0 == 0 && yy) {
            const float t = la_block_sum_256(sq, red);
            if (threadIdx.x == 0) yy[m] = t;
        }
    }
}

// xx[n] = |X_n|^2 (K elements, with the same modulo addressing)
__global__ __launch_bounds__(256) void la_rows_sqnorm_kernel(const float* __restrict__ X, long K, long ldx, long xmod,
                                                            float* __restrict__ xx) {
    __shared__ float red[4];
    const long n = blockIdx.x;
    float sq = 0.f;
    for (long k = threadIdx.x; k < K; k += blockDim.x) {
        const float v = X[n * ldx + ((xmod > 0) ? k % xmod : k)];
        sq += v * v;
    }
    const float t = la_block_sum_256(sq, red);
    if (threadIdx.x == 0) xx[n] = t;
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
    float a0 = 0.f, a1 = 0.f;
    if (k < K) {
        long r = w;
        for (; r + 4 < m; r += 8) { a0 += Y[r * K + k]; a1 += Y[(r + 4) * K + k]; }
        if (r < m) a0 += Y[r * K + k];
    }
    comb[w][kl] = a0 + a1;
    __syncthreads();
    if (w == 0 && k < K) colsum[k] = (comb[0][kl] + comb[1][kl]) + (comb[2][kl] + comb[3][kl]);
}

int la_bank_dot(const float* Y, long m, long K, const float* X, int n, long ldx, long xmod, float* yx, float* yy,
                hipStream_t stream) {
    LA_CHECK_ARG(Y && X && yx && m >= 1 && K >= 1 && n >= 1, "bank_dot: bad arguments");
    hipLaunchKernelGGL(la_bank_dot_kernel, dim3((unsigned)m), dim3(256), 0, stream, Y, K, X, n, ldx, xmod, yx, yy);
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
    hipLaunchKernelGGL(la_rows_sqnorm_kernel, dim3(n), dim3(256), 0, stream, X, K, ldx, xmod, xx_ws);
    hipLaunchKernelGGL(la_l2_finish_kernel, dim3(la_cdiv(m * n, 256)), dim3(256), 0, stream, yx_ws, yy_ws, xx_ws, m, n);
    hipLaunchKernelGGL(la_sum_scale_kernel, dim3(1), dim3(256), 0, stream, yx_ws, m * n, scale, out, accumulate);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// Public op: the reference's l2_loss_vectorized on flattened rows.  D [m][n] always written; mean optional.
extern "C" int la_pairwise_l2_f32(const float* X, int n, const float* Y, long m, long K, float* D, float* mean_out,
                                  float* workspace /* m + n floats */, hipStream_t stream) {
    LA_CHECK_ARG(X && Y && D && workspace, "pairwise_l2: null pointer");
    LA_CHECK_ARG(n >= 1 && m >= 1 && K >= 1, "pairwise_l2: empty input");
    float* yy = workspace;
    float* xx = workspace + m;
    int rc = la_bank_dot(Y, m, K, X, n, K, 0, D, yy, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(la_rows_sqnorm_kernel, dim3(n), dim3(256), 0, stream, X, K, K, (long)0, xx);
    hipLaunchKernelGGL(la_l2_finish_kernel, dim3(la_cdiv(m * n, 256)), dim3(256), 0, stream, D, yy, xx, m, n);
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
    for (int l = 0; l < num_ws; ++l) {
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
