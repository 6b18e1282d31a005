// Quality-metric kernels: the numeric core of the reference's FID and Improved Precision/Recall,
// downstream of the detector features.
//   la_feature_moments_f64    metrics/metric_utils.py:104-118  raw_mean += sum_k x[k], raw_cov += x^T x, float64 accumulators
//   la_pr_kth_f16             metrics/precision_recall.py:75-79 k-th neighbour radius of every manifold point
//   la_pr_member_f16          metrics/precision_recall.py:80-84 is a probe inside any manifold point's radius
//   la_cdist_f16              metrics/precision_recall.py:19-32 the distance matrix itself (torch.cdist)
// Distances follow torch.cdist's GEMM form |a|^2 + |b|^2 - 2 a.b, clamped at 1e-30, square root.  The reference hands
// cdist float16 features; their products are exact on the fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate), so the dot
// products here are fp32 sums of exact terms.  The [rows, cols] matrix is never materialised for the radii / membership
// kernels: each wave keeps the k+1 smallest distances (or an OR) per row in registers while it streams over the columns.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "la_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PR_KMAX 8

// squared norms of fp16 rows, fp32
__global__ __launch_bounds__(256) void la_rows_sqnorm_f16_kernel(const _Float16* __restrict__ x, long n, int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) { const float v = (float)x[r * D + k]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[r] = s;
}

// MODE 0: kth radius (k+1-th smallest distance per row, the row's own zero included, as torch.kthvalue(k+1) over the full
//         row of the manifold-vs-manifold matrix);  MODE 1: membership (any column with dist <= radius[col]);
// MODE 2: write the distance tile.
// Workgroup = 4 waves x 32 rows; every wave walks all columns 128 at a time (4 MFMA column tiles share one row fragment).
// Fragments come straight from the row-major fp16 matrices: lane (r, h) of a 32x16 block is 16 contiguous bytes of row r.
template <int MODE>
__global__ __launch_bounds__(256) void la_pr_tile_kernel(const _Float16* __restrict__ rows, const float* __restrict__ rown, long nr,
                                                        const _Float16* __restrict__ cols, const float* __restrict__ coln, long nc,
                                                        int D, int kk, const float* __restrict__ radius, float* __restrict__ out_kth,
                                                        unsigned char* __restrict__ out_member, float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const long r0 = (long)blockIdx.x * 128 + wid * 32;
    if (r0 >= nr) return;                                      // whole wave out of range (no barriers in this kernel)
    const long arow = r0 + l31 < nr ? r0 + l31 : nr - 1;
    const _Float16* ap = rows + arow * D + lh * 8;
    // rows held by this lane: m(r) = (r&3) + 8*(r>>2) + 4*lh
    float na[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long m = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        na[r] = rown[m < nr ? m : nr - 1];
    }
    float best[MODE == 0 ? 16 : 1][PR_KMAX];
    unsigned member = 0u;
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int q = 0; q < PR_KMAX; ++q) best[MODE == 0 ? r : 0][q] = __builtin_huge_valf();
    }
    for (long c0 = 0; c0 < nc; c0 += 128) {
        f32x16 acc[4];
        const _Float16* bp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            const long c = c0 + j * 32 + l31;
            bp[j] = cols + (c < nc ? c : nc - 1) * D + lh * 8;
        }
        for (int k = 0; k < D; k += 16) {
            const f16x8 af = *reinterpret_cast<const f16x8*>(ap + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x8 bf = *reinterpret_cast<const f16x8*>(bp[j] + k);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long c = c0 + j * 32 + l31;
            const bool cok = c < nc;
            const float nb = coln[cok ? c : nc - 1];
            const float rad = (MODE == 1) ? radius[cok ? c : nc - 1] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = sqrtf(fmaxf(na[r] + nb - 2.f * acc[j][r], 1e-30f));
                if (MODE == 0) {
                    // sorted insert into the row's k+1 smallest (ascending), columns past the end never enter
                    float v = cok ? d : __builtin_huge_valf();
#pragma unroll
                    for (int q = 0; q < PR_KMAX; ++q) {
                        if (q < kk) {
                            const float lo = fminf(best[MODE == 0 ? r : 0][q], v);
                            v = fmaxf(best[MODE == 0 ? r : 0][q], v);
                            best[MODE == 0 ? r : 0][q] = lo;
                        }
                    }
                } else if (MODE == 1) {
                    if (cok && d <= rad) member |= 1u << r;
                } else {
                    const long m = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (cok && m < nr) out_dist[m * nc + c] = d;
                }
            }
        }
    }
    if (MODE == 0) {
        // merge the 32 per-lane lists of every row: pop the global minimum kk times (ties: any owner, the value is what counts)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float kth = 0.f;
            for (int t = 0; t < kk; ++t) {
                float m = best[MODE == 0 ? r : 0][0];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));      // stays inside each 32-lane half
                kth = m;
                // the lowest lane holding m pops its head
                const unsigned long long has = __ballot(best[MODE == 0 ? r : 0][0] == m);
                const unsigned half = (unsigned)(lh ? (has >> 32) : (has & 0xffffffffull));
                const int owner = __builtin_ctz(half ? half : 1u);
                if (l31 == owner) {
#pragma unroll
                    for (int q = 0; q + 1 < PR_KMAX; ++q) best[MODE == 0 ? r : 0][q] = best[MODE == 0 ? r : 0][q + 1];
                    best[MODE == 0 ? r : 0][PR_KMAX - 1] = __builtin_huge_valf();
                }
            }
            const long mrow = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (l31 == 0 && mrow < nr) out_kth[mrow] = kth;
        }
    } else if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned long long any = __ballot((member >> r) & 1u);
            const unsigned half = (unsigned)(lh ? (any >> 32) : (any & 0xffffffffull));
            const long mrow = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (l31 == 0 && mrow < nr) out_member[mrow] = half ? 1 : 0;
        }
    }
}

static int pr_check(const void* rows, long nr, const void* cols, long nc, int D, const void* ws) {
    LA_CHECK_ARG(rows && cols && ws && nr >= 1 && nc >= 1, "pr: bad args");
    LA_CHECK_ARG(D >= 16 && D % 16 == 0, "pr: the feature dimension must be a multiple of 16 (pad with zeros)");
    LA_CHECK_ARG((((size_t)rows | (size_t)cols) & 15) == 0, "pr: feature matrices must be 16-byte aligned");
    return LA_OK;
}

extern "C" size_t la_pr_workspace_floats(long nr, long nc) { return (size_t)(nr + nc); }

// kth[i] = (k+1)-th smallest Euclidean distance from row i to the columns (k = nhood_size; i's own zero distance counts)
extern "C" int la_pr_kth_f16(const void* rows, long nr, const void* cols, long nc, int D, int nhood_size, float* kth, float* ws,
                             hipStream_t stream) {
    int rc = pr_check(rows, nr, cols, nc, D, ws);
    if (rc) return rc;
    LA_CHECK_ARG(kth && nhood_size >= 0 && nhood_size + 1 <= PR_KMAX && nhood_size + 1 <= nc, "pr_kth: nhood_size out of range");
    float* rn = ws;
    float* cn = ws + nr;
    hipLaunchKernelGGL(la_rows_sqnorm_f16_kernel, dim3((unsigned)la_cdiv(nr, 4)), dim3(256), 0, stream, (const _Float16*)rows, nr, D, rn);
    hipLaunchKernelGGL(la_rows_sqnorm_f16_kernel, dim3((unsigned)la_cdiv(nc, 4)), dim3(256), 0, stream, (const _Float16*)cols, nc, D, cn);
    hipLaunchKernelGGL(la_pr_tile_kernel<0>, dim3((unsigned)la_cdiv(nr, 128)), dim3(256), 0, stream, (const _Float16*)rows, rn, nr,
                       (const _Float16*)cols, cn, nc, D, nhood_size + 1, (const float*)nullptr, kth, (unsigned char*)nullptr,
                       (float*)nullptr);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// member[i] = 1 if dist(row i, col j) <= radius[j] for any j
extern "C" int la_pr_member_f16(const void* rows, long nr, const void* cols, long nc, int D, const float* radius,
                                unsigned char* member, float* ws, hipStream_t stream) {
    int rc = pr_check(rows, nr, cols, nc, D, ws);
    if (rc) return rc;
    LA_CHECK_ARG(radius && member, "pr_member: bad args");
    float* rn = ws;
    float* cn = ws + nr;
    hipLaunchKernelGGL(la_rows_sqnorm_f16_kernel, dim3((unsigned)la_cdiv(nr, 4)), dim3(256), 0, stream, (const _Float16*)rows, nr, D, rn);
    hipLaunchKernelGGL(la_rows_sqnorm_f16_kernel, dim3((unsigned)la_cdiv(nc, 4)), dim3(256), 0, stream, (const _Float16*)cols, nc, D, cn);
    hipLaunchKernelGGL(la_pr_tile_kernel<1>, dim3((unsigned)la_cdiv(nr, 128)), dim3(256), 0, stream, (const _Float16*)rows, rn, nr,
                       (const _Float16*)cols, cn, nc, D, 0, radius, (float*)nullptr, member, (float*)nullptr);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// dist[i][j] = Euclidean distance (the matrix torch.cdist returns), float32 [nr][nc]
extern "C" int la_cdist_f16(const void* rows, long nr, const void* cols, long nc, int D, float* dist, float* ws, hipStream_t stream) {
    int rc = pr_check(rows, nr, cols, nc, D, ws);
    if (rc) return rc;
    LA_CHECK_ARG(dist, "cdist: bad args");
    float* rn = ws;
    float* cn = ws + nr;
    hipLaunchKernelGGL(la_rows_sqnorm_f16_kernel, dim3((unsigned)la_cdiv(nr, 4)), dim3(256), 0, stream, (const _Float16*)rows, nr, D, rn);
    hipLaunchKernelGGL(la_rows_sqnorm_f16_kernel, dim3((unsigned)la_cdiv(nc, 4)), dim3(256), 0, stream, (const _Float16*)cols, nc, D, cn);
    hipLaunchKernelGGL(la_pr_tile_kernel<2>, dim3((unsigned)la_cdiv(nr, 128)), dim3(256), 0, stream, (const _Float16*)rows, rn, nr,
                       (const _Float16*)cols, cn, nc, D, 0, (const float*)nullptr, (float*)nullptr, (unsigned char*)nullptr, dist);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// FeatureStats.append: raw_mean[i] += sum_k x[k][i];  raw_cov[i][j] += sum_k x[k][i] * x[k][j]   (float64 accumulators,
// float32 features).  16x16 output tile per workgroup, the batch streamed through LDS 16 rows at a time.
__global__ __launch_bounds__(256) void la_feature_moments_kernel(const float* __restrict__ x, long n, int D, double* __restrict__ raw_mean,
                                                                double* __restrict__ raw_cov) {
    __shared__ float xi[16][17], xj[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    double acc = 0.0, msum = 0.0;
    for (long k0 = 0; k0 < n; k0 += 16) {
        const long k = k0 + ty;
        xi[ty][tx] = (k < n && i0 + tx < D) ? x[k * D + i0 + tx] : 0.f;
        xj[ty][tx] = (k < n && j0 + tx < D) ? x[k * D + j0 + tx] : 0.f;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc += (double)xi[q][ty] * (double)xj[q][tx];
            if (blockIdx.y == 0 && ty == 0) msum += (double)xj[q][tx];
        }
        __syncthreads();
    }
    if (i0 + ty < D && j0 + tx < D) raw_cov[(long)(i0 + ty) * D + j0 + tx] += acc;
    if (blockIdx.y == 0 && ty == 0 && j0 + tx < D) raw_mean[j0 + tx] += msum;
}

extern "C" int la_feature_moments_f64(const float* x, long n, int D, double* raw_mean, double* raw_cov, hipStream_t stream) {
    LA_CHECK_ARG(x && raw_mean && raw_cov && n >= 0 && D >= 1, "feature_moments: bad args");
    if (n == 0) return LA_OK;
    hipLaunchKernelGGL(la_feature_moments_kernel, dim3(la_cdiv(D, 16), la_cdiv(D, 16)), dim3(256), 0, stream, x, n, D, raw_mean, raw_cov);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
