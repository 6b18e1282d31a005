// Implicit-GEMM convolution on the 16-bit MFMA path with fp32 operands split into 16-bit terms.
//
// fp32 MFMA runs at 1/16 of the 16-bit MFMA rate on gfx950 (MI355X_MICROARCH.md, Matrix cores), so an fp32 contraction is
// re-expressed as a few 16-bit x 16-bit products with fp32 accumulation (the products are exact in fp32, the MFMA
// accumulates in fp32, so the only approximation is the dropped low-order cross terms):
//     FMT_BF16X3:  x = h + m + l (each term the bf16 rounding of what the previous terms left), same for w;
//                  x.w ~= hh + (hm + mh) + (mm + hl + lh)     6 MFMAs, error ~1e-7 relative (fp32-class)
//     FMT_BF16X2:  x.w ~= hh + (hm + mh)                       3 MFMAs, error ~4e-6 relative (approximate mode)
//     FMT_F16X2:   x and w are first brought into fp16 range by exact power-of-two scales (one per weight tensor, one per
//                  sample of the modulated input, scaled max in [2^14, 2^15)), then x = h + l in fp16 (22 significand
//                  bits);  x.w ~= hh + hl + lh               3 MFMAs, error ~2e-7 relative (fp32-class); the accumulators
//                  are multiplied by the inverse scales before the epilogue (exact)
// One accumulator per output; within a K step the correction products are issued before the leading one.  Measured errors:
// tests/test_hip_ops.py against the oracle.
//
// Tiling: as la_conv.hip (256 threads = 2x2 waves, tile MT x 128 pixels, wave 64x64 = 2x2 MFMA tiles), K chunk =
// (one tap, 32 input channels) = two K=16 MFMA steps.  Weights are split at pack time into a FRAGMENT-ORDER pack
// (pack_slab_offset) that the waves read straight from global memory.  Pixels: the halo kernel reads the fp32 input and
// splits on the way into LDS; the flat kernel (whose gather re-reads every element once per tap) reads a pre-split copy
// made once per launch input (la_presplit_*: 8 bytes per element {h | m<<16, l} bf16, 4 bytes {h | l<<16} fp16).
#include "la_conv_device.h"
#include <atomic>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// FMT template parameter of the kernels below
#define FMT_BF16X3 3
#define FMT_BF16X2 2
#define FMT_F16X2 16
#define PRESPLIT_HDR 512       // head of the workspace: [0,256) xscale[b] floats; the segment maxima follow the header

template <bool F16>
__device__ __forceinline__ f32x16 la_mma(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

#define KCB 32                 // channels per chunk
#ifndef LA_MF3_DW
#define LA_MF3_DW 5
#endif

// ------------------------------------------------------------------------------------------------------------
// weight packing: W[o][i][t] (fp32) -> out[term][t][cc][m/32][k/16][lane][8] with (m,k) = (o,i) forward or (i,o) backward.
// Inside a (tap, 32-channel chunk) slab the 32-row x 16-channel blocks are stored in MFMA A-FRAGMENT order: lane
// (r = m%32, h = (k/8)%2) holds A[r][8h .. 8h+7] as 16 contiguous bytes, lanes contiguous -- so a wave fetches one
// fragment with a single fully coalesced 1 KB load, no LDS staging.  M is padded to a multiple of 32 with zero rows.
__host__ __device__ static inline int pack_mp(int M) { return (M + 31) & ~31; }
__device__ __forceinline__ long pack_slab_offset(int m, int k) {      // element offset of (row m, channel k) inside a slab
    return ((((long)(m >> 5) * 2 + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (m & 31)) << 3) + (k & 7);
}
__global__ void la_pack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout, int cin, int ktaps,
                                    int transpose, int nterm, float scale, int m_pad) {
    const int Mreal = transpose ? cin : cout, C = transpose ? cout : cin;
    const int M = pack_mp(m_pad > Mreal ? m_pad : Mreal);
    const int nck = (C + KCB - 1) / KCB;
    const long per_term = (long)ktaps * nck * M * KCB;
    for (long lin = blockIdx.x * (long)blockDim.x + threadIdx.x; lin < per_term; lin += (long)gridDim.x * blockDim.x) {
        const int k = (int)(lin % KCB);
        const int m = (int)((lin / KCB) % M);
        const int cc = (int)((lin / ((long)KCB * M)) % nck);
        const int t = (int)(lin / ((long)KCB * M * nck));
        const long idx = ((long)t * nck + cc) * M * KCB + pack_slab_offset(m, k);
        const int c = cc * KCB + k;
        float v = 0.f;
        if (c < C && m < Mreal) {
            const int o = transpose ? c : m, i = transpose ? m : c;
            v = w[((long)o * cin + i) * ktaps + t] * scale;
        }
        for (int q = 0; q < nterm; ++q) {
            const __bf16 h = (__bf16)v;
            out[(long)q * per_term + idx] = h;
            v -= (float)h;
        }
    }
}

// max |w| of a tensor into *amax_bits as a float bit pattern (the caller zeroes it first)
int la_absmax_bits(const float* w, long n, unsigned* amax_bits, hipStream_t stream);
long la_conv_bf16_pack_elems(int M, int C, int ktaps) { return (long)ktaps * la_cdiv(C, KCB) * pack_mp(M) * KCB; }

// pack layout: [3 bf16 terms][2 fp16 terms][pad to 16 B][wscale float]
size_t la_conv_split_pack_bytes(int M, int C, int ktaps) {
    return (size_t)5 * la_conv_bf16_pack_elems(M, C, ktaps) * 2 + 256;
}
__host__ __device__ static inline size_t pack_f16_offset(long term_elems) { return (size_t)3 * term_elems * 2; }
static inline size_t pack_wscale_offset(long term_elems) { return ((size_t)5 * term_elems * 2 + 15) & ~(size_t)15; }

// |w * scale| max over the tensor -> bit pattern via atomicMax (non-negative floats order like unsigned ints)
__global__ void la_absmax_kernel(const float* __restrict__ w, long n, float scale, unsigned* __restrict__ amax_bits) {
    __shared__ float red[4];
    float m = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i] * scale));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

int la_absmax_bits(const float* w, long n, unsigned* amax_bits, hipStream_t stream) {
    long b2 = la_cdiv(n, 256); if (b2 > 1024) b2 = 1024;
    hipLaunchKernelGGL(la_absmax_kernel, dim3((unsigned)b2), dim3(256), 0, stream, w, n, 1.f, amax_bits);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

__global__ void la_pack_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ out, const unsigned* __restrict__ amax_bits,
                                   float* __restrict__ wscale_out, int cout, int cin, int ktaps, int transpose, float scale,
                                   int m_pad) {
    const int Mreal = transpose ? cin : cout, C = transpose ? cout : cin;
    const int M = pack_mp(m_pad > Mreal ? m_pad : Mreal);
    const int nck = (C + KCB - 1) / KCB;
    const long per_term = (long)ktaps * nck * M * KCB;
    const float ws = la_pow2_scale(__uint_as_float(*amax_bits));
    if (blockIdx.x == 0 && threadIdx.x == 0) *wscale_out = ws;
    for (long lin = blockIdx.x * (long)blockDim.x + threadIdx.x; lin < per_term; lin += (long)gridDim.x * blockDim.x) {
        const int k = (int)(lin % KCB);
        const int m = (int)((lin / KCB) % M);
        const int cc = (int)((lin / ((long)KCB * M)) % nck);
        const int t = (int)(lin / ((long)KCB * M * nck));
        const long idx = ((long)t * nck + cc) * M * KCB + pack_slab_offset(m, k);
        const int c = cc * KCB + k;
        float v = 0.f;
        if (c < C && m < Mreal) {
            const int o = transpose ? c : m, i = transpose ? m : c;
            v = w[((long)o * cin + i) * ktaps + t] * scale * ws;
        }
        const _Float16 h = (_Float16)v;
        out[idx] = h;
        out[per_term + idx] = (_Float16)(v - (float)h);
    }
}

// packs EVERY split precision into `out` (la_conv_split_pack_bytes): bf16 x3 terms, fp16 x2 terms (+ their weight scale)
int la_pack_conv_weights_bf16(const float* w, void* out, int cout, int cin, int ktaps, int transpose, int nterm,
                              hipStream_t stream, float scale, int m_pad) {
    LA_CHECK_ARG(w && out && nterm >= 1 && nterm <= 3, "pack_bf16: bad args");
    LA_CHECK_ARG(((size_t)out & 15) == 0, "pack_bf16: output must be 16-byte aligned");
    const int Mreal = transpose ? cin : cout;
    const long n = la_conv_bf16_pack_elems(m_pad > Mreal ? m_pad : Mreal, transpose ? cout : cin, ktaps);
    long blocks = la_cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(la_pack_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, (__bf16*)out, cout, cin, ktaps,
                       transpose, 3, scale, m_pad);
    char* base = static_cast<char*>(out);
    float* wscale = reinterpret_cast<float*>(base + pack_wscale_offset(n));
    unsigned* amax = reinterpret_cast<unsigned*>(wscale) + 1;
    LA_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned), stream));
    const long nw = (long)cout * cin * ktaps;
    long b2 = la_cdiv(nw, 256); if (b2 > 1024) b2 = 1024;
    hipLaunchKernelGGL(la_absmax_kernel, dim3((unsigned)b2), dim3(256), 0, stream, w, nw, scale, amax);
    hipLaunchKernelGGL(la_pack_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, (_Float16*)(base + pack_f16_offset(n)), amax,
                       wscale, cout, cin, ktaps, transpose, scale, m_pad);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// fp16 operand scale, pass 1: max |x * scale| of every (b, c) plane in PM_NS segments, one workgroup per segment (no atomics)
#define PM_NS 8
__global__ __launch_bounds__(256) void la_plane_absmax_kernel(const float* __restrict__ in, long in_bstride,
                                                             const float* __restrict__ scale, int scale_stride,
                                                             float* __restrict__ pm, int C, long HW, int ns) {
    __shared__ float red[4];
    const int seg = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const float* ip = in + (long)b * in_bstride + (long)c * HW;
    const long per = ((HW + ns - 1) / ns + 3) & ~3l;
    const long p0 = seg * per, p1 = p0 + per < HW ? p0 + per : HW;
    float m = 0.f;
    if ((((size_t)ip | (size_t)(HW * 4)) & 15) == 0) {          // 16-byte aligned plane: float4 stream, 4 loads in flight
        const float4* ip4 = reinterpret_cast<const float4*>(ip);
        long q = p0 / 4 + threadIdx.x;
        const long q1 = p1 / 4;
        for (; q + 768 < q1; q += 1024) {
            const float4 v0 = ip4[q], v1 = ip4[q + 256], v2 = ip4[q + 512], v3 = ip4[q + 768];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                               fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))),
                               fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)))));
        }
        for (; q < q1; q += 256) {
            const float4 v = ip4[q];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        for (long p = q1 * 4 + threadIdx.x; p < p1; p += 256) m = fmaxf(m, fabsf(ip[p]));
    } else {
        for (long p = p0 + threadIdx.x; p < p1; p += 256) m = fmaxf(m, fabsf(ip[p]));
    }
    m *= fabsf(scale ? scale[(long)b * scale_stride + c] : 1.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pm[((long)b * C + c) * ns + seg] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Activation backward of the layer that produced `yref`, fused with pass 1 of the fp16 operand scale of the contraction that consumes
// the result: dx = dy * act'(yref) and the segment maxima of |dx| in one sweep (the discriminator / feature-net backward passes ran
// la_bias_act_grad_f32, la_plane_absmax_kernel and la_xscale_kernel for every backward contraction; now this kernel and
// la_xscale_pmax_kernel).  Same grid and segment layout as la_plane_absmax_kernel; dx may alias dy.
__global__ __launch_bounds__(256) void la_act_grad_pmax_kernel(const float* dy, const float* __restrict__ yref, float* dx,
                                                              float* __restrict__ pm, int C, long HW, int ns, int act, float alpha,
                                                              float gain, float clamp) {
    __shared__ float red[4];
    const int seg = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const long base = ((long)b * C + c) * HW;
    const long per = ((HW + ns - 1) / ns + 3) & ~3l;
    const long p0 = seg * per, p1 = p0 + per < HW ? p0 + per : HW;
    float m = 0.f;
    auto one = [&](float g, float y) { const float v = g * la_act_bwd_from_y(y, act, alpha, gain, clamp); m = fmaxf(m, fabsf(v)); return v; };
    if (((((size_t)(dy + base)) | ((size_t)(yref + base)) | ((size_t)(dx + base)) | (size_t)(HW * 4)) & 15) == 0) {
        const float4* g4 = reinterpret_cast<const float4*>(dy + base);
        const float4* y4 = reinterpret_cast<const float4*>(yref + base);
        float4* d4 = reinterpret_cast<float4*>(dx + base);
        long q = p0 / 4 + threadIdx.x;
        const long q1 = p1 / 4;
        for (; q + 256 < q1; q += 512) {
            const float4 ga = g4[q], gb = g4[q + 256], ya = y4[q], yb = y4[q + 256];
            d4[q] = make_float4(one(ga.x, ya.x), one(ga.y, ya.y), one(ga.z, ya.z), one(ga.w, ya.w));
            d4[q + 256] = make_float4(one(gb.x, yb.x), one(gb.y, yb.y), one(gb.z, yb.z), one(gb.w, yb.w));
        }
        for (; q < q1; q += 256) {
            const float4 ga = g4[q], ya = y4[q];
            d4[q] = make_float4(one(ga.x, ya.x), one(ga.y, ya.y), one(ga.z, ya.z), one(ga.w, ya.w));
        }
        for (long p = q1 * 4 + threadIdx.x; p < p1; p += 256) dx[base + p] = one(dy[base + p], yref[base + p]);
    } else {
        for (long p = p0 + threadIdx.x; p < p1; p += 256) dx[base + p] = one(dy[base + p], yref[base + p]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pm[((long)b * C + c) * ns + seg] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

int la_conv_act_grad_segments(long HW) {
    int ns = (int)(HW / 8192);
    return ns < 1 ? 1 : (ns > PM_NS ? PM_NS : ns);
}

// dx [B][C][HW] = dy * act'(yref); pm [B][C][la_conv_act_grad_segments(HW)] = segment maxima of |dx| (-> LaConvArgs::in_pmax)
int la_conv_act_grad_pmax(const float* dy, const float* yref, float* dx, float* pm, int B, int C, long HW, int act, float alpha, float gain,
                          float clamp, hipStream_t stream) {
    LA_CHECK_ARG(dy && yref && dx && pm && B >= 1 && C >= 1 && HW >= 1, "act_grad_pmax: bad arguments");
    LA_CHECK_ARG(B <= 65535 && C <= 65535, "act_grad_pmax: grid too large");
    const int ns = la_conv_act_grad_segments(HW);
    hipLaunchKernelGGL(la_act_grad_pmax_kernel, dim3(ns, C, B), dim3(256), 0, stream, dy, yref, dx, pm, C, HW, ns, act, alpha, gain, clamp);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// per-sample power-of-two scale from the segment maxima: xscale[b] = pow2(max over the sample)
__global__ __launch_bounds__(256) void la_xscale_kernel(const float* __restrict__ pm, float* __restrict__ xscale, int n) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float m = 0.f;
    for (int k = threadIdx.x; k < n; k += blockDim.x) m = fmaxf(m, pm[(long)b * n + k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) xscale[b] = la_pow2_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

__global__ __launch_bounds__(1024) void la_xscale_pmax_kernel(const float* __restrict__ pmax, int nseg, const float* __restrict__ scale,
                                                             int scale_stride, float* __restrict__ xscale, int C, float mult) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const float* pb = pmax + (long)b * C * nseg;
    const float* sb = scale ? scale + (long)b * scale_stride : nullptr;
    const int n = C * nseg;
    float m = 0.f;
    if (!sb && (n & 3) == 0 && (((size_t)pb) & 15) == 0) {      // plain maximum of a contiguous array: 16-byte loads
        const float4* p4 = reinterpret_cast<const float4*>(pb);
        for (int k = threadIdx.x; k < (n >> 2); k += 1024) {
            const float4 v = p4[k];
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    } else {
        for (int k = threadIdx.x; k < n; k += 1024) m = fmaxf(m, pb[k] * fabsf(sb ? sb[k / nseg] : 1.f));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) t = fmaxf(t, red[w]);
        xscale[b] = la_pow2_scale(mult * t);
    }
}

// xscale[b] = power-of-two operand scale of a tensor bounded by mult * max_c(|scale[b][c]| * max_seg pmax[b][c][seg])
int la_conv_xscale_from_pmax(const float* pmax, int nseg, const float* scale, int scale_stride, float mult, float* xscale, int B, int C,
                             hipStream_t stream) {
    LA_CHECK_ARG(pmax && xscale && nseg >= 1 && B >= 1 && C >= 1, "xscale_from_pmax: bad arguments");
    hipLaunchKernelGGL(la_xscale_pmax_kernel, dim3(B), dim3(1024), 0, stream, pmax, nseg, scale, scale_stride, xscale, C, mult);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// Pre-split copy for the flat kernel, CHANNEL-INTERLEAVED: q[b][chunk][pixel][32 channels]; fp16: a 128-byte record per (chunk, pixel) =
// the h terms of the 32 channels (64 B) followed by their l terms (64 B), so that a 16-byte piece is one LDS slot of one term;
// bf16: 8 B per element {h | m<<16, l}.  Channels past C are zeros.  The flat kernel's gather thread (pixel, 16-channel half) then reads its
// operand as 64 / 128 contiguous bytes (4 / 8 dwordx4) instead of 16 strided dwords, and a stride-2 gather wastes no sectors.
// One workgroup = 32 channels x 64 pixels, transposed through LDS.
struct LaInMask { const float* y; int act; float alpha, gain, clamp, in_gain; long p_lo, p_hi; };      // LaConvArgs::in_mask_* / in_gain; pixel range that is read (in_row_lo), 0 / 0 = all
template <bool F16>
__global__ __launch_bounds__(256) void la_presplit_t_kernel(const float* __restrict__ in, long in_bstride,
                                                           const float* __restrict__ scale, int scale_stride,
                                                           const float* __restrict__ xscale, int xs_fan, unsigned* __restrict__ out, int C, long HW,
                                                           LaInMask mk) {
    constexpr int EW = F16 ? 1 : 2;                           // dwords per element
    __shared__ unsigned tile[EW][64][33];
    const int cc = blockIdx.y, b = blockIdx.z, nck = gridDim.y;
    const long p0 = (long)blockIdx.x * 64;
    if (mk.p_hi > 0 && (p0 + 64 <= mk.p_lo || p0 >= mk.p_hi)) return;      // rows the launch reads as zeros anyway (LaConvArgs::in_row_lo): not copied
    const float xs = F16 ? la_xs_get(xscale, b, xs_fan) : 1.f;
    {
        const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
        const long p = p0 + px;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cl = cg * 8 + i, c = cc * KCB + cl;
            float v = 0.f;
            if (c < C && p < HW) {
                const long o = (long)b * in_bstride + (long)c * HW + p;
                v = in[o] * ((scale ? scale[(long)b * scale_stride + c] : 1.f) * xs);
                if (mk.y) v *= la_act_bwd_from_y(mk.y[o], mk.act, mk.alpha, mk.gain, mk.clamp);
                v *= mk.in_gain;
            }
            if (F16) {
                const _Float16 h = (_Float16)v;
                const _Float16 l = (_Float16)(v - (float)h);
                tile[0][px][cl] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
            } else {
                unsigned short t[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const __bf16 h = (__bf16)v;
                    t[q] = __builtin_bit_cast(unsigned short, h);
                    v -= (float)h;
                }
                tile[0][px][cl] = (unsigned)t[0] | ((unsigned)t[1] << 16);
                tile[EW - 1][px][cl] = (unsigned)t[2];
            }
        }
    }
    __syncthreads();
    {
        const int px = threadIdx.x >> 2, qt = threadIdx.x & 3;         // 8 channels of one pixel per thread
        const long p = p0 + px;
        if (p < HW) {
            unsigned* op = out + (((long)b * nck + cc) * HW + p) * (KCB * EW) + qt * 8 * EW;
            if (F16) {      // record = [h of 32 channels | l of 32 channels]: this thread's 8 channels are slot qt of each half
                unsigned e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = tile[0][px][qt * 8 + k];
                uint4* rec = reinterpret_cast<uint4*>(out + (((long)b * nck + cc) * HW + p) * KCB);
                rec[qt] = make_uint4(__builtin_amdgcn_perm(e[1], e[0], 0x05040100u), __builtin_amdgcn_perm(e[3], e[2], 0x05040100u),
                                     __builtin_amdgcn_perm(e[5], e[4], 0x05040100u), __builtin_amdgcn_perm(e[7], e[6], 0x05040100u));
                rec[4 + qt] = make_uint4(__builtin_amdgcn_perm(e[1], e[0], 0x07060302u), __builtin_amdgcn_perm(e[3], e[2], 0x07060302u),
                                         __builtin_amdgcn_perm(e[5], e[4], 0x07060302u), __builtin_amdgcn_perm(e[7], e[6], 0x07060302u));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    reinterpret_cast<uint4*>(op)[k] = make_uint4(tile[0][px][qt * 8 + 2 * k], tile[EW - 1][px][qt * 8 + 2 * k],
                                                                 tile[0][px][qt * 8 + 2 * k + 1], tile[EW - 1][px][qt * 8 + 2 * k + 1]);
            }
        }
    }
}

static inline size_t presplit_hdr_bytes(int B, int C) { return (PRESPLIT_HDR + (size_t)B * C * PM_NS * 4 + 255) & ~(size_t)255; }
size_t la_conv_presplit_bytes(int B, int C, int Hin, int Win) {      // channel-interleaved copy, C padded to whole chunks, 8 B / element
    return (size_t)B * la_cdiv(C, KCB) * KCB * Hin * Win * 8 + 16 + presplit_hdr_bytes(B, C);
}

// fp16 path: per-sample operand scale (segment maxima -> xscale[b]) in the header of the workspace; advances a.ws past it
static int prepare_scale(LaConvArgs& a, hipStream_t stream) {
    if (a.precision != LA_PREC_F16X2 || a.acc_scale_x) return LA_OK;      // (a preset scale: e.g. from the clamp bound of the producer)
    const long HW = (long)a.Hin * a.Win;
    const size_t hb = presplit_hdr_bytes(a.B, a.C);
    LA_CHECK_ARG(a.ws && a.ws_bytes >= hb, "conv: split precisions need a workspace (la_modconv_workspace_bytes)");
    LA_CHECK_ARG(((size_t)a.ws & 15) == 0, "conv: workspace must be 16-byte aligned");
    LA_CHECK_ARG(a.B <= 64, "conv: split precisions support at most 64 samples per launch");
    char* base = static_cast<char*>(a.ws);
    float* xscale = reinterpret_cast<float*>(base);
    float* pm = reinterpret_cast<float*>(base + PRESPLIT_HDR);       // segment maxima [B][C][ns]
    int ns = (int)(HW / 8192);
    ns = ns < 1 ? 1 : (ns > PM_NS ? PM_NS : ns);
    if (a.in_pmax) {      // the producer of `in` already reduced every plane: max over the sample of |style| * plane max
        hipLaunchKernelGGL(la_xscale_pmax_kernel, dim3(a.B), dim3(1024), 0, stream, a.in_pmax, a.in_pmax_nseg > 0 ? a.in_pmax_nseg : 1, a.in_scale,
                           a.scale_stride, xscale, a.C, 1.f);
    } else {
        hipLaunchKernelGGL(la_plane_absmax_kernel, dim3(ns, a.C, a.B), dim3(256), 0, stream, a.in, a.in_bstride, a.in_scale,
                           a.scale_stride, pm, a.C, HW, ns);
        hipLaunchKernelGGL(la_xscale_kernel, dim3(a.B), dim3(256), 0, stream, pm, xscale, a.C * ns);
    }
    LA_CHECK_LAUNCH();
    a.acc_scale_x = xscale; a.acc_scale_fan = 1;
    a.ws = base + hb;
    a.ws_bytes -= hb;
    return LA_OK;
}

// can this launch use the halo kernel?  dense stride-1 3x3 taps within +-1, grid = whole 4x32 tiles, above the split-K sizes
bool la_conv_bf16_uses_halo(const LaConvArgs& a) {
    if (a.precision == LA_PREC_F32 || a.in_q) return false;
    if (a.in_mask_y || (a.in_gain != 0.f && a.in_gain != 1.f)) return false;      // (an input mask is applied by the pre-split copy)
    if (a.in_sy != 1 || a.in_sx != 1 || a.out_sy != 1 || a.out_sx != 1 || a.out_oy != 0 || a.out_ox != 0) return false;
    if ((a.Gx & 31) != 0 || (a.Gy & 3) != 0 || a.Gy != a.Hout || a.Gx != a.Wout || a.ntaps != 9) return false;
    // Grids up to 34x34 stay on the split-K path (la_conv.hip SPLITK_MAX_G).  Round 3 measured the 32x32 layers (512 -> 512, K = 4608) on
    // this kernel: 114 us against 135 us for slices + finish pass + pre-split copy in isolation, no difference inside a batch -- and
    // the first-step gradient of the 1024^2 loop 12x further from float64 (rms 4.4e-6 against 3.7e-7 of max |g| 0.4; the reference's
    // float32: 1.5e-6): K slices summed afterwards are a blocked summation, one accumulator walking all 4608 terms is not.  Dev knob
    // LA_KNOB_HALO_MING: grids of at least that many points run here.
    if ((long)a.Gy * a.Gx < (la_dev_knob(LA_KNOB_HALO_MING) ? la_dev_knob(LA_KNOB_HALO_MING) : 1157)) return false;
    if ((long)a.C * a.Hin * a.Win >= (1l << 28) || a.C > 4096) return false;   // 32-bit byte offsets inside one sample, below the
                                                                             // out-of-range sentinel of the pixel-stationary loader
    for (int t = 0; t < a.ntaps; ++t)
        if (a.tap_dy[t] < -1 || a.tap_dy[t] > 1 || a.tap_dx[t] < -1 || a.tap_dx[t] > 1) return false;
    return true;
}

// Operand preparation of a split-precision launch.  Halo launches read the fp32 input directly (modulation, scaling and
// the split happen on the way into LDS) and only need the fp16 scale; every other launch gets a pre-split copy.
int la_conv_prepare_input(LaConvArgs& a, hipStream_t stream) {
    if (a.precision == LA_PREC_F32 || a.in_q) return LA_OK;
    const long HW = (long)a.Hin * a.Win;
    // launch profiler: operand preparation = its own class (read the fp32 input once; the pre-split copy is written once)
    struct Bracket { int slot; hipStream_t st; ~Bracket() { la_prof_close(slot, st); } };
    if (la_conv_bf16_uses_halo(a)) {
        if (a.precision != LA_PREC_F16X2 || a.acc_scale_x) return LA_OK;      // (scale known: nothing is launched, nothing is bracketed)
        Bracket br{la_prof_open(LA_PC_PRESPLIT, 0.0, a.in_pmax ? 0.0 : 4.0 * a.B * (double)a.C * HW * (a.in_bstride ? 1.0 : 1.0 / a.B), stream), stream};
        return prepare_scale(a, stream);
    }
    Bracket br{la_prof_open(LA_PC_PRESPLIT, 0.0, 4.0 * a.B * (double)a.C * HW * ((a.in_bstride ? 1.0 : 1.0 / a.B) + 1.0), stream), stream};
    const size_t qb = la_conv_presplit_bytes(a.B, a.C, a.Hin, a.Win);
    LA_CHECK_ARG(a.ws && a.ws_bytes >= qb, "conv: split precisions need a workspace (la_modconv_workspace_bytes)");
    LA_CHECK_ARG(((size_t)a.ws & 15) == 0, "conv: workspace must be 16-byte aligned");
    LA_CHECK_ARG(a.B <= 64, "conv: split precisions support at most 64 samples per launch");
    LA_CHECK_ARG(qb < 0x7ff00000u, "conv: pre-split operand too large for 32-bit buffer offsets");
    char* base = static_cast<char*>(a.ws);
    const size_t ws_bytes = a.ws_bytes;
    const dim3 pgrid((unsigned)la_cdiv(HW, 64), (unsigned)la_cdiv(a.C, KCB), (unsigned)a.B);
    const LaInMask mk{a.in_mask_y, a.in_mask_act, a.in_mask_alpha, a.in_mask_gain, a.in_mask_clamp, a.in_gain != 0.f ? a.in_gain : 1.f,
                      a.in_row_hi > 0 ? (long)a.in_row_lo * a.Win : 0, a.in_row_hi > 0 ? (long)a.in_row_hi * a.Win : 0};
    LA_CHECK_ARG((!a.in_mask_y && mk.in_gain == 1.f) || a.acc_scale_x || a.precision != LA_PREC_F16X2, "conv: an input mask needs a preset operand scale");
    if (a.precision == LA_PREC_F16X2) {
        int rc = prepare_scale(a, stream);
        if (rc) return rc;
        void* q = a.ws;
        hipLaunchKernelGGL(la_presplit_t_kernel<true>, pgrid, dim3(256), 0, stream, a.in, a.in_bstride, a.in_scale, a.scale_stride,
                           a.acc_scale_x, a.acc_scale_fan, (unsigned*)q, a.C, HW, mk);
        a.in_q = q;
    } else {
        void* q = base + presplit_hdr_bytes(a.B, a.C);
        hipLaunchKernelGGL(la_presplit_t_kernel<false>, pgrid, dim3(256), 0, stream, a.in, a.in_bstride, a.in_scale, a.scale_stride,
                           (const float*)nullptr, 0, (unsigned*)q, a.C, HW, mk);
        a.in_q = q;
    }
    LA_CHECK_LAUNCH();
    const size_t off = (qb + 255) & ~(size_t)255;
    a.ws = ws_bytes > off ? base + off : nullptr;
    a.ws_bytes = ws_bytes > off ? ws_bytes - off : 0;
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Flat variant: 128 consecutive grid positions per tile, any stride / tap table / ragged grid, optional split-K.
//   * B (pixels): thread (pixel, 16-channel half) gathers the tap-shifted inputs of one (chunk, tap) step with 16
//     unconditional buffer loads (clamped addresses; out-of-image pixels are zeroed on the way to LDS), one step ahead,
//     into the other of two swizzled LDS buffers (64-byte rows, slots XOR (row >> 2) & 3): ONE barrier per step.
//   * A (weights): MFMA fragments straight from the fragment-order pack, re-loaded for the next step right after the
//     MFMAs that read them have issued.  Never in LDS.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define BPITCH 64
// WV = waves per SIMD the kernel is compiled for (3: fp16 x2 pieces form only): as in the halo kernel below, ONE set of B fragments
// (K-step 1 re-loaded in place under the MFMAs of K-step 0; the next step's K-step 0 after the barrier) and one set of weight
// fragments re-loaded right after its MFMAs have issued.
// MF = 1 (three-wave fp16 x2 form, 128-row tiles): the step on v_mfma_f32_16x16x32_f16, as in the halo kernel (2 x 8 tiles of 16 x 16
// per wave, the pixel fragments read twice per step, LDS slots swizzled by 2 * ((pixel >> 2) & 1), accumulators brought into the
// 32x32 layout through LDS before the shared epilogue).
template <int MT, bool SPLIT, int FMT, int WV, int MFX = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WV))) void la_conv_bf16_kernel(LaConvArgs a_in) {
    // MFX bits 4 / 5 (development build only, WRONG results, timing only): the pixel records / the weight fragments are fetched for the
    // first tap of a chunk only and re-used for its other taps -- what ANY scheme that removes the per-tap re-read (a linear halo in
    // LDS, parity planes) could gain at most
    constexpr int MF = MFX & 15;
    constexpr bool ABL_PIX = (MFX & 16) != 0, ABL_WGT = (MFX & 32) != 0;
    constexpr bool ABL_BAR = (MFX & 64) != 0, ABL_LDSW = (MFX & 128) != 0;      // (same status) no barrier per step / no LDS write per step
    static_assert(MF == 0 || ((WV == 3 || ((MF == 3 || MF == 4) && WV <= 2)) && FMT == FMT_F16X2 && MT == 128), "the 16x16x32 form exists for the fp16 x2 kernel on 128-row tiles");
    // MF = 3 (round 5, split-K launches, two waves per SIMD): MF = 1 with the operand loads FOUR steps ahead instead of one.  A K slice of
    // the small grids is 9-72 (chunk, tap) steps of 48 MFMAs (0.3 us) walked by one or two workgroups per CU: with the weights of step
    // s + 1 requested during step s (MF 1) every step waited a memory round trip (1.2 us per step at 4^2 .. 16^2, whose weights come from
    // beyond L2, each byte once) -- nothing else is resident to cover it.  Here the weight fragments of steps s .. s + 3 and the pixel
    // pieces of steps s + 1 .. s + 3 are in registers / in flight (rings of four, loop unrolled by four so that the slots are static;
    // 128 more registers, hence two waves per SIMD), in ONE issue order in the prologue and in the loop so that every wait is a count.
    // MF = 2: MF = 1 on THREE pixel buffers.  The barrier at the end of step s then publishes the buffer of step s + 2, so the buffer of
    // step s + 1 is already complete while step s computes: its first fragments are read under the last MFMAs of step s, and no LDS
    // read latency is left exposed behind the barrier (two buffers: every step began with eight fragment reads nothing could cover).
    // merged output phases: blockIdx.z = phase * B + sample; the phase's grid, output offset and taps replace the launch-wide ones
    LaConvArgs a = a_in;
    int bz = blockIdx.z;
    int bx = blockIdx.x;
    if (a_in.nphase > 0) {
        int ph;
        if (SPLIT) {      // split-K form: blockIdx.x walks the phases' tiles back to back, blockIdx.z stays the K slice
            ph = 0;
#pragma unroll
            for (int q = 1; q < LA_CONV_MAX_PHASES; ++q)
                if (q < a_in.nphase && bx >= a_in.ph[q].tile0) ph = q;
            bx -= a_in.ph[ph].tile0;
            a.splitk_ws = a_in.splitk_ws + a_in.ph[ph].ws_off;
        } else {
            ph = bz / a_in.B;
            bz -= ph * a_in.B;
        }
        a.Gy = a_in.ph[ph].Gy; a.Gx = a_in.ph[ph].Gx; a.out_oy = a_in.ph[ph].out_oy; a.out_ox = a_in.ph[ph].out_ox; a.ntaps = a_in.ph[ph].ntaps;
#pragma unroll
        for (int t = 0; t < LA_CONV_PHASE_TAPS; ++t) { a.tap_dy[t] = a_in.ph[ph].tap_dy[t]; a.tap_dx[t] = a_in.ph[ph].tap_dx[t]; a.tap_w[t] = a_in.ph[ph].tap_w[t]; }
    }
    constexpr int NTERM = FMT == FMT_BF16X3 ? 3 : 2;
    constexpr bool F16 = FMT == FMT_F16X2;
    constexpr int WM_ = MT == 128 ? 4 : 2;         // wave grid WM_ x WN_ over the MT x 128 tile: every wave owns 32 rows
    constexpr int WN_ = 4 / WM_;                   // (128-row tiles: 4 x 1, no weight fragment is loaded by two waves)
    constexpr int TM = 1;
    constexpr int NJ = 4 / WN_;                    // 32-pixel MFMA tiles per wave
    constexpr int EB = F16 ? 4 : 8;                // bytes per pre-split element
    constexpr int BPLANE = NT * BPITCH;            // one term of one pixel buffer
    constexpr int BBUF = NTERM * BPLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2][NTERM][NT][BPITCH]
    float (*red)[MT] = reinterpret_cast<float (*)[MT]>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN_, wn = wid % WN_;
    // XCD-aware tile order (direct mode): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
    // run of pixel tiles -- vertically adjacent tiles (which share the +-1 row halos of the 3x3 taps) then hit the same L2.
    int ntile = bx;
    const int m0 = blockIdx.y * MT;
    const int G = a.Gy * a.Gx;
    const int Ntot = SPLIT ? a.B * G : G;
    if (!SPLIT) {
        // row window (LaConvArgs::row_lo): the tiles that hold a wanted row are a contiguous run [t0, t0 + nt) of the flattened tiles; the
        // first nt workgroups of the launch take them (in the XCD-aware order below, so the run is spread over all XCDs), the rest return
        int t0 = 0, nt = (int)gridDim.x;
        if (a.row_hi > 0) {
            // (round 5, as the halo kernel: the launch holds the window's tiles only, rounded up to a multiple of eight -- for merged phases
            //  those of the phase with the most -- and its workgroups zero the partials of the tiles outside the window in turn)
            const int tall = (G + NT - 1) / NT;
            t0 = (a.row_lo * a.Gx) / NT;
            int t1 = ((a.row_hi < a.Gy ? a.row_hi : a.Gy) * a.Gx + NT - 1) / NT;
            t1 = t1 < tall ? t1 : tall;
            nt = t1 - t0;
            for (int j = (int)blockIdx.x; j < tall - nt; j += (int)gridDim.x) la_conv_zero_partials<MT>(a, bz, m0, j < t0 ? j : j + nt);
            const int n8 = (int)gridDim.x;
            if ((n8 & 7) == 0) ntile = (blockIdx.x & 7) * (n8 >> 3) + (blockIdx.x >> 3);
            if (ntile >= nt) return;
        } else if ((nt & 7) == 0) ntile = (blockIdx.x & 7) * (nt >> 3) + (blockIdx.x >> 3);
        ntile += t0;
    }
    if (!SPLIT && (long)ntile * NT >= G) return;          // merged phases: the launch is sized for the largest phase
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- loader role: thread = (pixel n_l, 16-channel half khalf)
    const int n_l = tid & (NT - 1);
    const int khalf = tid >> 7;
    const int nidx_l = ntile * NT + n_l;
    const bool nvalid = nidx_l < Ntot;
    const int b_l = SPLIT ? (nvalid ? nidx_l / G : 0) : bz;
    const int g_l = SPLIT ? nidx_l - b_l * G : nidx_l;
    const int gy_l = nvalid ? g_l / a.Gx : 0;
    const int gx_l = nvalid ? g_l - gy_l * a.Gx : 0;
    const int iy0 = gy_l * a.in_sy, ix0 = gx_l * a.in_sx;
    const unsigned HWin = (unsigned)(a.Hin * a.Win);
    const int vy0 = a.in_row_hi > 0 ? a.in_row_lo : 0, vy1 = a.in_row_hi > 0 ? a.in_row_hi : a.Hin;      // valid input rows (LaConvArgs::in_row_lo)
    // fp16 pieces: load k of a thread is the 16-byte piece tid & 7 of the 128-byte record of pixel k * 32 + (tid >> 3) -- pieces 0-3 are
    // the h terms of channels 0-7 / 8-15 / 16-23 / 24-31 of the chunk, pieces 4-7 their l terms (la_presplit_t_kernel, la_fir4x4_adj_pack) --
    // so that the 8 lanes of a pixel read its whole record (a wave instruction touches 8 lines instead of 64) and a piece IS one 16-byte
    // LDS slot of one term: no unpacking between the load and the LDS write.
    // Pixel-stationary addressing (as the halo kernel's loader): the record offset of the un-shifted pixel and the set of taps that fall
    // outside the image are computed ONCE per piece; per step the tap adds a scalar to the offset and an out-of-image tap turns it
    // into an out-of-range buffer offset, which the hardware reads as zeros -- 3 vector instructions per piece and step (was ~25:
    // clamps, comparisons, a 64-bit multiply-add and a select per value).
    constexpr bool PIECES = FMT == FMT_F16X2;
    constexpr unsigned OOB = 0x7ff00000u;          // >= every operand size (checked by la_conv_prepare_input)
    unsigned plin[4] = {0u, 0u, 0u, 0u}, pinv[4] = {0u, 0u, 0u, 0u};
    if constexpr (PIECES) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n_k = k * 32 + (tid >> 3);
            const int nidx = ntile * NT + n_k;
            const bool pv = nidx < Ntot;
            const int bb = SPLIT ? (pv ? nidx / G : 0) : bz;
            const int g = SPLIT ? nidx - bb * G : nidx;
            const int gy = pv ? g / a.Gx : 0, gx = pv ? g - gy * a.Gx : 0;
            const int py = gy * a.in_sy, px = gx * a.in_sx;
            const unsigned base = (SPLIT ? (unsigned)bb * ((unsigned)((a.C + KCB - 1) / KCB) * KCB * HWin * 4u) : 0u) + (unsigned)(tid & 7) * 16u;
            plin[k] = base + (unsigned)(py * a.Win + px) * (unsigned)(KCB * EB);
            unsigned m = 0u;
#pragma unroll
            for (int t = 0; t < LA_CONV_MAX_TAPS; ++t) {
                const int iy = py + a.tap_dy[t], ix = px + a.tap_dx[t];
                const bool bad = !pv || iy < vy0 || iy >= vy1 || ix < 0 || ix >= a.Win;
                m |= (bad ? 1u : 0u) << t;
            }
            pinv[k] = m;
        }
    }

    const int nck = (a.C + KCB - 1) / KCB;
    int ck_beg = 0, ck_end = nck;
    if (SPLIT) {
        const int per = (nck + a.ksplit - 1) / a.ksplit;
        ck_beg = blockIdx.z * per;
        ck_end = ck_beg + per < nck ? ck_beg + per : nck;
    }
    const int ntaps = a.ntaps;
    const int nstep = ck_end > ck_beg ? (ck_end - ck_beg) * ntaps : 0;
    const long term_elems = a.wgt_bf16_term_elems;
    // buffer descriptors (wave-uniform).  Direct mode: this sample's pre-split input; split-K: the whole batch.
    // pre-split layout: [b][chunk][pixel][32 channels] -> a gather thread reads 16 contiguous channels of its pixel
    const unsigned samp_bytes = (unsigned)nck * KCB * HWin * EB;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.in_q)) + (SPLIT ? (size_t)0 : (size_t)bz * samp_bytes), 0,
        (int)(SPLIT ? samp_bytes * (unsigned)a.B : samp_bytes), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.wgt_bf16)) + (F16 ? pack_f16_offset(term_elems) : 0), 0, (int)(NTERM * term_elems * 2),
        0x00020000);
    const unsigned lane_base = (SPLIT ? (unsigned)b_l * samp_bytes : 0u) + (unsigned)(khalf * 16) * EB;

    // tap table -> packed scalars (offsets are within +-7), so the step loop needs no indexed kernarg reads
    unsigned long long dypack = 0ull, dxpack = 0ull, wpack = 0ull;
#pragma unroll
    for (int t = 0; t < LA_CONV_MAX_TAPS; ++t) {
        dypack |= (unsigned long long)((a.tap_dy[t] + 8) & 15) << (4 * t);
        dxpack |= (unsigned long long)((a.tap_dx[t] + 8) & 15) << (4 * t);
        wpack |= (unsigned long long)(a.tap_w[t] & 15) << (4 * t);
    }

    // ---- B gather of one step: 16 channels of this thread's pixel = 64 / 128 contiguous bytes
    unsigned ex[16], ey[NTERM == 3 ? 16 : 1];
    bool ok_r = false;
    auto load_b = [&](int cc, int t, unsigned kill = 0u) {      // kill = OOB (uniform): the pieces read as zeros whatever the tap (MF 3's padding steps)
        if constexpr (ABL_PIX) { if (t != 0) return; }
        if constexpr (PIECES) {
            const int dy = (int)((dypack >> (4 * t)) & 15u) - 8, dx = (int)((dxpack >> (4 * t)) & 15u) - 8;
            const unsigned delta = (unsigned)((dy * a.Win + dx) * (KCB * EB));      // (scalar)
            const unsigned so = (unsigned)cc * HWin * (KCB * EB);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int bad = __builtin_amdgcn_sbfe((int)pinv[k], (unsigned)t, 1u);      // -1: the tap is outside the image for this piece
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (plin[k] + delta) | ((unsigned)bad & OOB) | kill, so, 0);
                ex[4 * k] = v.x; ex[4 * k + 1] = v.y; ex[4 * k + 2] = v.z; ex[4 * k + 3] = v.w;
            }
            return;
        }
        const int iy = iy0 + (int)((dypack >> (4 * t)) & 15u) - 8, ix = ix0 + (int)((dxpack >> (4 * t)) & 15u) - 8;
        ok_r = nvalid && iy >= vy0 && iy < vy1 && ix >= 0 && ix < a.Win;
        const int iyc = iy < 0 ? 0 : (iy >= a.Hin ? a.Hin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= a.Win ? a.Win - 1 : ix);
        const unsigned vo = lane_base + (unsigned)(iyc * a.Win + ixc) * (KCB * EB);
        const unsigned so = (unsigned)cc * HWin * (KCB * EB);
        if (NTERM == 3) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, so + 16 * k, 0);
                ex[2 * k] = v.x; ey[NTERM == 3 ? 2 * k : 0] = v.y; ex[2 * k + 1] = v.z; ey[NTERM == 3 ? 2 * k + 1 : 0] = v.w;
            }
        } else if (F16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, so + 16 * k, 0);
                ex[4 * k] = v.x; ex[4 * k + 1] = v.y; ex[4 * k + 2] = v.z; ex[4 * k + 3] = v.w;
            }
        } else {      // 2 bf16 terms: the {h | m} words of the 8-byte elements
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, so + 16 * k, 0);
                ex[2 * k] = v.x; ex[2 * k + 1] = v.z;
            }
        }
    };
    const int wrow = n_l * BPITCH, wsw = (n_l >> 2) & 3;
    auto write_b = [&](unsigned char* buf) {
        if constexpr (PIECES) {      // piece (tid & 7) = slot (tid & 3) of term (tid >> 2 & 1); the tile row only adds k * 32 rows
            const int wsw = MF ? ((tid >> 5) & 1) << 1 : (tid >> 5);      // slot swizzle of pixel k * 32 + (tid >> 3): by (pixel >> 2) & 3, MF: 2 * ((pixel >> 2) & 1)
            unsigned char* p0 = buf + ((tid >> 2) & 1) * BPLANE + (tid >> 3) * BPITCH + ((((tid & 3) ^ wsw) & 3) << 4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<uint4*>(p0 + k * 32 * BPITCH) = make_uint4(ex[4 * k], ex[4 * k + 1], ex[4 * k + 2], ex[4 * k + 3]);
            return;
        }
#pragma unroll
        for (int q = 0; q < NTERM; ++q) {
            const unsigned sel = q == 1 ? 0x07060302u : 0x05040100u;
            unsigned w[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const unsigned e0 = q == 2 ? ey[NTERM == 3 ? 2 * d : 0] : ex[2 * d];
                const unsigned e1 = q == 2 ? ey[NTERM == 3 ? 2 * d + 1 : 0] : ex[2 * d + 1];
                const unsigned v = __builtin_amdgcn_perm(e1, e0, sel);
                w[d] = ok_r ? v : 0u;
            }
            unsigned char* p = buf + q * BPLANE + wrow;
            *reinterpret_cast<uint4*>(p + ((((khalf * 2) ^ wsw) & 3) << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(p + ((((khalf * 2 + 1) ^ wsw) & 3) << 4)) = make_uint4(w[4], w[5], w[6], w[7]);
        }
    };
    // B fragments: lane (l31, lh) of N-subtile j reads slot ks*2 + lh of row (wn*NJ + j)*32 + l31
    const int rsw = (l31 >> 2) & 3;
    const int rbase = (wn * NJ * 32 + l31) * BPITCH;
    auto read_b = [&](const unsigned char* buf, int ks, bf16x8 (&dst)[NTERM][NJ]) {
        const int o = rbase + ((((ks * 2 + lh) ^ rsw) & 3) << 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < NTERM; ++q) dst[q][j] = *reinterpret_cast<const bf16x8*>(buf + q * BPLANE + j * 32 * BPITCH + o);
    };

    // ---- A fragments straight from the fragment-order pack (blocks past M are clamped: their rows are never stored)
    const int Mp = pack_mp(a.M);
    unsigned a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int mblk = (m0 + wm * (MT / WM_) + i * 32) >> 5;
        mblk = mblk < (Mp >> 5) ? mblk : (Mp >> 5) - 1;
        a_off[i] = (unsigned)mblk * 2048u + (unsigned)lane * 16u;
    }
    const unsigned slab_bytes = (unsigned)Mp * KCB * 2u;         // one (tap, chunk) slab of one term
    const unsigned term_bytes = (unsigned)term_elems * 2u;
    auto load_a = [&](int cc, int t, int ks, bf16x8 (&dst)[NTERM][TM]) {
        const unsigned tw = (unsigned)((wpack >> (4 * t)) & 15u);
        const unsigned so = (tw * nck + cc) * slab_bytes + ks * 1024;
#pragma unroll
        for (int q = 0; q < NTERM; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                dst[q][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_off[i], so + q * term_bytes, 0));
    };

    f32x16 acc[TM][NJ];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma_step = [&](bf16x8 (&af)[NTERM][TM], bf16x8 (&bf)[NTERM][NJ]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // smallest products first, so they are not swamped by the leading term inside the accumulator
                if constexpr (NTERM == 3) {
                    acc[i][j] = la_mma<F16>(af[2][i], bf[0][j], acc[i][j]);   // lh
                    acc[i][j] = la_mma<F16>(af[0][i], bf[2][j], acc[i][j]);   // hl
                    acc[i][j] = la_mma<F16>(af[1][i], bf[1][j], acc[i][j]);   // mm
                }
                acc[i][j] = la_mma<F16>(af[1][i], bf[0][j], acc[i][j]);   // mh
                acc[i][j] = la_mma<F16>(af[0][i], bf[1][j], acc[i][j]);   // hm
                acc[i][j] = la_mma<F16>(af[0][i], bf[0][j], acc[i][j]);   // hh
            }
    };

    if constexpr (MF >= 1) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const int c16 = lane & 15, kq = lane >> 4;
      f32x4 acc16[2][8];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int n = 0; n < 8; ++n) acc16[mi][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (nstep > 0) {
        int c1 = ck_beg, t1 = 0;
        auto adv = [&](int& c, int& t) {
            if (t + 1 < ntaps) ++t;
            else if (c + 1 < ck_end) { ++c; t = 0; }
        };
        int mblk16 = (m0 + wm * 32) >> 5;
        mblk16 = mblk16 < (Mp >> 5) ? mblk16 : (Mp >> 5) - 1;
        const unsigned a16_off = (unsigned)mblk16 * 2048u + (unsigned)(kq * 32 + c16) * 16u;
        auto load_a16 = [&](int cc, int t, int mi, f16x8 (&dst)[2]) {
            if constexpr (ABL_WGT) { if (t != 0) return; }
            const unsigned tw = (unsigned)((wpack >> (4 * t)) & 15u);
            const unsigned so = (tw * nck + cc) * slab_bytes;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                dst[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a16_off + (unsigned)mi * 256u, so + q * term_bytes, 0));
        };
        const int rb16 = c16 * BPITCH + (((kq ^ (((c16 >> 2) & 1) << 1)) & 3) << 4);      // tile n adds n * 16 rows
        auto read_b16 = [&](const unsigned char* buf, int n, f16x8 (&dst)[2]) {
#pragma unroll
            for (int q = 0; q < 2; ++q) dst[q] = *reinterpret_cast<const f16x8*>(buf + q * BPLANE + n * 16 * BPITCH + rb16);
        };
        f16x8 a16[2][2], b16[4][2];
        const int c0 = c1, t0 = t1;      // (prologue in the loop's issue order: see the 32x32x16 form below)
      if constexpr (MF == 3 || MF == 4) {
        constexpr int DW = MF == 4 ? 9 : 3;            // weight ring depth = unroll factor (9-tap slices: no padding step).  MF 4 (one wave per SIMD,
                                                       // launches of at most one workgroup per CU): a whole 9-step slice's weights are requested up front
        f16x8 wr[DW][2][2];                            // weight fragments [slot][16-row half][term]
        // The loop walks the steps three at a time in straight-line code (static ring slots, one entry and one exit: the waits stay counts
        // and the accumulators keep one home); a slice whose step count is not a multiple of three is padded with steps whose pixel
        // pieces read as zeros (out-of-range buffer offsets) under re-loaded weights: they add exact zeros.
        // Issue order, prologue and loop alike:  P(0) | W(0) W(1) | [P(0) -> LDS] P(1) W(2) | iteration s: [P(s+1) -> LDS] P(s+2) W(s+3)
        // -- the wait for P(s+1) covers the weights up to step s+1 and leaves W(s+2) in flight: two steps for a weight fragment to arrive
        // where MF 1 gives it half a step.
        int cw = ck_beg, tw = 0;
        auto load_w = [&](int d) {                     // (d: a constant after unrolling)
            load_a16(cw, tw, 0, wr[d][0]);
            __builtin_amdgcn_sched_barrier(0);
            load_a16(cw, tw, 1, wr[d][1]); adv(cw, tw);
            __builtin_amdgcn_sched_barrier(0);
        };
        load_b(c0, t0);
        adv(c1, t1);                                   // (c1, t1) = step 1
        int c2 = c1, t2 = t1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < DW - 1; ++d) load_w(d);
        write_b(smem);                                 // step 0 -> buffer 0
        __builtin_amdgcn_sched_barrier(0);
        load_b(c1, t1, nstep > 1 ? 0u : OOB);          // step 1
        __builtin_amdgcn_sched_barrier(0);
        load_w(DW - 1);
        __syncthreads();
        adv(c2, t2);                                   // (c2, t2) = step 2
        auto body = [&](int D, int s) {                // (D: a constant after unrolling)
            const unsigned char* cur = smem + (s & 1) * BBUF;
            unsigned char* nxt = smem + ((s + 1) & 1) * BBUF;
#pragma unroll
            for (int n = 0; n < 4; ++n) read_b16(cur, n, b16[n]);
            write_b(nxt);                              // step s+1 (requested during step s-1)
            load_b(c2, t2, s + 2 < nstep ? 0u : OOB); adv(c2, t2);      // step s+2 (zeros past the slice's last step)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    f16x8 (&bs)[2] = b16[n & 3];
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[D][mi][1], bs[0], acc16[mi][n], 0, 0, 0);   // lh
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[D][mi][0], bs[1], acc16[mi][n], 0, 0, 0);   // hl
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[D][mi][0], bs[0], acc16[mi][n], 0, 0, 0);   // hh
                    if (n < 4) read_b16(cur, n + 4, bs);
                    else if (mi == 0) read_b16(cur, n - 4, bs);
                    __builtin_amdgcn_sched_barrier(0);
                }
                load_a16(cw, tw, mi, wr[D][mi]);       // this half's weights of step s+3
                __builtin_amdgcn_sched_barrier(0);
            }
            adv(cw, tw);
            __syncthreads();
        };
#pragma unroll 1
        for (int s = 0; s < nstep; s += DW) {
#pragma unroll
            for (int d = 0; d < DW; ++d) body(d, s + d);
        }
      } else if constexpr (MF == 2) {
        load_b(c0, t0);
        adv(c1, t1);                                   // (c1, t1) = step 1
        int c2 = c1, t2 = t1;
        write_b(smem);                                 // step 0 -> buffer 0
        __builtin_amdgcn_sched_barrier(0);
        load_b(c1, t1);
        adv(c2, t2);                                   // (c2, t2) = step 2
        write_b(smem + BBUF);                          // step 1 -> buffer 1
        __builtin_amdgcn_sched_barrier(0);
        load_b(c2, t2);                                // step 2: written by iteration 0
        __builtin_amdgcn_sched_barrier(0);
        load_a16(c0, t0, 0, a16[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_a16(c0, t0, 1, a16[1]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        adv(c2, t2);                                   // (c2, t2) = step 3
#pragma unroll
        for (int n = 0; n < 4; ++n) read_b16(smem, n, b16[n]);
        int ib = 0;                                    // buffer of step s
#pragma unroll 1
        for (int s = 0; s < nstep; ++s) {
            const int ib1 = ib == 2 ? 0 : ib + 1, ib2 = ib1 == 2 ? 0 : ib1 + 1;
            const unsigned char* cur = smem + ib * BBUF;
            const unsigned char* nx1 = smem + ib1 * BBUF;
            if constexpr (!ABL_LDSW) write_b(smem + ib2 * BBUF);                // step s+2 (loaded during step s-1)
            load_b(c2, t2);                            // step s+3
            __builtin_amdgcn_sched_barrier(0);
            const bool more = s + 1 < nstep;
            // every pixel fragment read ONCE per step (round 5, as the halo kernel's tap loop): quarters tiles 0-3 x rows 0-15, tiles 0-3 x rows
            // 16-31, tiles 4-7 x rows 0-15, tiles 4-7 x rows 16-31; a slot is re-filled after its second use with the tile four sub-steps
            // ahead (tile k + 4 of this step, then tile k of step s+1, published by the PREVIOUS barrier); the next step's weights get a
            // quarter step to land.  Round 4 walked all eight tiles per 16-row half and read every fragment twice
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int hf = q4 >> 1, mi = q4 & 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n = hf * 4 + k;
                    f16x8 (&bs)[2] = b16[k];
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][1], bs[0], acc16[mi][n], 0, 0, 0);   // lh
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][0], bs[1], acc16[mi][n], 0, 0, 0);   // hl
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][0], bs[0], acc16[mi][n], 0, 0, 0);   // hh
                    if (mi == 1) {
                        if (hf == 0) read_b16(cur, 4 + k, bs);
                        else if (more) read_b16(nx1, k, bs);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (q4 >= 2) load_a16(c1, t1, mi, a16[mi]);         // this half's weights of step s+1
            }
            adv(c1, t1);           // weights run one step ahead, pieces three
            adv(c2, t2);
            ib = ib1;
            if constexpr (!ABL_BAR) __syncthreads();
        }
      } else {
        load_b(c0, t0);
        adv(c1, t1);                                   // (c1, t1) = step 1
        int c2 = c1, t2 = t1;
        write_b(smem);
        __builtin_amdgcn_sched_barrier(0);
        load_b(c1, t1);
        __builtin_amdgcn_sched_barrier(0);
        load_a16(c0, t0, 0, a16[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_a16(c0, t0, 1, a16[1]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        adv(c2, t2);                                   // (c2, t2) = step 2
#pragma unroll 1
        for (int s = 0; s < nstep; ++s) {
            const unsigned char* cur = smem + (s & 1) * BBUF;
            unsigned char* nxt = smem + ((s + 1) & 1) * BBUF;
#pragma unroll
            for (int n = 0; n < 4; ++n) read_b16(cur, n, b16[n]);
            write_b(nxt);                              // step s+1 (loaded during step s-1)
            load_b(c2, t2);                            // step s+2
            __builtin_amdgcn_sched_barrier(0);
            // (every pixel fragment read once per step, as in the three-buffer form above)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int hf = q4 >> 1, mi = q4 & 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n = hf * 4 + k;
                    f16x8 (&bs)[2] = b16[k];
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][1], bs[0], acc16[mi][n], 0, 0, 0);   // lh
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][0], bs[1], acc16[mi][n], 0, 0, 0);   // hl
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][0], bs[0], acc16[mi][n], 0, 0, 0);   // hh
                    if (mi == 1 && hf == 0) read_b16(cur, 4 + k, bs);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (q4 >= 2) load_a16(c1, t1, mi, a16[mi]);         // this half's weights of step s+1
            }
            c1 = c2; t1 = t2;
            adv(c2, t2);
            __syncthreads();
        }
      }
      }
      // 16x16 tiles -> the 32x32 accumulator layout of the epilogue through LDS (free after the last barrier), two 32-pixel blocks at a time
      {
        float* tb = reinterpret_cast<float*>(smem) + wid * (64 * 36);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (hf) __syncthreads();
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) {
                    const int n = hf * 4 + n4;
                    *reinterpret_cast<f32x4*>(tb + (n4 * 16 + c16) * 36 + mi * 16 + kq * 4) = acc16[mi][n];
                }
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (jj * 32 + l31) * 36 + 8 * g + 4 * lh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[0][hf * 2 + jj][4 * g + r] = v[r];
                }
        }
        __syncthreads();
      }
    } else
    if constexpr (WV == 3) {
      if (nstep > 0) {
        static_assert(WV != 3 || (FMT == FMT_F16X2 ), "the three-wave form exists for the fp16 pieces loader only");
        int c1 = ck_beg, t1 = 0;
        auto adv = [&](int& c, int& t) {
            if (t + 1 < ntaps) ++t;
            else if (c + 1 < ck_end) { ++c; t = 0; }
        };
        bf16x8 acur[2][NTERM][TM], bf[NTERM][NJ];
        // Prologue in the loop's own issue order -- [pieces of the step after next | weights K-step 0 | weights K-step 1] are the
        // youngest loads when an iteration starts, on the first entry as on the back edge -- so that the counted waits of the loop
        // (pieces: all but the 4 weight loads; weights: per K-step) hold for both and never fall back to vmcnt(0): with the weights
        // requested first, every step began by waiting for the weight fragments issued just before its barrier.
        const int c0 = c1, t0 = t1;
        load_b(c0, t0);
        adv(c1, t1);                                   // (c1, t1) = step 1
        int c2 = c1, t2 = t1;
        write_b(smem);
        __builtin_amdgcn_sched_barrier(0);
        load_b(c1, t1);
        __builtin_amdgcn_sched_barrier(0);
        load_a(c0, t0, 0, acur[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_a(c0, t0, 1, acur[1]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        adv(c2, t2);                                   // (c2, t2) = step 2
#pragma unroll 1
        for (int s = 0; s < nstep; ++s) {
            const unsigned char* cur = smem + (s & 1) * BBUF;
            unsigned char* nxt = smem + ((s + 1) & 1) * BBUF;
            read_b(cur, 0, bf);
            write_b(nxt);                              // step s+1 (loaded during step s-1)
            load_b(c2, t2);                            // step s+2
            __builtin_amdgcn_sched_barrier(0);
            // K-step 0, every sub-tile re-loaded with its K-step 1 fragments right after its MFMAs
            {
                const int o1 = rbase + ((((2 + lh) ^ rsw) & 3) << 4);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[0][j] = la_mma<true>(acur[0][1][0], bf[0][j], acc[0][j]);
                    acc[0][j] = la_mma<true>(acur[0][0][0], bf[1][j], acc[0][j]);
                    acc[0][j] = la_mma<true>(acur[0][0][0], bf[0][j], acc[0][j]);
#pragma unroll
                    for (int q = 0; q < NTERM; ++q) bf[q][j] = *reinterpret_cast<const bf16x8*>(cur + q * BPLANE + j * 32 * BPITCH + o1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            load_a(c1, t1, 0, acur[0]);                // weights of step s+1, K-step 0
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[0][j] = la_mma<true>(acur[1][1][0], bf[0][j], acc[0][j]);
                acc[0][j] = la_mma<true>(acur[1][0][0], bf[1][j], acc[0][j]);
                acc[0][j] = la_mma<true>(acur[1][0][0], bf[0][j], acc[0][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_a(c1, t1, 1, acur[1]);
            c1 = c2; t1 = t2;
            adv(c2, t2);
            __syncthreads();
        }
      }
    } else
    if (nstep > 0) {
        // (chunk, tap) of steps s, s+1, s+2; past the end they stay on the last valid step (harmless re-loads)
        int c1 = ck_beg, t1 = 0;
        auto adv = [&](int& c, int& t) {
            if (t + 1 < ntaps) ++t;
            else if (c + 1 < ck_end) { ++c; t = 0; }
        };
        bf16x8 acur[2][NTERM][TM], anxt[2][NTERM][TM], bf0[NTERM][NJ], bf1[NTERM][NJ];
        // (prologue in the loop's issue order -- pixel loads of the step after next, then the weight loads -- so that the loop's
        //  waits are exact counts on both of its entries: see the three-wave form above)
        const int c0 = c1, t0 = t1;
        load_b(c0, t0);
        adv(c1, t1);                                   // (c1, t1) = step 1
        int c2 = c1, t2 = t1;
        write_b(smem);
        __builtin_amdgcn_sched_barrier(0);
        load_b(c1, t1);
        __builtin_amdgcn_sched_barrier(0);
        load_a(c0, t0, 0, acur[0]);
        load_a(c0, t0, 1, acur[1]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        adv(c2, t2);                                   // (c2, t2) = step 2
#pragma unroll 1
        for (int s = 0; s < nstep; ++s) {
            const unsigned char* cur = smem + (s & 1) * BBUF;
            unsigned char* nxt = smem + ((s + 1) & 1) * BBUF;
            // (the fences pin the issue order: left alone, the scheduler sinks every load to just before its first use)
            read_b(cur, 0, bf0);
            read_b(cur, 1, bf1);
            write_b(nxt);                              // step s+1 (loaded during step s-1)
            load_b(c2, t2);                            // step s+2
            load_a(c1, t1, 0, anxt[0]);                // weights of step s+1: a full step ahead (they may come from beyond L2)
            load_a(c1, t1, 1, anxt[1]);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(acur[0], bf0);
            mma_step(acur[1], bf1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < NTERM; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acur[ks][q][i] = anxt[ks][q][i];
            c1 = c2; t1 = t2;
            adv(c2, t2);
            __syncthreads();
        }
    }
    if (F16) {
        // undo the power-of-two operand scales (exact)
        const float iw = 1.f / a.acc_scale_w[0];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int bb = bz;
            if (SPLIT) { const int nidx = ntile * NT + (wn * NJ + j) * 32 + l31; bb = nidx < Ntot ? nidx / G : 0; }
            const float inv = iw / la_xs_get(a.acc_scale_x, bb, a.acc_scale_fan);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
        }
    }
    la_conv_epilogue<MT, SPLIT, false, WM_>(a, acc, red, ntile, m0, G, Ntot, SPLIT ? -1 : bz);
}

// ------------------------------------------------------------------------------------------------------------
// Halo variant for dense stride-1 3x3 launches on grids that tile exactly into 4 x 32 pixel tiles (the >= 64x64 layers,
// i.e. the bulk of the FLOPs).  The flat kernel above re-gathers every input element once per tap (9x) from L2; here the
// (4+2) x (32+2) halo of a 32-channel chunk is staged in LDS ONCE and the 9 taps read shifted fragments from it.
//   * B (pixels): two halo buffers.  While chunk cc computes, chunk cc+1 streams in, one ninth per tap: each thread loads a
//     4-channel unit, holds it for one tap (4 / 8 registers) and writes it to the other buffer during the next tap, so an
//     HBM miss has a whole tap of MFMAs to land and no wait ever covers more than one tap's loads.  ONE barrier per chunk.
//     Rows are 64 B (32 channels of one term) with the 16-byte slots XOR-swizzled by (pixel >> 2) & 3: conflict-free
//     ds_read_b128 fragments at any tap shift, no padding.
//   * A (weights): never touches LDS.  The pack stores every 32-row x 16-channel block in MFMA fragment order, so a wave
//     loads a fragment with one coalesced 1 KB buffer load; each fragment register is re-loaded for the next tap right
//     after the MFMAs that read it have issued.
// LDS: 2 x 204 px x 64 B x NTERM = 51 / 76.5 KB; registers <= 168 (NTERM = 2: three waves per SIMD) / <= 256.
#define HALO_W 34
#define HALO_PX (6 * HALO_W)
#define HPITCH 64
#define H_UNITS (8 * HALO_PX)          // (4-channel group, halo pixel) load units per chunk
#define H_UPT 182                      // units per tap (9 x 182 >= 1632)
// WV = waves per SIMD the kernel is compiled for.  2: two sets of B fragments (K-step 1 is read under the MFMAs of K-step 0), the chunk
// loop in a with-next and a last instance.  3 (<= 168 registers, three workgroups per CU -- the partner workgroups cover a
// workgroup's prologue and store bursts): ONE set of B fragments, every 32-pixel sub-tile re-loaded for the following K-step right
// after its own three MFMAs, and ONE instance of the chunk loop (the last chunk issues dummy loads): the merge of two instances
// cost a second set of 64 accumulator registers and 64 moves per chunk.  Measured per 154.6-GFLOP launch (WV 3 against 2):
// 128->128 @256^2 fwd -9.5 %, bwd -7 %; 256->256 @128^2 fwd -8 %, bwd -8 %; 512->512 @64^2 +-0 (two rounds of workgroups only) --
// la_conv_bf16_dispatch uses WV 3 for every 128-row fp16 launch.
// MF = 1 (three-wave fp16 x2 form on 128-row tiles only): the same wave tile (32 rows x 128 pixels) on v_mfma_f32_16x16x32_f16 -- 2 x 8
// tiles of 16 x 16, one MFMA per (tile, term pair) over the whole 32-channel chunk.  A 16x16x32 fragment feeds half the FLOPs of a
// 32x32x16 one, so with the same 32 fragment registers per operand every pixel fragment is read from LDS twice per tap (once per
// 16-row half); the weight fragments of a half die at the middle of the tap and are re-loaded then, as the K-step fragments are in the
// 32x32x16 form.  Same weight pack (a 16-row fragment is four 256-byte pieces of the 32-row block), LDS slots swizzled by
// 2 * ((pixel >> 2) & 1) (conflict-free for this lane map at every tap shift).  The accumulators are brought into the 32x32 layout
// through LDS before the shared epilogue.
#ifdef LA_DEV
// Development build, dev knob LA_KNOB_HALO_STAMP = 1: every wave of the MF 5 halo kernel accumulates s_memtime differences per segment
// (prologue issue / prologue wait + first stage / tap loops / chunk barriers / accumulator hand-over / epilogue) in scalar registers and
// leaves them in la_dbg_buf[wave][16] (la_dev_dbg_read); segments 6-8 are stamped inside the epilogue (la_conv_device.h, LA_ESTAMP).  scripts/halo_wave_timeline.py
__device__ unsigned long long la_dbg_buf[1 << 18];
#define LA_STAMP_DECL LaStamp stv; stv.on = a.dbg_stamp != 0; stv.last = 0ull; for (int i_ = 0; i_ < 12; ++i_) stv.seg[i_] = 0ull; if (stv.on) stv.last = __builtin_amdgcn_s_memtime();
#define LA_STAMP(i) do { if (stv.on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stv.seg[i] += t_ - stv.last; stv.last = t_; } } while (0)
#define LA_STAMP_ARG , -1, &stv
#define LA_STAMP_OUT do { if (stv.on && (threadIdx.x & 63) == 0) { const long wv_ = ((long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)) * 4 + (threadIdx.x >> 6); \
    if (wv_ * 16 + 16 <= (1 << 18)) { for (int i_ = 0; i_ < 12; ++i_) la_dbg_buf[wv_ * 16 + i_] = stv.seg[i_]; la_dbg_buf[wv_ * 16 + 12] = stv.last; \
    unsigned hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); \
    la_dbg_buf[wv_ * 16 + 13] = hw_; la_dbg_buf[wv_ * 16 + 14] = xcc_; } } } while (0)
extern "C" int la_dev_dbg_read(unsigned long long* dst, long n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(la_dbg_buf), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#else
#define LA_STAMP_DECL
#define LA_STAMP(i)
#define LA_STAMP_ARG
#define LA_STAMP_OUT
#endif
template <int MT, int FMT, int WV, int MF = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WV))) void la_conv_bf16_halo_kernel(LaConvArgs a) {
    constexpr bool SB = WV == 3;
    static_assert(MF == 0 || FMT == FMT_F16X2, "the MF forms exist for the fp16 x2 format");
    static_assert((MF & 1) == 0 || (WV == 3 && MT == 128), "the 16x16x32 form exists for the three-wave kernel on 128-row tiles");
    // MF bits: 1 = 16x16x32 MFMA; 2 = dev ablation (the loader skips the modulation and the fp16 split arithmetic: WRONG results, timing
    // only); 4 = pixel-stationary halo loader (below)
    constexpr bool M16 = (MF & 1) != 0, ABL = (MF & 2) != 0, PSL = (MF & 4) != 0;
    constexpr bool FR1 = (MF & 16) != 0;        // 16x16x32 form with every pixel fragment read once per tap (tap loop below)
    static_assert(!FR1 || (M16 && PSL), "the fragment-once order exists for MF 5");
    constexpr int NTERM = FMT == FMT_BF16X3 ? 3 : 2;
    constexpr bool F16 = FMT == FMT_F16X2;
    constexpr int WM_ = MT / 32;                   // wave grid WM_ x WN_ over the MT x 128 tile: every wave owns 32 rows
    constexpr int WN_ = 4 / WM_;                   // (128-row tiles: 4 x 1, no weight fragment is loaded by two waves; 32-row tiles
                                                   //  for the 32-channel layers of the 1024^2 generators: 1 x 4)
    constexpr int TM = 1;
    constexpr int NJ = 4 / WN_;                    // 32-pixel MFMA tiles (= tile rows) per wave
    constexpr int EB = 4;                          // the halo kernel reads the fp32 input itself
    constexpr int HPLANE = HALO_PX * HPITCH;       // one term of one halo buffer
    constexpr int HBUF = NTERM * HPLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2][NTERM][HALO_PX][HPITCH] + scl[nck*32]
    float (*red)[MT] = reinterpret_cast<float (*)[MT]>(smem);
    // per-channel factor (style modulation x fp16 sample scale) behind the halo buffers; a single-chunk launch (<= 32 input channels: the
    // top layers of the 1024^2 generators) never stages a second chunk and gets ONE buffer, i.e. twice the workgroups per CU
    float* scl = reinterpret_cast<float*>(smem + (a.C > KCB ? 2 : 1) * HBUF);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN_, wn = wid % WN_;
    const int tpr = a.Gx >> 5;
    // row window (LaConvArgs::row_lo): the 4-row tiles that hold a wanted row are the run [t0, t0 + nt) of the row-major tile order; the
    // first nt workgroups of the launch take them -- in the XCD-aware order, so that the run is spread over all eight XCDs (a test on the
    // tile row alone left the XCDs that own the top and the bottom of the frame idle and the launch as long as before) -- the rest return
    // Round 5: the launch holds the window's tiles ONLY, rounded up to a multiple of eight workgroups (la_conv_window_tiles; round 4 launched
    // a workgroup per tile of the whole frame and let those outside the window return: a launch with 1 104 wanted tiles of 2 048 took
    // ~30 us longer than the wanted tiles alone).  Workgroup x of the launch takes tile (x & 7) * (n8 / 8) + (x >> 3) of the window's
    // row-major run -- a contiguous run of tiles per XCD whatever the tile count (round 4 fell back to the plain order whenever the
    // count was not a multiple of eight: the 256^2 and 64^2 windows of config B) -- and the per-tile partials of the tiles outside
    // the window, which the one-pass style finish sums, are zeroed by the launch's workgroups in turn.
    int nt = (int)gridDim.x;
    int r0 = 0, c0 = 0, cw = tpr;                 // window rectangle in tiles: rows [r0, r1), columns [c0, c0 + cw)
    int ntile = blockIdx.x;
    if (a.row_hi > 0) {
        const int r1 = ((a.row_hi < a.Gy ? a.row_hi : a.Gy) + 3) >> 2;
        r0 = a.row_lo >> 2;
        if (a.col_hi > 0) { c0 = a.col_lo >> 5; cw = (((a.col_hi < a.Gx ? a.col_hi : a.Gx) + 31) >> 5) - c0; }
        nt = (r1 - r0) * cw;
        const int tall = (a.Gy >> 2) * tpr, n_out = tall - nt;
        for (int j = (int)blockIdx.x; j < n_out; j += (int)gridDim.x) {      // tiles outside the window, in row-major order
            int otile, k = j;
            const int per = tpr - cw;             // outside tiles per window row
            if (k < r0 * tpr) otile = k;
            else if ((k -= r0 * tpr) < (r1 - r0) * per) { const int rr = k / per, kk = k - rr * per; otile = (r0 + rr) * tpr + (kk < c0 ? kk : kk + cw); }
            else otile = r1 * tpr + (k - (r1 - r0) * per);
            la_conv_zero_partials<MT>(a, (int)blockIdx.z, (int)blockIdx.y * MT, otile);
        }
        const int n8 = (int)gridDim.x;            // (host: the window's tile count rounded up to a multiple of 8, or the whole frame's)
        if ((n8 & 7) == 0) {
            // eight runs of tiles, one per XCD, as even as the count allows (nt = 8 q + r: the first r runs hold q + 1 tiles); the run an
            // XCD takes rotates with the sample, so that over the samples of a launch every XCD sees long and short runs alike (the
            // 28-tile window at 64^2 with one fixed run per XCD: 4 4 4 4 4 4 4 0 tiles per sample; rotated: 28 per XCD over 8 samples)
            const int q = nt >> 3, r = nt & 7, rot = gridDim.z >= 8 ? 1 : (gridDim.z >= 4 ? 2 : (gridDim.z >= 2 ? 4 : 0));
            const int kv = (int)((blockIdx.x + blockIdx.z * rot) & 7), idx = (int)(blockIdx.x >> 3);
            if (idx >= q + (kv < r ? 1 : 0)) return;
            ntile = kv * q + (kv < r ? kv : r) + idx;
        } else if (ntile >= nt) return;
        const int rr = ntile / cw;
        ntile = (r0 + rr) * tpr + c0 + (ntile - rr * cw);
    } else if ((nt & 7) == 0) ntile = (blockIdx.x & 7) * (nt >> 3) + (blockIdx.x >> 3);   // XCD-contiguous tile runs
    const int m0 = blockIdx.y * MT;
    const int b = blockIdx.z;
    const int G = a.Gy * a.Gx;
    const int tyb = ntile / tpr, txb = ntile - tyb * tpr;
    const int y0 = tyb * 4 - 1, x0 = txb * 32 - 1;              // grid coordinates of halo pixel (0, 0)
    const unsigned HWin = (unsigned)(a.Hin * a.Win);
    const int vy0 = a.in_row_hi > 0 ? a.in_row_lo : 0, vy1 = a.in_row_hi > 0 ? a.in_row_hi : a.Hin;      // valid input rows (LaConvArgs::in_row_lo)
    const int nck = (a.C + KCB - 1) / KCB;
    const long term_elems = a.wgt_bf16_term_elems;
    const int l31 = lane & 31, lh = lane >> 5;
    // buffer descriptors (wave-uniform): this sample's fp32 input, and the weight pack of this format
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.in) + (size_t)b * a.in_bstride, 0, (int)((unsigned)a.C * HWin * EB), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.wgt_bf16)) + (F16 ? pack_f16_offset(term_elems) : 0), 0, (int)(NTERM * term_elems * 2),
        0x00020000);

    // tap table -> two packed scalars, so the tap loop needs no indexed kernarg reads
    unsigned long long shpack = 0ull, wpack = 0ull;
    unsigned xpack = 0u;                       // 1 + dx of every tap (PSL: the LDS slot swizzle follows the halo COLUMN)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        shpack |= (unsigned long long)((1 + a.tap_dy[t]) * HALO_W + (1 + a.tap_dx[t])) << (7 * t);
        wpack |= (unsigned long long)a.tap_w[t] << (4 * t);
        xpack |= (unsigned)(1 + a.tap_dx[t]) << (2 * t);
    }

    // ---- halo slices.  Every load is unconditional (clamped address; out-of-image pixels are zeroed on the way to LDS,
    // channels past C meet zero weights), so the compiler can count them: no wait in the tap loop is a vmcnt(0).
    struct Slice { float x[4]; int wr, c0; bool ok; };
    auto slice_load = [&](int cc, int t, Slice& sl, bool live = true) {      // !live (uniform): one dword of traffic per wave, nothing written
        const int lt = (tid + 64 * t) & 255;                      // the idle lanes rotate over the waves
        int u = t * H_UPT + (lt < H_UPT ? lt : H_UPT - 1);
        const bool valid = live && lt < H_UPT && u < H_UNITS;
        u = u < H_UNITS ? u : H_UNITS - 1;
        const int c4 = u / HALO_PX, hp = u - c4 * HALO_PX;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int iy = y0 + hy, ix = x0 + hx;
        sl.ok = iy >= vy0 && iy < vy1 && ix >= 0 && ix < a.Win;
        const int iyc = iy < 0 ? 0 : (iy >= a.Hin ? a.Hin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= a.Win ? a.Win - 1 : ix);
        const unsigned off = (unsigned)(iyc * a.Win + ixc) * EB;
        const int swz = M16 ? ((hp >> 2) & 1) << 1 : (hp >> 2);
        sl.wr = valid ? hp * HPITCH + ((((c4 >> 1) ^ swz) & 3) << 4) + (c4 & 1) * 8 : -1;
        sl.c0 = cc * KCB + c4 * 4;
        const bool fast = cc * KCB + KCB <= a.C;                  // uniform: only a ragged last chunk clamps channels
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned vo, so;
            if (fast) { vo = live ? (unsigned)(c4 * 4) * HWin * EB + off : 0u; so = live ? (unsigned)(cc * KCB + j) * HWin * EB : 0u; }
            else {
                const int c = cc * KCB + c4 * 4 + j;
                vo = (unsigned)(c < a.C ? c : a.C - 1) * HWin * EB + off;
                so = 0u;
            }
            sl.x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, vo, so, 0));
        }
    };
    // modulate (+ scale), split into NTERM 16-bit terms (each the rounding of the remainder), 8 bytes per term
    auto slice_write = [&](unsigned char* buf, const Slice& sl) {
        if (sl.wr >= 0) {
            // (2-wide vector types so that the packed v_cvt_pk_* / v_pk_* instructions are selected)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const float4 f = *reinterpret_cast<const float4*>(scl + sl.c0);
            // (scalar fp32 arithmetic, only the conversions are packed: no v_pk_*_f32, Makefile)
            float p[4] = {sl.x[0] * f.x, sl.x[1] * f.y, sl.x[2] * f.z, sl.x[3] * f.w};
            if constexpr (ABL) {
                const uint2 w = make_uint2(__builtin_bit_cast(unsigned, sl.x[0]), __builtin_bit_cast(unsigned, sl.x[2]));
                *reinterpret_cast<uint2*>(buf + sl.wr) = w;
                *reinterpret_cast<uint2*>(buf + HPLANE + sl.wr) = w;
                return;
            }
#pragma unroll
            for (int q = 0; q < NTERM; ++q) {
                uint2 w;
                if constexpr (F16) {
                    const f16x2 h0 = __builtin_convertvector(f32x2{p[0], p[1]}, f16x2), h1 = __builtin_convertvector(f32x2{p[2], p[3]}, f16x2);
                    w = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                    if (q + 1 < NTERM) { p[0] -= (float)h0[0]; p[1] -= (float)h0[1]; p[2] -= (float)h1[0]; p[3] -= (float)h1[1]; }
                } else {
                    const bf16x2_t h0 = __builtin_convertvector(f32x2{p[0], p[1]}, bf16x2_t), h1 = __builtin_convertvector(f32x2{p[2], p[3]}, bf16x2_t);
                    w = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                    if (q + 1 < NTERM) { p[0] -= (float)h0[0]; p[1] -= (float)h0[1]; p[2] -= (float)h1[0]; p[3] -= (float)h1[1]; }
                }
                *reinterpret_cast<uint2*>(buf + q * HPLANE + sl.wr) = sl.ok ? w : make_uint2(0u, 0u);
            }
        }
    };

    // ---- pixel-stationary form of the halo loader (PSL).  The slices above give every thread a different (channel group, halo pixel)
    // unit in every tap and recompute its image position, clamps and LDS slot from scratch: ~45 integer instructions per tap and
    // thread (five of them quarter-rate 32-bit multiplies) beside the MFMAs.  Here thread hp < 204 owns halo pixel hp for the whole
    // kernel -- image offset, validity and LDS row are computed ONCE -- and tap t (0..7) stages channel group t of the next chunk
    // for it (tap 8 repeats the loads of tap 0 and drops them, so that every wait in the tap loop stays a counted vmcnt): the channel is
    // wave-uniform, i.e. scalar arithmetic, and what is left per tap are the four loads, the split and one XOR for the LDS slot.
    // With it the 16-byte slots of a pixel row are XOR-swizzled by the halo COLUMN (hx >> 2) instead of the linear pixel index: equally
    // conflict-free (a fragment read covers consecutive columns of one halo row), but the swizzle of a fragment read then depends on
    // the lane and the tap's dx only -- not on the tile row or dy -- so ONE lane address per tap serves every fragment read of the tap
    // through immediate offsets (was: five instructions per read).  Out-of-image pixels are loaded with an out-of-range buffer
    // offset, which the hardware returns as zeros (no select per value).
    const bool ps_act = tid < HALO_PX;
    unsigned ps_off = 0u; int ps_row = 0, ps_swz = 0; bool ps_ok = false;
    if constexpr (PSL) {
        const int hp = ps_act ? tid : 0;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int iy = y0 + hy, ix = x0 + hx;
        ps_ok = ps_act && iy >= vy0 && iy < vy1 && ix >= 0 && ix < a.Win;
        const int iyc = iy < 0 ? 0 : (iy >= a.Hin ? a.Hin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= a.Win ? a.Win - 1 : ix);
        ps_off = ps_ok ? (unsigned)(iyc * a.Win + ixc) * EB : 0x7ffffff0u;      // (raw buffer: voffset >= num_records reads as 0)
        ps_row = hp * HPITCH;
        ps_swz = M16 ? ((hx >> 2) & 1) << 1 : (hx >> 2) & 3;
    }
    auto ps_load = [&](int cc, int t, float (&x)[4]) {      // cc, t: wave-uniform.  Tap 8 (and the last chunk, which passes its own
        // cc) re-loads data that is already on its way / in L2 instead of branching around the loads: a uniform condition here is
        // turned into scalar branches with one load form per path, and the waits of the tap loop stop being exact counts
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = cc * KCB + (t & 7) * 4 + j;
            c = c < a.C ? c : a.C - 1;                                 // (ragged last chunk: the channel meets zero weights)
            x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, ps_off, (unsigned)c * HWin * EB, 0));
        }
    };
    // per-channel factors of a slice (style x fp16 sample scale): requested at the START of the tap that writes the slice, half a tap
    // before they are used, so that the LDS read is long complete and its wait does not drain the fragment reads in flight
    auto ps_factors = [&](int cc, int t) -> float4 {
        return *reinterpret_cast<const float4*>(scl + cc * KCB + (t & 7) * 4);
    };
    auto ps_write = [&](unsigned char* buf, int t, const float (&x)[4], const float4 f) {      // the slice loaded with the same t
        if (t < 8 && ps_act) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            unsigned char* dst = buf + ps_row + ((((t >> 1) ^ ps_swz) & 3) << 4) + (t & 1) * 8;
            if constexpr (ABL) {
                const uint2 w = make_uint2(__builtin_bit_cast(unsigned, x[0]), __builtin_bit_cast(unsigned, x[2]));
                *reinterpret_cast<uint2*>(dst) = w;
                *reinterpret_cast<uint2*>(dst + HPLANE) = w;
                return;
            }
            // scalar fp32 arithmetic on purpose: v_pk_mul_f32 / v_pk_fma_f32 beside MFMAs cost ~20 cycles each (MI355X_MICROARCH.md,
            // 'price of one filler beside MFMAs'); only the two conversions are packed
            const float p0 = x[0] * f.x, p1 = x[1] * f.y, p2 = x[2] * f.z, p3 = x[3] * f.w;
            const f16x2 h0 = __builtin_convertvector(f32x2{p0, p1}, f16x2), h1 = __builtin_convertvector(f32x2{p2, p3}, f16x2);
            const float r0 = __builtin_fmaf(x[0], f.x, -(float)h0[0]), r1 = __builtin_fmaf(x[1], f.y, -(float)h0[1]);
            const float r2 = __builtin_fmaf(x[2], f.z, -(float)h1[0]), r3 = __builtin_fmaf(x[3], f.w, -(float)h1[1]);
            const f16x2 l0 = __builtin_convertvector(f32x2{r0, r1}, f16x2), l1 = __builtin_convertvector(f32x2{r2, r3}, f16x2);
            const uint2 wh = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
            const uint2 wl = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
            *reinterpret_cast<uint2*>(dst) = wh;
            *reinterpret_cast<uint2*>(dst + HPLANE) = wl;
        }
    };

    // ---- A fragments straight from the fragment-order pack (blocks past M are clamped: their rows are never stored)
    const int Mp = pack_mp(a.M);
    unsigned a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int mblk = (m0 + wm * (MT / WM_) + i * 32) >> 5;
        mblk = mblk < (Mp >> 5) ? mblk : (Mp >> 5) - 1;
        a_off[i] = (unsigned)mblk * 2048u + (unsigned)lane * 16u;
    }
    const unsigned slab_bytes = (unsigned)Mp * KCB * 2u;         // one (tap, chunk) slab of one term
    const unsigned term_bytes = (unsigned)term_elems * 2u;
    auto load_a = [&](int cc, int t, int ks, bf16x8 (&dst)[NTERM][TM]) {
        const unsigned tw = (unsigned)((wpack >> (4 * t)) & 15u);
        const unsigned so = (tw * nck + cc) * slab_bytes + ks * 1024;
#pragma unroll
        for (int q = 0; q < NTERM; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                dst[q][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_off[i], so + q * term_bytes, 0));
    };
    // B fragments: lane (l31, lh) of N-subtile j reads slot ks*2 + lh of halo pixel (wn*NJ + j) * 34 + shift + l31
    // PSL (column swizzle): byte offset of this lane's K-step 0 fragment of tile row 0 for a tap; K-step 1 = ^ 32, tile row j = + j * 34 * 64
    auto lane_b = [&](int shift, int dxp) -> int {
        return (shift + l31) * HPITCH + (((lh ^ ((l31 + dxp) >> 2)) & 3) << 4);
    };
    auto read_b = [&](const unsigned char* buf, int shift, int ks, bf16x8 (&dst)[NTERM][NJ], int lb = 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int o;
            if constexpr (PSL) o = (lb ^ (ks << 5)) + (wn * NJ + j) * HALO_W * HPITCH;
            else {
                const int p = (wn * NJ + j) * HALO_W + shift + l31;
                o = p * HPITCH + ((((ks * 2 + lh) ^ (p >> 2)) & 3) << 4);
            }
#pragma unroll
            for (int q = 0; q < NTERM; ++q) dst[q][j] = *reinterpret_cast<const bf16x8*>(buf + q * HPLANE + o);
        }
    };

    f32x16 acc[TM][NJ];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma_step = [&](bf16x8 (&af)[NTERM][TM], bf16x8 (&bf)[NTERM][NJ]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // smallest products first, so they are not swamped by the leading term inside the accumulator
                if constexpr (NTERM == 3) {
                    acc[i][j] = la_mma<F16>(af[2][i], bf[0][j], acc[i][j]);   // lh
                    acc[i][j] = la_mma<F16>(af[0][i], bf[2][j], acc[i][j]);   // hl
                    acc[i][j] = la_mma<F16>(af[1][i], bf[1][j], acc[i][j]);   // mm
                }
                acc[i][j] = la_mma<F16>(af[1][i], bf[0][j], acc[i][j]);   // mh
                acc[i][j] = la_mma<F16>(af[0][i], bf[1][j], acc[i][j]);   // hm
                acc[i][j] = la_mma<F16>(af[0][i], bf[0][j], acc[i][j]);   // hh
            }
    };

    // ---- prologue: chunk 0's halo (all nine slices in flight at once) and the first tap's weights
    bf16x8 acur[2][NTERM][TM];
    LA_STAMP_DECL
    if constexpr (PSL) {
        float pre[8][4];
#pragma unroll
        for (int t = 0; t < 8; ++t) ps_load(0, t, pre[t]);
        if constexpr (!M16) { load_a(0, 0, 0, acur[0]); load_a(0, 0, 1, acur[1]); }
        {
            const float xs = F16 ? la_xs_get(a.acc_scale_x, b, a.acc_scale_fan) : 1.f;
            for (int k = tid; k < nck * KCB; k += 256)
                scl[k] = k < a.C ? (a.in_scale ? a.in_scale[(long)b * a.scale_stride + k] : 1.f) * xs : 0.f;
        }
        LA_STAMP(0);
        __syncthreads();                           // scl is complete before any slice is scaled with it
#pragma unroll
        for (int t = 0; t < 8; ++t) ps_write(smem, t, pre[t], ps_factors(0, t));
    } else {
        Slice pre[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) slice_load(0, t, pre[t]);
        load_a(0, 0, 0, acur[0]);
        load_a(0, 0, 1, acur[1]);
        {      // per-channel factors: filled after the halo / weight loads were issued, so that the latencies overlap
            const float xs = F16 ? la_xs_get(a.acc_scale_x, b, a.acc_scale_fan) : 1.f;
            for (int k = tid; k < nck * KCB; k += 256)
                scl[k] = k < a.C ? (a.in_scale ? a.in_scale[(long)b * a.scale_stride + k] : 1.f) * xs : 0.f;
        }
        __syncthreads();                           // scl is complete before any slice is scaled with it
#pragma unroll
        for (int t = 0; t < 9; ++t) slice_write(smem, pre[t]);
    }
    __syncthreads();
    LA_STAMP(1);

  if constexpr (M16) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int c16 = lane & 15, kq = lane >> 4;
    f32x4 acc16[2][8];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc16[mi][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight fragments of a 16-row half: lane (c16, kq) holds rows mi*16 + c16, channels 8 kq .. 8 kq + 7 of the chunk
    int mblk16 = (m0 + wm * 32) >> 5;
    mblk16 = mblk16 < (Mp >> 5) ? mblk16 : (Mp >> 5) - 1;
    const unsigned a16_off = (unsigned)mblk16 * 2048u + (unsigned)(kq * 32 + c16) * 16u;
    auto load_a16 = [&](int cc, int t, int mi, f16x8 (&dst)[2]) {
        const unsigned tw = (unsigned)((wpack >> (4 * t)) & 15u);
        const unsigned so = (tw * nck + cc) * slab_bytes;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            dst[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a16_off + (unsigned)mi * 256u, so + q * term_bytes, 0));
    };
    // pixel fragment of tile n (tile row n >> 1, x half n & 1): lane (c16, kq) reads slot kq of its pixel's 64-byte row
    // (PSL, column swizzle: lb16 = this lane's offset for the tap, the tile adds an immediate)
    auto lane_b16 = [&](int shift, int dxp) -> int {
        return (shift + c16) * HPITCH + (((kq ^ ((((c16 + dxp) >> 2) & 1) << 1)) & 3) << 4);
    };
    auto read_b16 = [&](const unsigned char* buf, int shift, int n, f16x8 (&dst)[2], int lb16 = 0) {
        int o;
        if constexpr (PSL) o = lb16 + ((n >> 1) * HALO_W + (n & 1) * 16) * HPITCH;
        else {
            const int p = (n >> 1) * HALO_W + shift + (n & 1) * 16 + c16;
            o = p * HPITCH + (((kq ^ (((p >> 2) & 1) << 1)) & 3) << 4);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) dst[q] = *reinterpret_cast<const f16x8*>(buf + q * HPLANE + o);
    };
    f16x8 a16[2][2], b16[4][2];
    // (issue order pinned: the tap loop's first wait is counted for "everything but the two youngest loads" on both of its entries)
    __builtin_amdgcn_sched_barrier(0);
    load_a16(0, 0, 0, a16[0]);
    __builtin_amdgcn_sched_barrier(0);
    load_a16(0, 0, 1, a16[1]);
    __builtin_amdgcn_sched_barrier(0);
    for (int cc = 0; cc < nck; ++cc) {
        const unsigned char* cur = smem + (cc & 1) * HBUF;
        unsigned char* nxt = smem + ((cc + 1) & 1) * HBUF;
        const bool has_next = cc + 1 < nck;
        Slice sl;
        sl.wr = -1;
        sl.ok = false;
        float psx[4] = {0.f, 0.f, 0.f, 0.f};
        int lb_cur = PSL ? lane_b16((int)(shpack & 127u), (int)(xpack & 3u)) : 0;
#pragma unroll
        for (int n = 0; n < 4; ++n) read_b16(cur, (int)(shpack & 127u), n, b16[n], lb_cur);
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int shift = (int)((shpack >> (7 * t)) & 127u);
            const int tn = t + 1 < 9 ? t + 1 : 0;
            const int ccn = t + 1 < 9 ? cc : (has_next ? cc + 1 : 0);
            const int shift_n = (int)((shpack >> (7 * (t + 1 < 9 ? t + 1 : 8))) & 127u);
            const int lb_nxt = PSL ? lane_b16(shift_n, (int)((xpack >> (2 * (t + 1 < 9 ? t + 1 : 8))) & 3u)) : 0;
            float4 psf = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PSL) { if (has_next && t >= 1) psf = ps_factors(cc + 1, t - 1); }
            if constexpr (FR1) {
                // every pixel fragment read ONCE per tap: four quarters (tiles 0-3 x rows 0-15, tiles 0-3 x rows 16-31, tiles 4-7 x rows 0-15,
                // tiles 4-7 x rows 16-31); a slot is re-filled after its second use with the tile four steps ahead (tile k + 4 of this tap, then
                // tile k of the next tap).  Half the LDS fragment traffic of the order below (the LDS pipe of a CU is busy 54-90 % under three
                // workgroups, profiles/r05_pmc_halo_waits.txt) -- at the price of a QUARTER tap instead of half a tap for the next weights to land
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int hf = q4 >> 1, mi = q4 & 1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int n = hf * 4 + k;
                        f16x8 (&bs)[2] = b16[k];
                        acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][1], bs[0], acc16[mi][n], 0, 0, 0);   // lh
                        acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][0], bs[1], acc16[mi][n], 0, 0, 0);   // hl
                        acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][0], bs[0], acc16[mi][n], 0, 0, 0);   // hh
                        if (mi == 1) {
                            if (hf == 0) read_b16(cur, shift, 4 + k, bs, lb_cur);
                            else if (t + 1 < 9) read_b16(cur, shift_n, k, bs, lb_nxt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (q4 == 2) load_a16(ccn, tn, 0, a16[0]);      // rows 0-15 are done with this tap's weights: the next tap's, a quarter tap to land
                    if (q4 == 3) load_a16(ccn, tn, 1, a16[1]);
                    if (q4 == 1) {
                        if (has_next && t >= 1) ps_write(nxt, t - 1, psx, psf);      // the slice loaded one tap ago
                        ps_load(has_next ? cc + 1 : cc, t, psx);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    f16x8 (&bs)[2] = b16[n & 3];
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][1], bs[0], acc16[mi][n], 0, 0, 0);   // lh
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][0], bs[1], acc16[mi][n], 0, 0, 0);   // hl
                    acc16[mi][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mi][0], bs[0], acc16[mi][n], 0, 0, 0);   // hh
                    // the slot is re-filled with the tile four steps ahead: (mi, n + 4), then (mi + 1, n - 4) of this tap (the same
                    // pixels again for the other 16 rows), then tile n - 4 of the next tap (never across the chunk barrier)
                    if (n < 4) read_b16(cur, shift, n + 4, bs, lb_cur);
                    else if (mi == 0) read_b16(cur, shift, n - 4, bs, lb_cur);
                    else if (t + 1 < 9) read_b16(cur, shift_n, n - 4, bs, lb_nxt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                load_a16(ccn, tn, mi, a16[mi]);          // this half's weights of the next tap: half a tap to land
                if (mi == 0) {
                    if constexpr (PSL) {
                        if (has_next && t >= 1) ps_write(nxt, t - 1, psx, psf);      // the slice loaded one tap ago
                        ps_load(has_next ? cc + 1 : cc, t, psx);
                    } else {
                        slice_write(nxt, sl);
                        slice_load(has_next ? cc + 1 : cc, t, sl, has_next);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
            lb_cur = lb_nxt;
        }
        if constexpr (!PSL) slice_write(nxt, sl);      // (PSL: tap 8 loads nothing)
        LA_STAMP(2);
        __syncthreads();
        LA_STAMP(3);
    }
    // 16x16 tiles -> the 32x32 accumulator layout of the epilogue, through LDS (free after the loop's last barrier), two tile rows
    // at a time: image [wave][64 pixels][36 floats] (32 rows + 4 of padding), 16-byte writes and reads
    {
        const float inv = 1.f / (a.acc_scale_w[0] * la_xs_get(a.acc_scale_x, b, a.acc_scale_fan));
        float* tb = reinterpret_cast<float*>(smem) + wid * (64 * 36);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (hf) __syncthreads();
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) {
                    const int n = hf * 4 + n4;
                    const int px = (n4 >> 1) * 32 + (n & 1) * 16 + c16;
                    // (element by element: a vector multiply here is two v_pk_mul_f32 -- no packed-FP32 arithmetic anywhere, Makefile)
                    const f32x4 v = acc16[mi][n];
                    *reinterpret_cast<f32x4*>(tb + px * 36 + mi * 16 + kq * 4) = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
                }
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (jj * 32 + l31) * 36 + 8 * g + 4 * lh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[0][hf * 2 + jj][4 * g + r] = v[r];
                }
        }
        __syncthreads();
    }
    LA_STAMP(4);
    la_conv_epilogue<MT, false, true, WM_>(a, acc, red, ntile, m0, G, G LA_STAMP_ARG);
    LA_STAMP(5);
    LA_STAMP_OUT;
    return;
  } else
  if constexpr (SB) {
    // single-buffer form: bf holds the fragments of ONE K-step; sub-tile j is re-loaded for the following K-step right after its own
    // three MFMAs have issued, so its LDS latency runs under the MFMAs of the other sub-tiles and no second fragment set is live
    bf16x8 bf[NTERM][NJ];
    auto mma_refill = [&](bf16x8 (&af)[NTERM][TM], const unsigned char* buf, int shift, int ks, bool refill, int lb = 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (NTERM == 3) {
                acc[0][j] = la_mma<F16>(af[2][0], bf[0][j], acc[0][j]);
                acc[0][j] = la_mma<F16>(af[0][0], bf[2][j], acc[0][j]);
                acc[0][j] = la_mma<F16>(af[1][0], bf[1][j], acc[0][j]);
            }
            acc[0][j] = la_mma<F16>(af[1][0], bf[0][j], acc[0][j]);
            acc[0][j] = la_mma<F16>(af[0][0], bf[1][j], acc[0][j]);
            acc[0][j] = la_mma<F16>(af[0][0], bf[0][j], acc[0][j]);
            if (refill) {
                int o;
                if constexpr (PSL) o = (lb ^ (ks << 5)) + (wn * NJ + j) * HALO_W * HPITCH;
                else {
                    const int p = (wn * NJ + j) * HALO_W + shift + l31;
                    o = p * HPITCH + ((((ks * 2 + lh) ^ (p >> 2)) & 3) << 4);
                }
#pragma unroll
                for (int q = 0; q < NTERM; ++q) bf[q][j] = *reinterpret_cast<const bf16x8*>(buf + q * HPLANE + o);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int cc = 0; cc < nck; ++cc) {
        const unsigned char* cur = smem + (cc & 1) * HBUF;
        unsigned char* nxt = smem + ((cc + 1) & 1) * HBUF;
        const bool has_next = cc + 1 < nck;
        Slice sl;
        sl.wr = -1;
        sl.ok = false;
        float psx[4] = {0.f, 0.f, 0.f, 0.f};
        int lb = PSL ? lane_b((int)(shpack & 127u), (int)(xpack & 3u)) : 0;      // lane address of the current tap (PSL)
        read_b(cur, (int)(shpack & 127u), 0, bf, lb);
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int shift = (int)((shpack >> (7 * t)) & 127u);
            const int tn = t + 1 < 9 ? t + 1 : 0;
            const int ccn = t + 1 < 9 ? cc : (has_next ? cc + 1 : 0);
            const int shift_n = (int)((shpack >> (7 * (t + 1 < 9 ? t + 1 : 8))) & 127u);
            float4 psf = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PSL) { if (has_next && t >= 1) psf = ps_factors(cc + 1, t - 1); }
            mma_refill(acur[0], cur, shift, 1, true, lb);      // K-step 0; refilled with this tap's K-step 1
            if constexpr (PSL) lb = lane_b(shift_n, (int)((xpack >> (2 * (t + 1 < 9 ? t + 1 : 8))) & 3u));
            load_a(ccn, tn, 0, acur[0]);
            if constexpr (PSL) {
                if (has_next && t >= 1) ps_write(nxt, t - 1, psx, psf);      // the slice loaded one tap ago
                ps_load(has_next ? cc + 1 : cc, t, psx);
            } else {
                slice_write(nxt, sl);
                slice_load(has_next ? cc + 1 : cc, t, sl, has_next);      // (last chunk: dummy loads, nothing staged)
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_refill(acur[1], cur, shift_n, 0, t + 1 < 9, lb);   // K-step 1; refilled with the next tap's K-step 0 (not across the barrier)
            load_a(ccn, tn, 1, acur[1]);
        }
        if constexpr (!PSL) slice_write(nxt, sl);      // (PSL: tap 8 loads nothing)
        __syncthreads();       // next halo complete, everyone done with this one (and, at the end, LDS free for the epilogue)
    }
  } else {
    bf16x8 bf0[NTERM][NJ], bf1[NTERM][NJ];
    auto chunk = [&](int cc, auto has_next) {
        constexpr bool NEXT = decltype(has_next)::value;
        const unsigned char* cur = smem + (cc & 1) * HBUF;
        unsigned char* nxt = smem + ((cc + 1) & 1) * HBUF;
        Slice sl;
        sl.wr = -1;
        sl.ok = false;
        float psx[4] = {0.f, 0.f, 0.f, 0.f};
        int lb = PSL ? lane_b((int)(shpack & 127u), (int)(xpack & 3u)) : 0;      // lane address of the current tap (PSL)
        read_b(cur, (int)(shpack & 127u), 0, bf0, lb);
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int shift = (int)((shpack >> (7 * t)) & 127u);
            // the tap after this one: next tap, else first tap of the next chunk (the final one re-reads a valid slab)
            const int tn = t + 1 < 9 ? t + 1 : 0;
            const int ccn = t + 1 < 9 ? cc : (NEXT ? cc + 1 : 0);
            const int shift_n = (int)((shpack >> (7 * (t + 1 < 9 ? t + 1 : 8))) & 127u);   // (last tap: harmless re-read)
            // (the fences pin the issue order: left alone, the scheduler sinks every load to just before its first use)
            float4 psf = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PSL) { if (NEXT && t >= 1) psf = ps_factors(cc + 1, t - 1); }
            read_b(cur, shift, 1, bf1, lb);        // B of K-step 1 flies under the MFMAs of K-step 0
            mma_step(acur[0], bf0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(ccn, tn, 0, acur[0]);           // re-loaded as soon as its MFMAs have issued
            if constexpr (PSL) lb = lane_b(shift_n, (int)((xpack >> (2 * (t + 1 < 9 ? t + 1 : 8))) & 3u));
            read_b(cur, shift_n, 0, bf0, lb);      // B of the next tap's K-step 0
            if (NEXT) {
                if constexpr (PSL) {
                    if (t >= 1) ps_write(nxt, t - 1, psx, psf);      // the slice loaded one tap ago
                    ps_load(cc + 1, t, psx);
                } else {
                    slice_write(nxt, sl);
                    slice_load(cc + 1, t, sl);
                }
            }
            mma_step(acur[1], bf1);
            __builtin_amdgcn_sched_barrier(0);
            load_a(ccn, tn, 1, acur[1]);
        }
        if (NEXT && !PSL) slice_write(nxt, sl);      // (PSL: tap 8 loads nothing new)
    };
    for (int cc = 0; cc < nck; ++cc) {
        if (cc + 1 < nck) chunk(cc, std::true_type{});
        else chunk(cc, std::false_type{});
        __syncthreads();       // next halo complete, everyone done with this one (and, at the end, LDS free for the epilogue)
    }
  }
    if (F16) {
        const float inv = 1.f / (a.acc_scale_w[0] * la_xs_get(a.acc_scale_x, b, a.acc_scale_fan));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
    }
    la_conv_epilogue<MT, false, true, WM_>(a, acc, red, ntile, m0, G, G);
}

template <int FMT>
static int launch_bf16(const LaConvArgs& as, int MTsel, dim3 grid, bool split, hipStream_t stream) {
    constexpr int NTERM = FMT == FMT_BF16X3 ? 3 : 2;
    constexpr int W3 = NTERM == 2 ? 3 : 2;      // (the three-term format does not fit three waves: its W3 alias is the plain kernel)
    const size_t lds128 = (size_t)2 * NTERM * NT * BPITCH, lds64 = lds128;     // two pixel buffers (>= the epilogue's 4 * MT floats)
    if (!split && la_conv_bf16_uses_halo(as)) {
        // two halo buffers (>= the epilogue's 4 * MT floats) + the per-channel factor table
        size_t h128 = (size_t)(as.C > KCB ? 2 : 1) * NTERM * HALO_PX * HPITCH + (size_t)la_cdiv(as.C, KCB) * KCB * sizeof(float);
        const size_t epi = (size_t)160 * MTsel + (size_t)2048 * (MTsel / 32);      // what the epilogue addresses (row tables + fused-ToRGB partials)
        if (h128 < epi) h128 = epi;
        const size_t h64 = h128;
        // > 64 KB of dynamic LDS needs the opt-in, and the attribute is per DEVICE: track it per device (atomic flags: the
        // entry points may be entered from several host threads, one per device)
        static std::atomic<bool> attr_done[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_done[dev].load(std::memory_order_acquire)) {
            const int cap = 2 * 3 * HALO_PX * HPITCH + 4096 * (int)sizeof(float);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&la_conv_bf16_halo_kernel<128, FMT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&la_conv_bf16_halo_kernel<128, FMT, W3>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&la_conv_bf16_halo_kernel<64, FMT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&la_conv_bf16_halo_kernel<32, FMT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e != hipSuccess) { la_set_error(hipGetErrorString(e)); return LA_ERR_HIP; }
            attr_done[dev].store(true, std::memory_order_release);
        }
        // three-wave form for the fp16 x2 launches on 128-row tiles (see the kernel comment); dev knob LA_HALO_W3=0|1
        static const int w3_knob = []() { const char* e = la_dev_env("LA_HALO_W3"); return e ? atoi(e) : -1; }();
        const bool w3 = NTERM == 2 && MTsel == 128 && (w3_knob >= 0 ? w3_knob != 0 : true);
        if constexpr (FMT == FMT_F16X2) {
            // fp16 x2 launches on 128-row tiles with more than one chunk: the 16x16x32 / pixel-stationary form (MF 5, kernel comment).
            // Dev knob (in-process A/B): 0 = this default, 8 = the round-2 form (MF 0), otherwise the MF bits to run.
            const int knob = la_dev_knob(LA_KNOB_HALO_MF);
            const int mf = knob == 0 ? 21 : (knob == 8 ? 0 : knob);      // (21 = MF 5 with every pixel fragment read once per tap, round 5)
            if (MTsel == 128 && w3 && as.C > KCB && mf > 0) {
                auto go = [&](auto tag) -> int {
                    constexpr int MFV = decltype(tag)::value;
                    static std::atomic<bool> done[64];
                    if (!done[dev].load(std::memory_order_acquire)) {
                        #ifdef LA_DEV
                        const int cap = 160 * 1024;      // (room for LA_KNOB_HALO_LDSPAD)
#else
                        const int cap = 2 * 3 * HALO_PX * HPITCH + 4096 * (int)sizeof(float);
#endif
                        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&la_conv_bf16_halo_kernel<128, FMT_F16X2, 3, MFV>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) {
                            la_set_error("halo MF: hipFuncSetAttribute failed"); return LA_ERR_HIP;
                        }
                        done[dev].store(true, std::memory_order_release);
                    }
                    // (dev knob LA_KNOB_HALO_LDSPAD: extra KB of dynamic LDS per workgroup -- fewer workgroups per CU, for scripts/halo_wave_timeline.py)
                    hipLaunchKernelGGL((la_conv_bf16_halo_kernel<128, FMT_F16X2, 3, MFV>), grid, dim3(256), h128 + (size_t)la_dev_knob(LA_KNOB_HALO_LDSPAD) * 1024, stream, as);
                    return LA_OK;
                };
                switch (mf) {
#ifdef LA_DEV
                    case 1: return go(std::integral_constant<int, 1>{});
                    case 4: return go(std::integral_constant<int, 4>{});
                    case 7: return go(std::integral_constant<int, 7>{});      // (loader ablation: wrong results, timing only)
                    case 5: return go(std::integral_constant<int, 5>{});      // (round 4: every pixel fragment read twice per tap)
#endif
                    case 21: return go(std::integral_constant<int, 21>{});
                    default: break;
                }
            }
        }
        if constexpr (FMT == FMT_F16X2) {
            // every other fp16 x2 halo launch (64- / 32-row tiles, single-chunk launches): the pixel-stationary loader on the 32x32x16 form
            // (MF 4); dev knob 8 = the round-2 loader
            if (la_dev_knob(LA_KNOB_HALO_MF) != 8) {
                auto attr = [&](const void* fn, std::atomic<bool>& flag) -> bool {
                    if (!flag.load(std::memory_order_acquire)) {
                        const int cap = 2 * 3 * HALO_PX * HPITCH + 4096 * (int)sizeof(float);
                        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) { la_set_error("halo PSL: hipFuncSetAttribute failed"); return false; }
                        flag.store(true, std::memory_order_release);
                    }
                    return true;
                };
                static std::atomic<bool> d128[64], d64[64], d32[64];
                if (MTsel == 128 && w3) {
                    if (!attr(reinterpret_cast<const void*>(&la_conv_bf16_halo_kernel<128, FMT_F16X2, 3, 4>), d128[dev])) return LA_ERR_HIP;
                    hipLaunchKernelGGL((la_conv_bf16_halo_kernel<128, FMT_F16X2, 3, 4>), grid, dim3(256), h128, stream, as);
                    return LA_OK;
                }
                if (MTsel == 64) {
                    if (!attr(reinterpret_cast<const void*>(&la_conv_bf16_halo_kernel<64, FMT_F16X2, 2, 4>), d64[dev])) return LA_ERR_HIP;
                    hipLaunchKernelGGL((la_conv_bf16_halo_kernel<64, FMT_F16X2, 2, 4>), grid, dim3(256), h64, stream, as);
                    return LA_OK;
                }
                if (MTsel == 32) {
                    if (!attr(reinterpret_cast<const void*>(&la_conv_bf16_halo_kernel<32, FMT_F16X2, 2, 4>), d32[dev])) return LA_ERR_HIP;
                    hipLaunchKernelGGL((la_conv_bf16_halo_kernel<32, FMT_F16X2, 2, 4>), grid, dim3(256), h64, stream, as);
                    return LA_OK;
                }
            }
        }
        if (MTsel == 128 && w3) hipLaunchKernelGGL((la_conv_bf16_halo_kernel<128, FMT, W3>), grid, dim3(256), h128, stream, as);
        else if (MTsel == 128) hipLaunchKernelGGL((la_conv_bf16_halo_kernel<128, FMT, 2>), grid, dim3(256), h128, stream, as);
        else if (MTsel == 64) hipLaunchKernelGGL((la_conv_bf16_halo_kernel<64, FMT, 2>), grid, dim3(256), h64, stream, as);
        else hipLaunchKernelGGL((la_conv_bf16_halo_kernel<32, FMT, 2>), grid, dim3(256), h64, stream, as);
        return LA_OK;
    }
    // three-wave form of the 128-row fp16 x2 launches (kernel comment); dev knob LA_FLAT_W3=0|1
    static const int f3_knob = []() { const char* e = la_dev_env("LA_FLAT_W3"); return e ? atoi(e) : -1; }();
    const bool f3 = FMT == FMT_F16X2  && MTsel == 128 && (f3_knob >= 0 ? f3_knob != 0 : true);
    constexpr int FW3 = (FMT == FMT_F16X2 ) ? 3 : 2;
    if constexpr (FMT == FMT_F16X2) {
        // 16x16x32 form of the three-wave kernel (dev knob LA_KNOB_FLAT_MF: 8 = the 32x32x16 form); its accumulator hand-over needs 36 KB of LDS
        const int fk = la_dev_knob(LA_KNOB_FLAT_MF);
        if (MTsel == 128 && f3 && fk != 8) {
            const size_t lds_mf = lds128 > (size_t)4 * 64 * 36 * 4 ? lds128 : (size_t)4 * 64 * 36 * 4;
            const size_t lds_3 = (size_t)3 * NTERM * NT * BPITCH;      // three pixel buffers (>= the accumulator hand-over's 36 KB)
            const bool three = fk == 2 || (fk == 0 && !split) || fk == 3;      // default: direct launches on three buffers (knob 2: both, 1: neither)
#ifdef LA_DEV
            if (fk >= 16 && !split) {      // ablations of the direct launches (kernel comment): 18 = no pixel re-reads, 34 = no weight re-reads, 50 = neither
                if (fk == 18) hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT_F16X2, 3, 18>), grid, dim3(256), lds_3, stream, as);
                else if (fk == 66) hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT_F16X2, 3, 66>), grid, dim3(256), lds_3, stream, as);       // no barrier
                else if (fk == 130) hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT_F16X2, 3, 130>), grid, dim3(256), lds_3, stream, as);     // no LDS write
                else if (fk == 242) hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT_F16X2, 3, 242>), grid, dim3(256), lds_3, stream, as);     // none of the four
                else if (fk == 34) hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT_F16X2, 3, 34>), grid, dim3(256), lds_3, stream, as);
                else hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT_F16X2, 3, 50>), grid, dim3(256), lds_3, stream, as);
                return LA_OK;
            }
#endif
            if (three && !split) hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT_F16X2, 3, 2>), grid, dim3(256), lds_3, stream, as);
            else if (fk == 2 && split) hipLaunchKernelGGL((la_conv_bf16_kernel<128, true, FMT_F16X2, 3, 2>), grid, dim3(256), lds_3, stream, as);
#ifdef LA_DEV
            else if (split && (fk == 11 || fk == 12)) {
                // round-5 experiment, measured and NOT faster (profiles/r05_exp_splitk_prefetch.txt): weights three steps ahead (MF 3, two
                // waves per SIMD; knob 11), and on top of it launches of at most one workgroup per CU whose slices are whole 9-tap chunks with a
                // slice's first nine steps of weights requested up front, at one wave per SIMD (MF 4; knob 12).  Bit-identical to MF 1.
                const long wgs = (long)grid.x * grid.y * grid.z;
                const bool whole9 = as.nphase == 0 && as.ntaps == 9;
                if (fk == 12 && wgs <= 256 && whole9) hipLaunchKernelGGL((la_conv_bf16_kernel<128, true, FMT_F16X2, 1, 4>), grid, dim3(256), lds_mf, stream, as);
                else hipLaunchKernelGGL((la_conv_bf16_kernel<128, true, FMT_F16X2, 2, 3>), grid, dim3(256), lds_mf, stream, as);
            }
#endif
            else if (split) hipLaunchKernelGGL((la_conv_bf16_kernel<128, true, FMT_F16X2, 3, 1>), grid, dim3(256), lds_mf, stream, as);
            else hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT_F16X2, 3, 1>), grid, dim3(256), lds_mf, stream, as);
            return LA_OK;
        }
    }
    if (MTsel == 128 && f3) {
        if (split) hipLaunchKernelGGL((la_conv_bf16_kernel<128, true, FMT, FW3>), grid, dim3(256), lds128, stream, as);
        else hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT, FW3>), grid, dim3(256), lds128, stream, as);
    } else if (MTsel == 128) {
        if (split) hipLaunchKernelGGL((la_conv_bf16_kernel<128, true, FMT, 2>), grid, dim3(256), lds128, stream, as);
        else hipLaunchKernelGGL((la_conv_bf16_kernel<128, false, FMT, 2>), grid, dim3(256), lds128, stream, as);
    } else {
        if (split) hipLaunchKernelGGL((la_conv_bf16_kernel<64, true, FMT, 2>), grid, dim3(256), lds64, stream, as);
        else hipLaunchKernelGGL((la_conv_bf16_kernel<64, false, FMT, 2>), grid, dim3(256), lds64, stream, as);
    }
    return LA_OK;
}

int la_conv_bf16_dispatch(const LaConvArgs& args, int MTsel, dim3 grid, bool split, hipStream_t stream) {
    LaConvArgs as = args;
    if (as.precision == LA_PREC_F16X2)      // the fp16 weight scale lives behind the terms of the pack
        as.acc_scale_w = reinterpret_cast<const float*>(static_cast<const char*>(as.wgt_bf16) + pack_wscale_offset(as.wgt_bf16_term_elems));
    if (as.precision == LA_PREC_BF16X3) return launch_bf16<FMT_BF16X3>(as, MTsel, grid, split, stream);
    if (as.precision == LA_PREC_F16X2) return launch_bf16<FMT_F16X2>(as, MTsel, grid, split, stream);
    return launch_bf16<FMT_BF16X2>(as, MTsel, grid, split, stream);
}
