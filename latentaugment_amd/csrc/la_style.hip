// Small kernels around the modulated convolutions of the SG2 synthesis pass:
//   weight packing (once per generator), affine (style) FC forward/backward over all layers in one launch,
//   demodulation coefficients, ToRGB forward (+skip add), the backward "seam" kernel that fuses
//   ToRGB-backward + bias_act-backward + demod-gradient reductions, and the style-gradient finish.
// All are HBM- or latency-bound; the contraction work lives in la_conv.hip.
#include "la_style.h"
#include "la_conv.h"

// ------------------------------------------------------------------------------------------------------------
// weight packing:  W[o][i][ky][kx]  ->  wf[t][i][o], wb[t][o][i], wsq[o][i] = sum_t W^2      (t = ky*3+kx)
__global__ void la_pack_conv_kernel(const float* __restrict__ w, float* wf, float* wb, float* wsq, int cout, int cin,
                                    int ktaps, float scale, int wb_ld) {
    const long n = (long)cout * cin;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int o = (int)(idx / cin), i = (int)(idx - (long)o * cin);
        float sq = 0.f;
        for (int t = 0; t < ktaps; ++t) {
            const float v = w[idx * ktaps + t] * scale;
            sq += v * v;
            if (wf) wf[((long)t * cin + i) * cout + o] = v;
            if (wb) wb[((long)t * cout + o) * wb_ld + i] = v;
        }
        if (wsq) wsq[idx] = sq;
    }
}

int la_pack_conv_weights(const float* w, float* wf, float* wb, float* wsq, int cout, int cin, int ktaps,
                         hipStream_t stream, float scale, int wb_ld) {
    LA_CHECK_ARG(w && cout > 0 && cin > 0 && ktaps > 0, "pack: bad args");
    if (wb_ld <= 0) wb_ld = cin;
    if (wb && wb_ld != cin) LA_HIP(hipMemsetAsync(wb, 0, sizeof(float) * (size_t)ktaps * cout * wb_ld, stream));   // zero the pad columns
    const long n = (long)cout * cin;
    hipLaunchKernelGGL(la_pack_conv_kernel, dim3(la_cdiv(n, 256) < 4096 ? la_cdiv(n, 256) : 4096), dim3(256), 0, stream,
                       w, wf, wb, wsq, cout, cin, ktaps, scale, wb_ld);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// affine forward: s[b][row] = (dot(ws[b][widx_l], A_l[i]) * wgain + ab_l[i]) * post_gain_l
// One wave per 4 consecutive rows (never straddling a layer: channel counts are multiples of 4), 16-byte loads: per 64-lane
// step 4 row loads + BCH latent loads in flight (the first version: one dword row load per step, 37 us for 20 MB of rows).
#define BCH 8
__global__ __launch_bounds__(256) void la_affine_fwd_kernel(LaStyleTable t, const float* __restrict__ ws,
                                                           long ws_bstride, long ws_lstride, int B, int wdim,
                                                           float wgain, float* __restrict__ s_all) {
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    const int lane = threadIdx.x & 63;
    if (row >= t.total_rows) return;
    int l = 0;
    while (l + 1 < t.nlayers && row >= t.row_start[l + 1]) ++l;
    const int i = row - t.row_start[l];
    const float* arow = t.aw[l] + (long)i * wdim;
    const int w4 = wdim >> 2;
    for (int b0 = 0; b0 < B; b0 += BCH) {
        float acc[4][BCH];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < BCH; ++q) acc[r][q] = 0.f;
        for (int j = lane; j < w4; j += 64) {
            float4 av[4], wv[BCH];
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] = reinterpret_cast<const float4*>(arow + (long)r * wdim)[j];
#pragma unroll
            for (int q = 0; q < BCH; ++q)
                wv[q] = b0 + q < B ? reinterpret_cast<const float4*>(ws + (long)(b0 + q) * ws_bstride + (long)t.widx[l] * ws_lstride)[j]
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < BCH; ++q)
                    acc[r][q] += av[r].x * wv[q].x + av[r].y * wv[q].y + av[r].z * wv[q].z + av[r].w * wv[q].w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ab = t.ab[l][i + r];
#pragma unroll
            for (int q = 0; q < BCH; ++q) {
                const float v = la_wave_sum(acc[r][q]);
                if (lane == 0 && b0 + q < B) s_all[(long)(b0 + q) * t.total_rows + row + r] = (v * wgain + ab) * t.post_gain[l];
            }
        }
    }
}

int la_affine_forward(const LaStyleTable& t, const float* ws, long ws_bstride, long ws_lstride, int B, int wdim,
                      float* s_all, hipStream_t stream) {
    LA_CHECK_ARG(wdim % 4 == 0 && ws_bstride % 4 == 0 && ws_lstride % 4 == 0 && ((size_t)ws & 15) == 0 && t.total_rows % 4 == 0,
                 "affine_forward: w_dim / latent strides must be multiples of 4 floats (16-byte rows)");
    hipLaunchKernelGGL(la_affine_fwd_kernel, dim3(la_cdiv(t.total_rows, 16)), dim3(256), 0, stream, t, ws, ws_bstride,
                       ws_lstride, B, wdim, 1.0f / sqrtf((float)wdim), s_all);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// demod: d[b][doff_l + o] = rsqrt(sum_i s[b][soff_l+i]^2 * wsq_l[o][i] + 1e-8)     one wave per 4 rows (l, o .. o+3), 16-byte loads
__global__ __launch_bounds__(256) void la_demod_kernel(LaDemodTable t, const float* __restrict__ s_all, int s_stride,
                                                      int B, float* __restrict__ d_all) {
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    const int lane = threadIdx.x & 63;
    if (row >= t.total_rows) return;
    int l = 0;
    while (l + 1 < t.nlayers && row >= t.row_start[l + 1]) ++l;
    const int o = row - t.row_start[l];
    const int cin = t.cin[l], c4 = cin >> 2;
    const float* wrow = t.wsq[l] + (long)o * cin;
    for (int b0 = 0; b0 < B; b0 += BCH) {
        float acc[4][BCH];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < BCH; ++q) acc[r][q] = 0.f;
        for (int i = lane; i < c4; i += 64) {
            float4 wv[4], sv[BCH];
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[r] = reinterpret_cast<const float4*>(wrow + (long)r * cin)[i];
#pragma unroll
            for (int q = 0; q < BCH; ++q) {
                sv[q] = b0 + q < B ? reinterpret_cast<const float4*>(s_all + (long)(b0 + q) * s_stride + t.s_off[l])[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                sv[q].x *= sv[q].x; sv[q].y *= sv[q].y; sv[q].z *= sv[q].z; sv[q].w *= sv[q].w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < BCH; ++q)
                    acc[r][q] += sv[q].x * wv[r].x + sv[q].y * wv[r].y + sv[q].z * wv[r].z + sv[q].w * wv[r].w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < BCH; ++q) {
                const float v = la_wave_sum(acc[r][q]);
                if (lane == 0 && b0 + q < B) d_all[(long)(b0 + q) * t.total_rows + row + r] = rsqrtf(v + 1e-8f);
            }
    }
}

int la_demod_forward(const LaDemodTable& t, const float* s_all, int s_stride, int B, float* d_all,
                     hipStream_t stream) {
    LA_CHECK_ARG(t.total_rows % 4 == 0 && s_stride % 4 == 0, "demod_forward: channel counts must be multiples of 4");
    hipLaunchKernelGGL(la_demod_kernel, dim3(la_cdiv(t.total_rows, 16)), dim3(256), 0, stream, t, s_all, s_stride, B,
                       d_all);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Start of a pass (fp16 x2 mode), one launch: the slot rows of every conv layer's fp16 operand scales are reset -- forward rows
// xs_fwd [l][b][LA_XS_FAN] and backward rows xs_bwd to LA_XS_INIT -- and what the producers need to lower the forward rows is laid out:
// xs_mult [l][b] = max_i |s[b][i]| of layer l (the input of layer l is x * s, so |x * s| <= max|x| * max|s|: the producer of x lowers
// row l to pow2(xs_mult * its max |x|), la_xs_lower).  Layer 0 reads the constant input, which no kernel of the pass produces: its
// row gets pow2(max|const| * max|s|) here (max|const| reduced from the tensor itself every pass: no stale bound if a caller
// rewrites the parameter in place).  Every scale is therefore derived from the DATA of this pass -- there is no a-priori bound and
// no calibration (round 3 kept conv_clamp * max|s| for the forward rows and checked once, on the first batch, that it was tight enough).
__global__ __launch_bounds__(256) void la_xscale_bound_kernel(LaDemodTable t, const float* __restrict__ s_all, int s_stride,
                                                             const float* __restrict__ cst, int cst_n, float* __restrict__ xs,
                                                             float* __restrict__ xs_mult, int B, unsigned* __restrict__ xs_bwd) {
    __shared__ float red[4], red2[4];
    const int l = blockIdx.x, b = blockIdx.y;
    const float* sp = s_all + (long)b * s_stride + t.s_off[l];
    float m = 0.f, mc = 0.f;
    for (int i = threadIdx.x; i < t.cin[l]; i += 256) m = fmaxf(m, fabsf(sp[i]));
    if (l == 0) for (int i = threadIdx.x; i < cst_n; i += 256) mc = fmaxf(mc, fabsf(cst[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); mc = fmaxf(mc, __shfl_xor(mc, o, 64)); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; red2[threadIdx.x >> 6] = mc; }
    __syncthreads();
    const float smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float cmax = fmaxf(fmaxf(red2[0], red2[1]), fmaxf(red2[2], red2[3]));
    if (threadIdx.x == 0) xs_mult[(long)l * B + b] = smax;
    if (threadIdx.x < LA_XS_SUBS) {
        const long o = ((long)l * B + b) * LA_XS_FAN + threadIdx.x * LA_XS_LINE;
        xs[o] = (l == 0 && threadIdx.x == 0) ? la_pow2_scale(cmax * smax) : __uint_as_float(LA_XS_INIT);
        if (xs_bwd) xs_bwd[o] = LA_XS_INIT;
    }
}

int la_xscale_from_bounds(const LaDemodTable& t, const float* s_all, int s_stride, const float* cst, int cst_n, float* xs, float* xs_mult, int B,
                          hipStream_t stream, float* xs_bwd) {
    hipLaunchKernelGGL(la_xscale_bound_kernel, dim3(t.nlayers, B), dim3(256), 0, stream, t, s_all, s_stride, cst, cst_n, xs, xs_mult, B,
                       reinterpret_cast<unsigned*>(xs_bwd));
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// ToRGB forward (+ skip add):  rgb_pre[b][c][p] = sum_i wrgb[c][i] * s[b][i] * x[b][i][p] + bias[c]
//                              img[b][c][p]     = clamp(rgb_pre) + (skip ? skip[b][c][p] : 0)
// Streams x once (HBM-bound); lanes = consecutive pixels, 4 pixels per thread (float4).
// Skip image of a ToRGB layer computed on the fly (round 5): img = upsample2d(img_prev) + rgb needs the up-sampled image of the block
// below at the pixels this thread writes -- four consecutive pixels of one row read 2 x 4 values of the (tiny, 2-channel) low-resolution
// image.  Arithmetic = la_fir4x4_up2_kernel's (la_upfirdn2d.hip: taps in ascending row, column order), so the image equals the one the
// separate up-2 launch + skip read produced; that launch (one 5 us link of a step's serial chain per block) is gone for every block
// whose ToRGB is its own kernel.
struct LaSkipUp { const float* lo; int Hl, Wl; float f[16]; };      // lo: [B][imgc][Hl][Wl] (null: none); f: the up-2 taps (gain included)
__device__ __forceinline__ float4 la_up2_quad(const float* __restrict__ ip, int Hl, int Wl, const float* f, int Y, int x0) {
    const int i = Y >> 1, py = Y & 1, j0 = x0 >> 1;
    float v[2][4];                                               // in[i + py - 1 + ua][j0 - 1 .. j0 + 2], zero outside
#pragma unroll
    for (int ua = 0; ua < 2; ++ua) {
        const int iy = i - 1 + py + ua;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ix = j0 - 1 + c;
            v[ua][c] = (iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) ? ip[(long)iy * Wl + ix] : 0.f;
        }
    }
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = k & 1, jj = k >> 1;
        float s = 0.f;
#pragma unroll
        for (int ua = 0; ua < 2; ++ua)
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) s += v[ua][jj + px + ub] * f[(py + 2 * ua) * 4 + px + 2 * ub];
        o[k] = s;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

template <int IMGC>
__global__ __launch_bounds__(256) void la_torgb_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wrgb,
                                                          const float* __restrict__ s, int s_stride,
                                                          const float* __restrict__ bias, const float* __restrict__ skip,
                                                          float* __restrict__ rgb_pre, float* __restrict__ img, int C,
                                                          long HW, float clamp, LaTorgbMask mk, long p_lo, long p_hi, LaSkipUp su) {
    extern __shared__ float weff[];   // [IMGC][C]
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < IMGC * C; k += blockDim.x) {
        const int i = k % C;
        weff[k] = wrgb[k] * (s ? s[(long)b * s_stride + i] : 1.f);
    }
    __syncthreads();
    const long p4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (p4 >= HW) return;
    if (p_hi > 0 && (p4 < p_lo || p4 >= p_hi)) return;      // pixel window (rows nobody reads: la_synth.hip)
    const float* xb = x + (long)b * C * HW + p4;
    float4 acc[IMGC];
#pragma unroll
    for (int c = 0; c < IMGC; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    // (eight channels' 16-byte loads in flight per thread, accumulated in channel order: the kernel is a plain stream of x -- with one
    //  load per trip of a short unrolled loop the scheduler left most of the latency exposed once the packed-FP32 forms were gone)
    int i = 0;
    for (; i + 7 < C; i += 8) {
        float4 xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const float4*>(xb + (long)(i + u) * HW);
        if (mk.y) {      // x is a gradient still to be taken through the activation whose saved output is mk.y (same layout as x)
            const float* mb = mk.y + (long)b * C * HW + p4;
            float4 mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) mv[u] = *reinterpret_cast<const float4*>(mb + (long)(i + u) * HW);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xv[u].x *= la_act_bwd_from_y(mv[u].x, mk.act, mk.alpha, mk.gain, mk.clamp);
                xv[u].y *= la_act_bwd_from_y(mv[u].y, mk.act, mk.alpha, mk.gain, mk.clamp);
                xv[u].z *= la_act_bwd_from_y(mv[u].z, mk.act, mk.alpha, mk.gain, mk.clamp);
                xv[u].w *= la_act_bwd_from_y(mv[u].w, mk.act, mk.alpha, mk.gain, mk.clamp);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < IMGC; ++c) {
                const float wv = weff[c * C + i + u];
                acc[c].x += wv * xv[u].x; acc[c].y += wv * xv[u].y; acc[c].z += wv * xv[u].z; acc[c].w += wv * xv[u].w;
            }
    }
    for (; i < C; ++i) {
        float4 xv = *reinterpret_cast<const float4*>(xb + (long)i * HW);
        if (mk.y) {
            const float4 mv = *reinterpret_cast<const float4*>(mk.y + (long)b * C * HW + p4 + (long)i * HW);
            xv.x *= la_act_bwd_from_y(mv.x, mk.act, mk.alpha, mk.gain, mk.clamp); xv.y *= la_act_bwd_from_y(mv.y, mk.act, mk.alpha, mk.gain, mk.clamp);
            xv.z *= la_act_bwd_from_y(mv.z, mk.act, mk.alpha, mk.gain, mk.clamp); xv.w *= la_act_bwd_from_y(mv.w, mk.act, mk.alpha, mk.gain, mk.clamp);
        }
#pragma unroll
        for (int c = 0; c < IMGC; ++c) {
            const float wv = weff[c * C + i];
            acc[c].x += wv * xv.x; acc[c].y += wv * xv.y; acc[c].z += wv * xv.z; acc[c].w += wv * xv.w;
        }
    }
#pragma unroll
    for (int c = 0; c < IMGC; ++c) {
        const float bv = bias ? bias[c] : 0.f;
        float4 v = make_float4(acc[c].x + bv, acc[c].y + bv, acc[c].z + bv, acc[c].w + bv);
        const long o = ((long)b * IMGC + c) * HW + p4;
        if (rgb_pre) *reinterpret_cast<float4*>(rgb_pre + o) = v;
        if (clamp >= 0.f) {
            v.x = fminf(fmaxf(v.x, -clamp), clamp); v.y = fminf(fmaxf(v.y, -clamp), clamp);
            v.z = fminf(fmaxf(v.z, -clamp), clamp); v.w = fminf(fmaxf(v.w, -clamp), clamp);
        }
        if (skip) {
            const float4 sv = *reinterpret_cast<const float4*>(skip + o);
            v.x += sv.x; v.y += sv.y; v.z += sv.z; v.w += sv.w;
        } else if (su.lo) {
            const int W = 2 * su.Wl;
            const float4 sv = la_up2_quad(su.lo + ((long)b * IMGC + c) * su.Hl * su.Wl, su.Hl, su.Wl, su.f, (int)(p4 / W), (int)(p4 % W));
            v.x += sv.x; v.y += sv.y; v.z += sv.z; v.w += sv.w;
        }
        *reinterpret_cast<float4*>(img + o) = v;
    }
}

// Small planes (<= 64x64): the pixel grid alone cannot fill the chip and one thread walking all C channels is a chain of
// dependent-latency loads, so the channels are split over PARTS thread groups and combined through LDS in a fixed order.
template <int IMGC>
__global__ __launch_bounds__(256) void la_torgb_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ wrgb,
                                                                const float* __restrict__ s, int s_stride,
                                                                const float* __restrict__ bias, const float* __restrict__ skip,
                                                                float* __restrict__ rgb_pre, float* __restrict__ img, int C,
                                                                long HW, float clamp, int px_lanes, LaSkipUp su) {
    extern __shared__ float sm[];     // weff [IMGC][C]  then  comb [parts][px_lanes][IMGC] float4
    float* weff = sm;
    float4* comb = reinterpret_cast<float4*>(sm + ((IMGC * C + 3) & ~3));
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < IMGC * C; k += blockDim.x) {
        const int i = k % C;
        weff[k] = wrgb[k] * (s ? s[(long)b * s_stride + i] : 1.f);
    }
    __syncthreads();
    const int parts = 256 / px_lanes;
    const int pl = threadIdx.x % px_lanes, part = threadIdx.x / px_lanes;
    const long p4 = ((long)blockIdx.x * px_lanes + pl) * 4;
    const int per = (C + parts - 1) / parts;
    const int i0 = part * per, i1 = i0 + per < C ? i0 + per : C;
    const float* xb = x + (long)b * C * HW + p4;
    float4 acc[IMGC];
#pragma unroll
    for (int c = 0; c < IMGC; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p4 < HW) {
        int i = i0;
        for (; i + 7 < i1; i += 8) {      // (eight loads in flight, summed in channel order)
            float4 xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const float4*>(xb + (long)(i + u) * HW);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < IMGC; ++c) {
                    const float wv = weff[c * C + i + u];
                    acc[c].x += wv * xv[u].x; acc[c].y += wv * xv[u].y; acc[c].z += wv * xv[u].z; acc[c].w += wv * xv[u].w;
                }
        }
        for (; i < i1; ++i) {
            const float4 xv = *reinterpret_cast<const float4*>(xb + (long)i * HW);
#pragma unroll
            for (int c = 0; c < IMGC; ++c) {
                const float wv = weff[c * C + i];
                acc[c].x += wv * xv.x; acc[c].y += wv * xv.y; acc[c].z += wv * xv.z; acc[c].w += wv * xv.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < IMGC; ++c) comb[(part * px_lanes + pl) * IMGC + c] = acc[c];
    __syncthreads();
    if (part != 0 || p4 >= HW) return;
#pragma unroll
    for (int c = 0; c < IMGC; ++c) {
        float4 t = comb[pl * IMGC + c];
        for (int q = 1; q < parts; ++q) {
            const float4 u = comb[(q * px_lanes + pl) * IMGC + c];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        const float bv = bias ? bias[c] : 0.f;
        float4 v = make_float4(t.x + bv, t.y + bv, t.z + bv, t.w + bv);
        const long o = ((long)b * IMGC + c) * HW + p4;
        if (rgb_pre) *reinterpret_cast<float4*>(rgb_pre + o) = v;
        if (clamp >= 0.f) {
            v.x = fminf(fmaxf(v.x, -clamp), clamp); v.y = fminf(fmaxf(v.y, -clamp), clamp);
            v.z = fminf(fmaxf(v.z, -clamp), clamp); v.w = fminf(fmaxf(v.w, -clamp), clamp);
        }
        if (skip) {
            const float4 sv = *reinterpret_cast<const float4*>(skip + o);
            v.x += sv.x; v.y += sv.y; v.z += sv.z; v.w += sv.w;
        } else if (su.lo) {
            const int W = 2 * su.Wl;
            const float4 sv = la_up2_quad(su.lo + ((long)b * IMGC + c) * su.Hl * su.Wl, su.Hl, su.Wl, su.f, (int)(p4 / W), (int)(p4 % W));
            v.x += sv.x; v.y += sv.y; v.z += sv.z; v.w += sv.w;
        }
        *reinterpret_cast<float4*>(img + o) = v;
    }
}

int la_torgb_forward(const float* x, const float* wrgb, const float* s, int s_stride, const float* bias,
                     const float* skip, float* rgb_pre, float* img, int B, int C, int imgc, int H, int W, float clamp,
                     hipStream_t stream, const LaTorgbMask* mask, int row_lo, int row_hi, const float* skip_lo, const float* fir_host) {
    // skip_lo (instead of skip): the image of the block below [B][imgc][H/2][W/2]; its upsample2d (up 2, pad (2,1,2,1), gain 4, upfirdn2d.py:342-348)
    // is computed inside the kernel (la_up2_quad) with the 4x4 filter fir_host
    LaSkipUp su; su.lo = nullptr; su.Hl = H / 2; su.Wl = W / 2;
    for (int k = 0; k < 16; ++k) su.f[k] = 0.f;
    if (skip_lo) {
        LA_CHECK_ARG(!skip && fir_host && H % 2 == 0 && W % 4 == 0, "torgb: the on-the-fly skip image needs the filter, even rows and W % 4 == 0");
        su.lo = skip_lo;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) su.f[i * 4 + j] = 4.f * fir_host[(3 - i) * 4 + (3 - j)];      // (fir_fill: gain * flipped filter)
    }
    const long HW = (long)H * W;
    LA_CHECK_ARG(row_lo >= 0 && (row_hi == 0 || (row_hi > row_lo && row_hi <= H)), "torgb: bad row window");
    const long p_lo = (long)row_lo * W, p_hi = (long)row_hi * W;
    LaTorgbMask mk{nullptr, 0, 0.f, 0.f, 0.f};
    if (mask) mk = *mask;
    LA_CHECK_ARG(!mk.y || HW > 4096, "torgb: the fused activation backward exists for planes above 64x64 only");
    LA_CHECK_ARG(HW % 4 == 0, "torgb: H*W must be a multiple of 4");
    LA_CHECK_ARG(imgc >= 1 && imgc <= 4, "torgb: img_channels must be 1..4");
    // launch profiler: x streamed once (+ the image-sized outputs / skip)
    struct Bracket { int slot; hipStream_t st; ~Bracket() { la_prof_close(slot, st); } }
        br{la_prof_open(LA_PC_TORGB, 2.0 * B * imgc * (double)C * HW, 4.0 * B * ((double)C * HW + (skip ? 3.0 : 2.0) * imgc * (double)HW), stream), stream};
    if (HW <= 4096) {
        const long nq = HW / 4;
        int px_lanes = 1;
        while (px_lanes < 64 && px_lanes * 2 <= nq) px_lanes *= 2;
        dim3 grid((unsigned)la_cdiv(nq, px_lanes), B);
        const size_t lds = (size_t)((imgc * C + 3) & ~3) * sizeof(float) + (size_t)256 * imgc * sizeof(float4);
#define LAUNCH(N) hipLaunchKernelGGL(la_torgb_fwd_small_kernel<N>, grid, dim3(256), lds, stream, x, wrgb, s, s_stride, bias, skip, rgb_pre, img, C, HW, clamp, px_lanes, su)
        switch (imgc) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
        LA_CHECK_LAUNCH();
        return LA_OK;
    }
    dim3 grid(la_cdiv(HW / 4, 256), B);
    const size_t lds = (size_t)imgc * C * sizeof(float);
#define LAUNCH(N) hipLaunchKernelGGL(la_torgb_fwd_kernel<N>, grid, dim3(256), lds, stream, x, wrgb, s, s_stride, bias, skip, rgb_pre, img, C, HW, clamp, mk, p_lo, p_hi, su)
    switch (imgc) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Backward seam for one SynthesisLayer output y[b][c][p] (post-activation, saved by the forward):
//   g      = gx_next (gradient from the consumer conv, optional)  +  sum_k weff[b][k][c] * g_rgb[b][k][p]   (ToRGB, optional)
//   dweff[b][k][c] += sum_p g_rgb[b][k][p] * y                                            (ToRGB weight/style gradient)
//   g1     = g * act'(y)                                                                  (bias_act backward from the saved output)
//   ddn[b][c]      += sum_p g1 * (act^-1(y) - bias[c] - noise[p]*ns)                      (= dL/d(demod) * demod)
//   gz     = g1 * demod[b][c]                                                             (gradient w.r.t. the raw contraction)
// One workgroup per (slab, c, b); partial sums go to [.. ][slab] buffers (deterministic, no atomics).
template <int IMGC>
__global__ __launch_bounds__(256) void la_seam_bwd_kernel(LaSeamArgs a) {
    __shared__ float red[4];
    const int c = blockIdx.y, b = blockIdx.z, slab = blockIdx.x;
    const long HW = a.HW;
    const long w0 = a.p_hi > 0 ? a.p_lo / 4 : 0, w1 = a.p_hi > 0 ? a.p_hi / 4 : HW / 4;      // float4 groups of the pixel window (LaSeamArgs::p_lo)
    const long per = (w1 - w0 + gridDim.x - 1) / gridDim.x;   // float4 groups per slab
    const long q0 = w0 + slab * per < w1 ? w0 + slab * per : w1, q1 = (q0 + per < w1) ? q0 + per : w1;
    const long plane = ((long)b * a.C + c) * HW;
    const float dm = a.demod ? a.demod[(long)b * a.demod_stride + c] : 1.f;
    const float bv = a.bias ? a.bias[c] : 0.f;
    float weff[IMGC > 0 ? IMGC : 1];
#pragma unroll
    for (int k = 0; k < IMGC; ++k) weff[k] = a.wrgb[k * a.C + c] * a.s_rgb[(long)b * a.s_stride + c];
    float ddn = 0.f, gmax = 0.f;
    float* xs_row = a.xs_out ? a.xs_out + (long)b * LA_XS_FAN + la_xs_sub() : nullptr;
    const float xs_seen = (xs_row && threadIdx.x == 0) ? la_xs_peek(xs_row) : 0.f;      // (early: its round trip hides under the stream)
    // activation backward from the saved output as straight-line selects with reciprocals (the same arithmetic as the seam fused into
    // the contraction epilogues, la_conv_device.h): la_act_bwd_from_y / la_act_inv up to the rounding of 1/gain, 1/alpha
    const float s_pos = a.gain, s_neg = a.act == LA_ACT_LRELU ? a.gain * a.alpha : (a.act == LA_ACT_RELU ? 0.f : a.gain);
    const float i_gain = 1.f / a.gain, i_neg = a.act == LA_ACT_LRELU ? 1.f / (a.gain * a.alpha) : 1.f / a.gain;
    const float s_cl = a.clamp >= 0.f ? a.clamp : __builtin_huge_valf();
    float dwe[IMGC > 0 ? IMGC : 1];
#pragma unroll
    for (int k = 0; k < IMGC; ++k) dwe[k] = 0.f;

    // the HBM streams (y, the incoming gradient) of the next step are requested before this step's arithmetic: the kernel is a plain
    // stream, what limits it is the number of bytes in flight (the image-sized operands are L2 hits)
    const long qf = q0 + threadIdx.x;
    float4 y_n = make_float4(0.f, 0.f, 0.f, 0.f), g_n = y_n;
    if (qf < q1) {
        y_n = *reinterpret_cast<const float4*>(a.y + plane + qf * 4);
        if (a.gx_next) g_n = *reinterpret_cast<const float4*>(a.gx_next + plane + qf * 4);
    }
    for (long q = qf; q < q1; q += blockDim.x) {
        const long p = q * 4;
        const float4 yv = y_n;
        float4 g = g_n;
        if (q + blockDim.x < q1) {
            y_n = *reinterpret_cast<const float4*>(a.y + plane + p + 4 * (long)blockDim.x);
            if (a.gx_next) g_n = *reinterpret_cast<const float4*>(a.gx_next + plane + p + 4 * (long)blockDim.x);
        }
#pragma unroll
        for (int k = 0; k < IMGC; ++k) {
            const long o = ((long)b * IMGC + k) * HW + p;
            float4 gr = *reinterpret_cast<const float4*>(a.g_img + o);
            if (a.rgb_clamp >= 0.f) {
                const float4 rp = *reinterpret_cast<const float4*>(a.rgb_pre + o);
                // torch clamp backward passes the gradient where -c <= v <= c
                if (fabsf(rp.x) > a.rgb_clamp) gr.x = 0.f;
                if (fabsf(rp.y) > a.rgb_clamp) gr.y = 0.f;
                if (fabsf(rp.z) > a.rgb_clamp) gr.z = 0.f;
                if (fabsf(rp.w) > a.rgb_clamp) gr.w = 0.f;
            }
            g.x += weff[k] * gr.x; g.y += weff[k] * gr.y; g.z += weff[k] * gr.z; g.w += weff[k] * gr.w;
            dwe[k] += gr.x * yv.x + gr.y * yv.y + gr.z * yv.z + gr.w * yv.w;
        }
        float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.noise) {
            nz = *reinterpret_cast<const float4*>(a.noise + (long)b * a.noise_bstride + p);
            nz.x *= a.noise_strength; nz.y *= a.noise_strength; nz.z *= a.noise_strength; nz.w *= a.noise_strength;
        }
        float4 gz;
#define ONE(f)                                                                                \
        {                                                                                     \
            const bool pos = yv.f > 0.f;                                                      \
            const float g1 = g.f * (fabsf(yv.f) >= s_cl ? 0.f : (pos ? s_pos : s_neg));       \
            ddn += g1 * (yv.f * (pos ? i_gain : i_neg) - bv - nz.f);                          \
            gz.f = g1 * dm;                                                                   \
        }
        ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
        *reinterpret_cast<float4*>(a.gz + plane + p) = gz;
        gmax = fmaxf(gmax, fmaxf(fmaxf(fabsf(gz.x), fabsf(gz.y)), fmaxf(fabsf(gz.z), fabsf(gz.w))));
    }
    if (a.pmax_out || a.xs_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (a.pmax_out) a.pmax_out[((long)b * a.C + c) * gridDim.x + slab] = m;
            // the consumer's fp16 operand scale: this workgroup's maximum lowers a sub-slot of the sample's row (la_common.h)
            if (xs_row) la_xs_lower(xs_row, xs_seen, a.xs_mult, m);
        }
        __syncthreads();
    }
    const float ddn_t = la_block_sum_256(ddn, red);
    if (threadIdx.x == 0 && a.ddn_part) a.ddn_part[((long)b * a.C + c) * gridDim.x + slab] = ddn_t;
#pragma unroll
    for (int k = 0; k < IMGC; ++k) {
        const float t = la_block_sum_256(dwe[k], red);
        if (threadIdx.x == 0) a.dweff_part[(((long)b * IMGC + k) * a.C + c) * gridDim.x + slab] = t;
    }
}

int la_seam_slabs(long HW) {
    long s = HW / 16384;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}

int la_seam_backward(const LaSeamArgs& a, int B, int imgc, hipStream_t stream) {
    LA_CHECK_ARG(a.HW % 4 == 0, "seam: H*W must be a multiple of 4");
    LA_CHECK_ARG(imgc >= 0 && imgc <= 4, "seam: img_channels must be 0..4");
    LA_CHECK_ARG(a.p_hi == 0 ? a.p_lo == 0 : (a.p_lo >= 0 && a.p_lo < a.p_hi && a.p_hi <= a.HW && a.p_lo % 4 == 0 && a.p_hi % 4 == 0), "seam: bad pixel window");
    dim3 grid(la_seam_slabs(a.HW), a.C, B);
    // launch profiler: read y and the incoming gradient, write gz (+ the image-sized ToRGB operands)
    const double wf = a.p_hi > 0 ? (double)(a.p_hi - a.p_lo) / (double)a.HW : 1.0;
    const int pslot = la_prof_open(LA_PC_SEAM, 0.0, wf * 4.0 * B * ((double)a.C * a.HW * (a.gx_next ? 3.0 : 2.0) + 2.0 * imgc * (double)a.HW), stream);
#define LAUNCH(N) hipLaunchKernelGGL(la_seam_bwd_kernel<N>, grid, dim3(256), 0, stream, a)
    switch (imgc) { case 0: LAUNCH(0); break; case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
    la_prof_close(pslot, stream);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// style-gradient finish for one conv layer:
//   ds[b][i] = sum_tiles ds_part[b][i][.]  -  s[b][i] * sum_o (sum_slabs ddn_part[b][o][.]) * d[b][o]^2 * wsq[o][i]
// pass 1 (wide): every partial row (one value per pixel tile / slab, contiguous) is summed by one wave with coalesced
//   reads and a fixed shuffle tree (deterministic); the sum replaces element 0 of the row.
// pass 2 (small): one thread per i (coalesced wsq reads over i), loop over o; q[b][o] staged in LDS.
__global__ __launch_bounds__(256) void la_rows_sum_inplace_kernel(float* __restrict__ a0, long rows0, int n0,
                                                                 float* __restrict__ a1, long rows1, int n1) {
    const int lane = threadIdx.x & 63;
    long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float* row;
    int n;
    if (r < rows0) { row = a0 + r * n0; n = n0; }
    else if (r - rows0 < rows1) { row = a1 + (r - rows0) * n1; n = n1; }
    else return;
    // The per-tile partials of a row (up to 8192 of them at 1024^2) are summed in float64: they are tiny next to the contraction
    // that produced them, and against a smooth image gradient the sum is much smaller than its terms -- a float32 sum of the
    // partials was the part of the style-gradient noise at 1024^2 that is NOT inherent in forming the data gradient first.
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int k = lane;
    for (; k + 192 < n; k += 256) { v0 += (double)row[k]; v1 += (double)row[k + 64]; v2 += (double)row[k + 128]; v3 += (double)row[k + 192]; }
    for (; k < n; k += 64) v0 += (double)row[k];
    double v = (v0 + v1) + (v2 + v3);
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft, 64);
    if (lane == 0) row[0] = (float)v;
}

__global__ __launch_bounds__(256) void la_style_bwd_conv_kernel(const float* __restrict__ ds_part, int ntiles,
                                                               const float* __restrict__ ddn_part, int nslabs,
                                                               const float* __restrict__ d, int d_stride,
                                                               const float* __restrict__ s, int s_stride,
                                                               const float* __restrict__ wsq, int cin, int cout,
                                                               float* __restrict__ ds_out, int ds_stride) {
    // block = 16 input channels x 16 output-channel parts (a wide grid of short loops: this kernel is pure latency);
    // q[b][o] staged in LDS, the parts combined through LDS in a fixed order
    extern __shared__ float q[];   // [cout] + [16][16]
    float* comb = q + cout;
    const int b = blockIdx.y;
    for (int o = threadIdx.x; o < cout; o += blockDim.x) {
        const float dv = d[(long)b * d_stride + o];
        q[o] = ddn_part[((long)b * cout + o) * nslabs] * dv * dv;
    }
    __syncthreads();
    const int il = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < cin) {
        const int per = (cout + 15) / 16;
        const int o0 = part * per, o1 = o0 + per < cout ? o0 + per : cout;
        int o = o0;
        for (; o + 3 < o1; o += 4) {
            a0 += q[o] * wsq[(long)o * cin + i];
            a1 += q[o + 1] * wsq[(long)(o + 1) * cin + i];
            a2 += q[o + 2] * wsq[(long)(o + 2) * cin + i];
            a3 += q[o + 3] * wsq[(long)(o + 3) * cin + i];
        }
        for (; o < o1; ++o) a0 += q[o] * wsq[(long)o * cin + i];
    }
    comb[part * 16 + il] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (part == 0 && i < cin) {
        float tot = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) tot += comb[p * 16 + il];
        ds_out[(long)b * ds_stride + i] = ds_part[((long)b * cin + i) * ntiles] - s[(long)b * s_stride + i] * tot;
    }
}

int la_style_backward_conv(float* ds_part, int ntiles, float* ddn_part, int nslabs, const float* d,
                           int d_stride, const float* s, int s_stride, const float* wsq, int cin, int cout, int B,
                           float* ds_out, int ds_stride, hipStream_t stream) {
    const long rows0 = (long)B * cout, rows1 = (long)B * cin;
    hipLaunchKernelGGL(la_rows_sum_inplace_kernel, dim3((unsigned)la_cdiv(rows0 + rows1, 4)), dim3(256), 0, stream, ddn_part, rows0,
                       nslabs, ds_part, rows1, ntiles);
    hipLaunchKernelGGL(la_style_bwd_conv_kernel, dim3(la_cdiv(cin, 16), B), dim3(256), (cout + 256) * sizeof(float), stream,
                       ds_part, ntiles, ddn_part, nslabs, d, d_stride, s, s_stride, wsq, cin, cout, ds_out, ds_stride);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// One-pass finish for all layers (LaStyleFinish): same arithmetic and summation order as the per-layer kernels above.
struct LaRowSegs { int nseg; long row0[2 * LA_FIN_MAX_CONV + LA_FIN_MAX_RGB + 1]; float* ptr[2 * LA_FIN_MAX_CONV + LA_FIN_MAX_RGB]; int n[2 * LA_FIN_MAX_CONV + LA_FIN_MAX_RGB]; };

__global__ __launch_bounds__(256) void la_rows_sum_all_kernel(LaRowSegs g) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= g.row0[g.nseg]) return;
    int sg = 0, hi = g.nseg - 1;                 // binary search of the segment (a linear scan cost ~50 scalar loads per wave)
    while (sg < hi) {
        const int mid = (sg + hi + 1) >> 1;
        if (r >= g.row0[mid]) sg = mid; else hi = mid - 1;
    }
    const int n = g.n[sg];
    float* row = g.ptr[sg] + (r - g.row0[sg]) * n;
    // The per-tile partials of a row (up to 8192 of them at 1024^2) are summed in float64: they are tiny next to the contraction
    // that produced them, and against a smooth image gradient the sum is much smaller than its terms -- a float32 sum of the
    // partials was the part of the style-gradient noise at 1024^2 that is NOT inherent in forming the data gradient first.
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int k = lane;
    for (; k + 192 < n; k += 256) { v0 += (double)row[k]; v1 += (double)row[k + 64]; v2 += (double)row[k + 128]; v3 += (double)row[k + 192]; }
    for (; k < n; k += 64) v0 += (double)row[k];
    double v = (v0 + v1) + (v2 + v3);
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft, 64);
    if (lane == 0) row[0] = (float)v;
}

__global__ __launch_bounds__(256) void la_style_bwd_conv_all_kernel(LaStyleFinish f) {
    extern __shared__ float q[];   // [cout] + [16][16]
    int l = 0;
    while (l + 1 < f.nconv && (int)blockIdx.x >= f.conv[l + 1].blk0) ++l;
    const LaStyleFinish::Conv& L = f.conv[l];
    const int cin = L.cin, cout = L.cout;
    float* comb = q + cout;
    const int b = blockIdx.y;
    for (int o = threadIdx.x; o < cout; o += blockDim.x) {
        const float dv = L.d[(long)b * f.d_stride + o];
        q[o] = L.ddn_part[((long)b * cout + o) * L.nslabs] * dv * dv;
    }
    __syncthreads();
    const int il = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int i = ((int)blockIdx.x - L.blk0) * 16 + il;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < cin) {
        const int per = (cout + 15) / 16;
        const int o0 = part * per, o1 = o0 + per < cout ? o0 + per : cout;
        int o = o0;
        for (; o + 3 < o1; o += 4) {
            a0 += q[o] * L.wsq[(long)o * cin + i];
            a1 += q[o + 1] * L.wsq[(long)(o + 1) * cin + i];
            a2 += q[o + 2] * L.wsq[(long)(o + 2) * cin + i];
            a3 += q[o + 3] * L.wsq[(long)(o + 3) * cin + i];
        }
        for (; o < o1; ++o) a0 += q[o] * L.wsq[(long)o * cin + i];
    }
    comb[part * 16 + il] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (part == 0 && i < cin) {
        float tot = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) tot += comb[p * 16 + il];
        L.ds_out[(long)b * f.ds_stride + i] = L.ds_part[((long)b * cin + i) * L.ntiles] - L.s[(long)b * f.s_stride + i] * tot;
    }
}

__global__ void la_style_bwd_rgb_all_kernel(LaStyleFinish f) {
    int l = 0;
    while (l + 1 < f.nrgb && (int)blockIdx.x >= f.rgb[l + 1].blk0) ++l;
    const LaStyleFinish::Rgb& T = f.rgb[l];
    const int b = blockIdx.y;
    const int i = ((int)blockIdx.x - T.blk0) * blockDim.x + threadIdx.x;
    if (i >= T.C) return;
    float acc = 0.f;
    for (int k = 0; k < f.imgc; ++k) acc += T.dweff_part[(((long)b * f.imgc + k) * T.C + i) * T.nslabs] * T.wrgb[k * T.C + i];
    T.ds_out[(long)b * f.ds_stride + i] = acc;
}

int la_style_backward_all(const LaStyleFinish& fin, int B, hipStream_t stream) {
    LA_CHECK_ARG(fin.nconv >= 0 && fin.nconv <= LA_FIN_MAX_CONV && fin.nrgb >= 0 && fin.nrgb <= LA_FIN_MAX_RGB, "style_backward_all: too many layers");
    LaStyleFinish f = fin;
    LaRowSegs g;
    g.nseg = 0; g.row0[0] = 0;
    // (rows of a single partial are already their own sum: nothing to do)
    auto seg = [&](float* p, long rows, int n) { if (n <= 1) return; g.ptr[g.nseg] = p; g.n[g.nseg] = n; g.row0[g.nseg + 1] = g.row0[g.nseg] + rows; ++g.nseg; };
    int cblk = 0, rblk = 0, max_cout = 0;
    for (int l = 0; l < f.nconv; ++l) {
        LaStyleFinish::Conv& L = f.conv[l];
        seg(L.ddn_part, (long)B * L.cout, L.nslabs);
        seg(L.ds_part, (long)B * L.cin, L.ntiles);
        L.blk0 = cblk; cblk += la_cdiv(L.cin, 16);
        if (L.cout > max_cout) max_cout = L.cout;
    }
    for (int l = 0; l < f.nrgb; ++l) {
        LaStyleFinish::Rgb& T = f.rgb[l];
        seg(T.dweff_part, (long)B * f.imgc * T.C, T.nslabs);
        T.blk0 = rblk; rblk += la_cdiv(T.C, 256);
    }
    if (g.nseg > 0)
        hipLaunchKernelGGL(la_rows_sum_all_kernel, dim3((unsigned)la_cdiv(g.row0[g.nseg], 4)), dim3(256), 0, stream, g);
    if (f.nconv) hipLaunchKernelGGL(la_style_bwd_conv_all_kernel, dim3(cblk, B), dim3(256), (max_cout + 256) * sizeof(float), stream, f);
    if (f.nrgb) hipLaunchKernelGGL(la_style_bwd_rgb_all_kernel, dim3(rblk, B), dim3(256), 0, stream, f);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ToRGB style gradient: ds[b][i] = sum_k wrgb[k][i] * sum_slabs dweff_part[b][k][i][.]
__global__ void la_style_bwd_rgb_kernel(const float* __restrict__ dweff_part, int nslabs, const float* __restrict__ wrgb,
                                        int C, int imgc, float* __restrict__ ds_out, int ds_stride) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    float acc = 0.f;
    for (int k = 0; k < imgc; ++k) {
        float v = 0.f;
        for (int t = 0; t < nslabs; ++t) v += dweff_part[(((long)b * imgc + k) * C + i) * nslabs + t];
        acc += v * wrgb[k * C + i];
    }
    ds_out[(long)b * ds_stride + i] = acc;
}

int la_style_backward_rgb(const float* dweff_part, int nslabs, const float* wrgb, int C, int imgc, int B,
                          float* ds_out, int ds_stride, hipStream_t stream) {
    hipLaunchKernelGGL(la_style_bwd_rgb_kernel, dim3(la_cdiv(C, 256), B), dim3(256), 0, stream, dweff_part, nslabs,
                       wrgb, C, imgc, ds_out, ds_stride);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

// ------------------------------------------------------------------------------------------------------------
// affine backward: dws[b][slot][j] = wgain * sum_{l: widx_l == slot} post_gain_l * sum_i ds[b][row_l + i] * A_l[i][j]
// stage 1: one block per 64-row chunk of a layer -> part[chunk][b][j] (thread = 2 consecutive j, coalesced A rows)
// stage 2: per (b, slot, j): sum the chunks of the layers feeding that slot in a fixed order (deterministic).
#define AFF_ROWS 16
__device__ __forceinline__ int la_chunk_layer(const LaStyleTable& t, int chunk, int* row0) {
    int l = 0, c0 = 0;
    for (; l < t.nlayers; ++l) {
        const int nc = (t.row_start[l + 1] - t.row_start[l] + AFF_ROWS - 1) / AFF_ROWS;
        if (chunk < c0 + nc) break;
        c0 += nc;
    }
    *row0 = (chunk - c0) * AFF_ROWS;
    return l;
}

// part[chunk][b][j] = sum_{i in chunk} A_l[i][j] * post_gain_l * ds[b][row_start_l + i]
// One workgroup per chunk of AFF_ROWS rows of one layer: thread = (float4 column group, row half); every thread keeps
// AFF_ROWS / 2 independent 16-byte row loads in flight (the first version walked 64 rows with one dword load each from 150
// workgroups: 93 us for 20 MB), the two row halves are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void la_affine_bwd_part_kernel(LaStyleTable t, const float* __restrict__ ds_all, int B,
                                                                int wdim, float* __restrict__ part, int nchunks) {
    __shared__ float dsl[BCH][AFF_ROWS];
    __shared__ float4 comb[128][BCH];
    const int chunk = blockIdx.x;
    int row0;
    const int l = la_chunk_layer(t, chunk, &row0);
    if (l >= t.nlayers) return;
    const int rows = t.row_start[l + 1] - t.row_start[l];
    const int nr = row0 + AFF_ROWS < rows ? AFF_ROWS : rows - row0;
    const float pg = t.post_gain[l];
    const float* A = t.aw[l] + (long)row0 * wdim;
    const int jq = threadIdx.x & 127, half = threadIdx.x >> 7;
    const int w4 = wdim >> 2;                 // (wdim % 4 == 0 is checked by the host)
    for (int b0 = blockIdx.y * BCH; b0 < B; b0 += gridDim.y * BCH) {
        __syncthreads();
        for (int k = threadIdx.x; k < BCH * AFF_ROWS; k += 256) {
            const int q = k / AFF_ROWS, i = k - q * AFF_ROWS;
            dsl[q][i] = (b0 + q < B && i < nr) ? ds_all[(long)(b0 + q) * t.total_rows + t.row_start[l] + row0 + i] * pg : 0.f;
        }
        __syncthreads();
        for (int j0 = 0; j0 < w4; j0 += 128) {          // (uniform trip count: the body holds barriers)
            const int j4 = j0 + jq;
            const bool jok = j4 < w4;
            float4 av[AFF_ROWS / 2];
#pragma unroll
            for (int r = 0; r < AFF_ROWS / 2; ++r) {
                const int i = half * (AFF_ROWS / 2) + r;
                av[r] = (jok && i < nr) ? reinterpret_cast<const float4*>(A + (long)i * wdim)[j4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4 acc[BCH];
#pragma unroll
            for (int q = 0; q < BCH; ++q) {
                acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < AFF_ROWS / 2; ++r) {
                    const float d = dsl[q][half * (AFF_ROWS / 2) + r];
                    acc[q].x += av[r].x * d; acc[q].y += av[r].y * d; acc[q].z += av[r].z * d; acc[q].w += av[r].w * d;
                }
            }
            if (half == 1) {
#pragma unroll
                for (int q = 0; q < BCH; ++q) comb[jq][q] = acc[q];
            }
            __syncthreads();
            if (half == 0) {
#pragma unroll
                for (int q = 0; q < BCH; ++q)
                    if (jok && b0 + q < B) {
                        const float4 o = comb[jq][q];
                        reinterpret_cast<float4*>(part + ((long)chunk * B + b0 + q) * wdim)[j4] =
                            make_float4(acc[q].x + o.x, acc[q].y + o.y, acc[q].z + o.z, acc[q].w + o.w);
                    }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void la_affine_bwd_sum_kernel(LaStyleTable t, const float* __restrict__ part, int B,
                                                               int wdim, float wgain, float* __restrict__ dws, int num_ws) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = blockIdx.y, b = blockIdx.z;
    if (j >= wdim) return;
    float acc = 0.f;
    int c0 = 0;
    for (int l = 0; l < t.nlayers; ++l) {
        const int nc = (t.row_start[l + 1] - t.row_start[l] + AFF_ROWS - 1) / AFF_ROWS;
        if (t.widx[l] == slot) {      // (eight chunk loads in flight, added in chunk order: the loop was one exposed load latency per chunk)
            int c = c0;
            for (; c + 7 < c0 + nc; c += 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = part[((long)(c + k) * B + b) * wdim + j];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k];
            }
            for (; c < c0 + nc; ++c) acc += part[((long)c * B + b) * wdim + j];
        }
        c0 += nc;
    }
    dws[((long)b * num_ws + slot) * wdim + j] = acc * wgain;
}

int la_affine_bwd_chunks(const LaStyleTable& t) {
    int n = 0;
    for (int l = 0; l < t.nlayers; ++l) n += (t.row_start[l + 1] - t.row_start[l] + AFF_ROWS - 1) / AFF_ROWS;
    return n;
}

int la_affine_backward(const LaStyleTable& t, const float* ds_all, int B, int wdim, float* dws, int num_ws, float* part,
                       hipStream_t stream) {
    const int nchunks = la_affine_bwd_chunks(t);
    LA_CHECK_ARG(wdim % 4 == 0 && (((size_t)part | (size_t)t.aw[0]) & 15) == 0, "affine_backward: w_dim must be a multiple of 4 (16-byte rows)");
    hipLaunchKernelGGL(la_affine_bwd_part_kernel, dim3(nchunks, la_cdiv(B, BCH)), dim3(256), 0, stream, t, ds_all, B, wdim,
                       part, nchunks);
    hipLaunchKernelGGL(la_affine_bwd_sum_kernel, dim3(la_cdiv(wdim, 256), num_ws, B), dim3(256), 0, stream, t, part, B, wdim,
                       1.0f / sqrtf((float)wdim), dws, num_ws);
    LA_CHECK_LAUNCH();
    return LA_OK;
}
