// Device-side pieces shared by the contraction kernels (fp32 MFMA and split-bf16 MFMA): the fused epilogue.
// Both kernel families leave the same kind of accumulator fragment: acc[i][j][r] of a 32x32 MFMA tile.  The 4 waves of a
// workgroup form a WM_ x (4/WM_) grid over the MT x 128 tile (2x2: fp32 kernel and the 64-row 16-bit tiles; 4x1: the
// 128-row 16-bit tiles, where every wave owns 32 rows x all 128 pixels so that no weight fragment is loaded twice):
//   m = m0 + wm*(MT/WM_) + i*32 + (r&3) + 8*(r>>2) + 4*(lane>>5) ;  pixel = ntile*128 + (wn*NJ + j)*32 + (lane&31).
#pragma once
#include "la_conv.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Development build: per-wave segment clocks (s_memtime differences in scalar registers) of a kernel that passes a LaStamp to its epilogue
// (la_conv_bf16.hip, dev knob LA_KNOB_HALO_STAMP; scripts/halo_wave_timeline.py).  Segments 0-5 belong to the kernel, 6.. to the epilogue.
#ifdef LA_DEV
struct LaStamp { unsigned long long last, seg[12]; bool on; };
#define LA_ESTAMP(k) do { if (stp && stp->on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stp->seg[k] += t_ - stp->last; stp->last = t_; } } while (0)
#define LA_ESTAMP_PARAM , LaStamp* stp = nullptr
#else
#define LA_ESTAMP(k)
#define LA_ESTAMP_PARAM
#endif

#define NT 128

__device__ __forceinline__ float la_conv_epi_fwd(const LaConvArgs& a, float v, float dmv, float nz, float bv) {
    return la_act_fwd(v * dmv + nz + bv, a.act, a.alpha, a.gain, a.clamp);
}

// TILE2D: the 128 pixels of a tile are 4 rows x 32 columns of the output grid (halo kernel) instead of 128 consecutive
// grid positions; wave N-subtile (wn, j) is then row wn*2 + j of the tile.
// A pixel tile outside the row window of a backward launch (LaConvArgs::row_lo): its gradient is exactly zero and nothing of it is computed
// or stored, but the one-pass style finish sums the per-tile partials of EVERY tile -- they are zeroed here by the workgroup that returns.
template <int MT>
__device__ __forceinline__ void la_conv_zero_partials(const LaConvArgs& a, int b, int m0, int tile) {
    const int tid = threadIdx.x;
    if (a.epi != LA_EPI_BWD || tid >= MT || m0 + tid >= a.M) return;
    const long slot = ((long)b * a.M + m0 + tid) * a.tiles_per_sample + tile;
    if (a.ds_part) a.ds_part[slot] = 0.f;
    if (a.seam_ddn_part) {
        a.seam_ddn_part[slot] = 0.f;
        if (a.seam_pmax) a.seam_pmax[slot] = 0.f;
        if (a.seam_dweff_part)
            for (int c = 0; c < a.seam_imgc && c < 4; ++c) a.seam_dweff_part[(((long)b * a.seam_imgc + c) * a.M + m0 + tid) * a.tiles_per_sample + tile] = 0.f;
    }
}

template <int MT, bool SPLIT, bool TILE2D = false, int WM_ = 2>
__device__ __forceinline__ void la_conv_epilogue(const LaConvArgs& a, f32x16 (&acc)[MT / (32 * WM_)][WM_], float (*red)[MT],   // red: [WN_ > 2 ? 6 : 4][MT] LDS floats
                                                 int ntile, int m0, int G, int Ntot, int b_sel = -1 LA_ESTAMP_PARAM) {
    constexpr int TM = MT / (32 * WM_);       // 32-row MFMA tiles per wave
    constexpr int WN_ = 4 / WM_;              // waves along the pixels
    constexpr int NJ = WM_;                   // 32-pixel MFMA tiles per wave (128 / 32 / WN_)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN_, wn = wid % WN_;
    const int l31 = lane & 31, lh = lane >> 5;
    if (SPLIT) {
        // raw slice accumulators -> ws[slice][b][m][g]
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nidx = ntile * NT + (wn * NJ + j) * 32 + l31;
            if (nidx >= Ntot) continue;
            const int b = nidx / G, g = nidx - b * G;
            float* wsp = a.splitk_ws + (((long)blockIdx.z * a.B + b) * a.M) * G + g;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (MT / WM_) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < a.M) wsp[(long)m * G] = acc[i][j][r];
                }
        }
        return;
    }

    const int b = b_sel >= 0 ? b_sel : (int)blockIdx.z;      // (merged-phase launches pass the sample explicitly)
    // row pitch / plane stride of `out`: raw (epilogue-free) launches may write a padded scratch layout (LaConvArgs::out_pitch)
    const int wpitch = (a.epi == LA_EPI_RAW && a.out_pitch > 0) ? a.out_pitch : a.Wout;
    const long oplane = (a.epi == LA_EPI_RAW && a.out_plane > 0) ? a.out_plane : (long)a.Hout * a.Wout;
    // ---- fast path: the whole MT x 128 tile is inside the output (block-uniform).  Straight-line code: per-row parameters
    // come from LDS (staged with one coalesced load), every global load is unconditional and issued in one batch, so the
    // 64 stores of a lane are not serialised behind 32 dependent round trips to L2/HBM.
    if (m0 + MT <= a.M && (TILE2D || (ntile + 1) * NT <= G)) {
        float* prm = &red[WN_ > 2 ? 4 : 2][0];           // [2][MT] row parameters (red[0 .. WN_) stay the ds_part scratch)
        long np[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int gy, gx;
            if (TILE2D) {
                const int tpr = a.Gx >> 5;
                const int tyb = ntile / tpr, txb = ntile - tyb * tpr;
                gy = tyb * 4 + wn * NJ + j; gx = txb * 32 + l31;
            } else {
                const int g = ntile * NT + (wn * NJ + j) * 32 + l31;
                gy = g / a.Gx; gx = g - gy * a.Gx;
            }
            np[j] = (long)(gy * a.out_sy + a.out_oy) * wpitch + (gx * a.out_sx + a.out_ox);
        }
        const long HWo = oplane;
        const int mw = wm * (MT / WM_) + 4 * lh;          // first row of this lane inside the tile
        float* o0 = a.out + ((long)b * a.M + m0 + mw) * HWo;
        if (a.epi == LA_EPI_BWD) {
            const float* os_b = a.out_scale ? a.out_scale + (long)b * a.oscale_stride : nullptr;
            if (tid < MT) prm[tid] = os_b ? os_b[m0 + tid] : 1.f;
            const float* x0 = a.xin ? a.xin + (long)b * a.xin_bstride + (long)(m0 + mw) * HWo : nullptr;
            // fused seam of the layer that produced xin (see LaConvArgs): LDS rows 6, 7 = its demod / bias, 8.. = ddn partials,
            // 12.. = maxima (the 16-bit kernels' LDS is large enough; the fp32 kernel never sets seam_ddn_part)
            const bool seam = a.seam_ddn_part != nullptr && x0 != nullptr;
            const float xs_seen = (seam && a.seam_xs_out) ? la_xs_peek(a.seam_xs_out + (long)b * LA_XS_FAN + la_xs_sub()) : 0.f;      // (early: its latency hides under the epilogue)
            // (activation backward from the saved output as straight-line selects with reciprocals: same values as
            //  la_act_bwd_from_y / la_act_inv up to the rounding of 1/gain, 1/alpha)
            const float s_pos = a.seam_gain, s_neg = a.seam_act == LA_ACT_LRELU ? a.seam_gain * a.seam_alpha : (a.seam_act == LA_ACT_RELU ? 0.f : a.seam_gain);
            const float i_gain = 1.f / a.seam_gain, i_neg = a.seam_act == LA_ACT_LRELU ? 1.f / (a.seam_gain * a.seam_alpha) : 1.f / a.seam_gain;
            const float s_cl = a.seam_clamp >= 0.f ? a.seam_clamp : __builtin_huge_valf();
            float nz0[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) nz0[j] = 0.f;
            // ToRGB part of the fused seam (seam_imgc image channels): per-pixel image gradients (masked by the ToRGB clamp) in
            // registers, per-row effective ToRGB weights in LDS rows 16.., weight-gradient partials in rows 20..
            constexpr int SEAM_MAXC = 4;
            const int imgc = seam ? a.seam_imgc : 0;
            float gr[SEAM_MAXC][NJ];
#pragma unroll
            for (int c = 0; c < SEAM_MAXC; ++c)
#pragma unroll
                for (int j = 0; j < NJ; ++j) gr[c][j] = 0.f;
            if (seam) {
                if (tid < MT) {
                    red[6][tid] = a.seam_demod ? a.seam_demod[(long)b * a.seam_demod_stride + m0 + tid] : 1.f;
                    red[7][tid] = a.seam_bias ? a.seam_bias[m0 + tid] : 0.f;
#pragma unroll
                    for (int c = 0; c < SEAM_MAXC; ++c)
                        if (c < imgc) red[16 + c][tid] = a.seam_wrgb[(long)c * a.M + m0 + tid] * a.seam_srgb[(long)b * a.seam_srgb_stride + m0 + tid];
                }
                if (a.seam_noise) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) nz0[j] = a.seam_noise[(long)b * a.seam_noise_bstride + np[j]] * a.seam_noise_strength;
                }
#pragma unroll
                for (int c = 0; c < SEAM_MAXC; ++c)
                    if (c < imgc) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const long o = ((long)b * imgc + c) * HWo + np[j];
                            float gv = a.seam_gimg[o];
                            if (a.seam_rgb_clamp >= 0.f && fabsf(a.seam_rgbpre[o]) > a.seam_rgb_clamp) gv = 0.f;
                            gr[c][j] = gv;
                        }
                    }
            }
            LA_ESTAMP(6);
            __syncthreads();
            LA_ESTAMP(7);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // xin of the next four-row group is requested while this group is processed (two groups of 4 x NJ registers live,
                // not all sixteen rows: the epilogue is the register peak of the kernels)
                float xn[4][NJ];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) xn[q][j] = x0 ? x0[(long)(i * 32 + q) * HWo + np[j]] : 0.f;
                // four rows at a time (r = 4g .. 4g+3 are four CONSECUTIVE channels), so that the per-row reductions over the 32
                // lanes of a half-wave run as one multi-value butterfly: 2 + 1 exchanges halve the values per lane from 4 to 1
                // (lane bits 4, 3 pick the row), 3 more finish it -- 6 shuffles per 4 rows and quantity instead of 20
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float xv[4][NJ];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) xv[q][j] = xn[q][j];
                    if (g4 < 3) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int j = 0; j < NJ; ++j) xn[q][j] = x0 ? x0[(long)(i * 32 + q + 8 * (g4 + 1)) * HWo + np[j]] : 0.f;
                    }
                    float part[4], dd[4], mx[4], dwe[SEAM_MAXC][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = g4 * 4 + q;
                        const int mr = i * 32 + q + 8 * g4;
                        const float sc = prm[mw + mr];
                        part[q] = dd[q] = mx[q] = 0.f;
#pragma unroll
                        for (int c = 0; c < SEAM_MAXC; ++c) dwe[c][q] = 0.f;
                        if (seam) {
                            const float dm0 = red[6][mw + mr], b0 = red[7][mw + mr];
                            float we[SEAM_MAXC];
#pragma unroll
                            for (int c = 0; c < SEAM_MAXC; ++c) we[c] = c < imgc ? red[16 + c][mw + mr] : 0.f;
#pragma unroll
                            for (int j = 0; j < NJ; ++j) {
                                const float v = acc[i][j][r], y = xv[q][j];
                                const bool pos = y > 0.f;
                                const float sl = fabsf(y) >= s_cl ? 0.f : (pos ? s_pos : s_neg);
                                float g = v * sc;
                                if (imgc > 0) {      // (block-uniform: the up layers' own seam carries no ToRGB part)
#pragma unroll
                                    for (int c = 0; c < SEAM_MAXC; ++c)
                                        if (c < imgc) { g += we[c] * gr[c][j]; dwe[c][q] += gr[c][j] * y; }
                                }
                                const float g1 = g * sl;
                                dd[q] += g1 * (y * (pos ? i_gain : i_neg) - b0 - nz0[j]);
                                const float gz = g1 * dm0;
                                o0[(long)mr * HWo + np[j]] = gz;
                                mx[q] = fmaxf(mx[q], fabsf(gz));
                                part[q] += v * y;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < NJ; ++j) {
                                const float v = acc[i][j][r];
                                o0[(long)mr * HWo + np[j]] = v * sc;
                                if (x0) part[q] += v * xv[q][j];
                            }
                        }
                    }
                    // lane (bit4, bit3) ends up with row 2*bit4 + bit3 of the group
                    const bool up16 = (l31 & 16) != 0, up8 = (l31 & 8) != 0;
                    const int mrow = mw + i * 32 + 8 * g4 + (up16 ? 2 : 0) + (up8 ? 1 : 0);
                    const bool writer = (l31 & 7) == 0;
                    auto bfly_sum = [&](float (&v)[4]) {
                        const float k0 = up16 ? v[2] : v[0], s0 = up16 ? v[0] : v[2];
                        const float k1 = up16 ? v[3] : v[1], s1 = up16 ? v[1] : v[3];
                        const float a0 = k0 + __shfl_xor(s0, 16, 64), a1 = k1 + __shfl_xor(s1, 16, 64);
                        float t = (up8 ? a1 : a0) + __shfl_xor(up8 ? a0 : a1, 8, 64);
                        t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
                        return t;
                    };
                    auto bfly_max = [&](float (&v)[4]) {
                        const float k0 = up16 ? v[2] : v[0], s0 = up16 ? v[0] : v[2];
                        const float k1 = up16 ? v[3] : v[1], s1 = up16 ? v[1] : v[3];
                        const float a0 = fmaxf(k0, __shfl_xor(s0, 16, 64)), a1 = fmaxf(k1, __shfl_xor(s1, 16, 64));
                        float t = fmaxf(up8 ? a1 : a0, __shfl_xor(up8 ? a0 : a1, 8, 64));
                        t = fmaxf(t, __shfl_xor(t, 4, 64)); t = fmaxf(t, __shfl_xor(t, 2, 64)); t = fmaxf(t, __shfl_xor(t, 1, 64));
                        return t;
                    };
                    if (a.ds_part) {
                        const float t = bfly_sum(part);
                        if (writer) red[wn][mrow] = t;
                    }
                    if (seam) {
                        const float t = bfly_sum(dd), m = bfly_max(mx);
                        if (writer) { red[8 + wn][mrow] = t; red[12 + wn][mrow] = m; }
#pragma unroll
                        for (int c = 0; c < SEAM_MAXC; ++c)
                            if (c < imgc) {
                                const float u = bfly_sum(dwe[c]);
                                if (writer) red[20 + c * 4 + wn][mrow] = u;
                            }
                    }
                }
            }
            LA_ESTAMP(8);
            if (a.ds_part || seam) {
                __syncthreads();
                if (tid < MT) {
                    const long slot = ((long)b * a.M + m0 + tid) * a.tiles_per_sample + ntile;
                    for (int c = 0; c < imgc; ++c) {
                        float t = red[20 + c * 4][tid];
#pragma unroll
                        for (int w = 1; w < WN_; ++w) t += red[20 + c * 4 + w][tid];
                        a.seam_dweff_part[(((long)b * imgc + c) * a.M + m0 + tid) * a.tiles_per_sample + ntile] = t;
                    }
                    if (a.ds_part) {
                        float t = red[0][tid];
#pragma unroll
                        for (int w = 1; w < WN_; ++w) t += red[w][tid];
                        a.ds_part[slot] = t;
                    }
                    if (seam) {
                        float t = red[8][tid], m = red[12][tid];
#pragma unroll
                        for (int w = 1; w < WN_; ++w) { t += red[8 + w][tid]; m = fmaxf(m, red[12 + w][tid]); }
                        a.seam_ddn_part[slot] = t;
                        if (a.seam_pmax) a.seam_pmax[slot] = m;
                        red[12][tid] = m;
                    }
                }
                if (seam && a.seam_xs_out) {      // this workgroup's maximum lowers the running operand scale of the sample (wave 0)
                    __syncthreads();
                    if (tid < 64) {
                        float m = 0.f;
#pragma unroll
                        for (int k = tid; k < MT; k += 64) m = fmaxf(m, red[12][k]);
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                        if (tid == 0) la_xs_lower(a.seam_xs_out + (long)b * LA_XS_FAN + la_xs_sub(), xs_seen, a.seam_xs_mult, m);
                    }
                }
            }
            return;
        }
        const bool fwd = a.epi == LA_EPI_FWD;
        float nz[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) nz[j] = 0.f;
        if (fwd) {
            const float* dm_b = a.demod ? a.demod + (long)b * a.demod_stride : nullptr;
            if (tid < MT) {
                prm[tid] = dm_b ? dm_b[m0 + tid] : 1.f;
                prm[MT + tid] = a.bias ? a.bias[m0 + tid] : 0.f;
            }
            if (a.noise) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) nz[j] = a.noise[(long)b * a.noise_bstride + np[j]] * a.noise_strength;
            }
            LA_ESTAMP(6);
            __syncthreads();
            LA_ESTAMP(7);
        }
        // activation as straight-line selects (same arithmetic as la_act_fwd)
        const float slope = a.act == LA_ACT_LRELU ? a.alpha : (a.act == LA_ACT_RELU ? 0.f : 1.f);
        const float cl = a.clamp >= 0.f ? a.clamp : __builtin_huge_valf();
        // fused ToRGB (LaConvArgs::rgb_*): effective weights of this tile's rows in LDS rows 16.., per-lane partial sums over the
        // lane's 16 rows, then over the two half-waves and the waves along M (LDS, floats [wm][c][128 pixels] behind the row tables)
        constexpr int RGB_MAXC = 4;
        const int rgbc = (fwd && TILE2D && a.rgb_imgc > 0 && MT == a.M) ? a.rgb_imgc : 0;
        float* rgbp = &red[0][0] + 40 * MT;
        float pr[RGB_MAXC][NJ];
#pragma unroll
        for (int c = 0; c < RGB_MAXC; ++c)
#pragma unroll
            for (int j = 0; j < NJ; ++j) pr[c][j] = 0.f;
        if (rgbc > 0) {
            if (tid < MT) {
#pragma unroll
                for (int c = 0; c < RGB_MAXC; ++c)
                    red[16 + c][tid] = c < rgbc ? a.rgb_w[(long)c * a.M + tid] * a.rgb_s[(long)b * a.rgb_s_stride + tid] : 0.f;
            }
            __syncthreads();
        }
        float* o2 = (fwd && a.out2) ? a.out2 + ((long)b * a.M + m0 + mw) * HWo : nullptr;
        const float* ad = (o2 && a.addend) ? a.addend + ((long)b * a.M + m0 + mw) * HWo : nullptr;
        // operand scale of the output for its consumer (LaConvArgs::fwd_xs_out): this workgroup's max |y| lowers a sub-slot of the row
        float* fxs_row = (fwd && a.fwd_xs_out) ? a.fwd_xs_out + (long)b * LA_XS_FAN + la_xs_sub() : nullptr;
        const float fxs_seen = (fxs_row && tid == 0) ? la_xs_peek(fxs_row) : 0.f;      // (early: the round trip hides under the stores)
        float ymax = 0.f;
        typedef float f32x4p __attribute__((ext_vector_type(4)));
        // The value loop, instantiated per block-uniform case (raw / forward; second output with or without addend; fused ToRGB) so that the
        // case in hand is straight-line code (round 4 tested `fwd`, `o2`, `ad`, `rgbc > 0` and `c < rgbc` per VALUE: ~250 scalar branches per
        // lane-loop, their blocks laid out far apart); the row tables are read four rows at a time (16-byte LDS reads)
        auto value_loop = [&](auto fwd_c, auto o2_c, auto ad_c, auto rgb_c) {
            constexpr bool FWD = decltype(fwd_c)::value, O2 = decltype(o2_c)::value, AD = decltype(ad_c)::value, RGB = decltype(rgb_c)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int mg = i * 32 + 8 * g;
                    f32x4p dm4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f}, wr4[RGB_MAXC];
                    if constexpr (FWD) {
                        dm4 = *reinterpret_cast<const f32x4p*>(&prm[mw + mg]);
                        b4 = *reinterpret_cast<const f32x4p*>(&prm[MT + mw + mg]);
                    }
                    if constexpr (RGB) {
#pragma unroll
                        for (int c = 0; c < RGB_MAXC; ++c) wr4[c] = *reinterpret_cast<const f32x4p*>(&red[16 + c][mw + mg]);      // (rows c >= rgbc: zeros)
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = 4 * g + q, mr = mg + q;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            float v = acc[i][j][r];
                            if constexpr (FWD) {
                                v = v * dm4[q] + nz[j] + b4[q];
                                v = v > 0.f ? v : v * slope + 0.f;
                                v *= a.gain;
                                v = fminf(fmaxf(v, -cl), cl);
                            }
                            o0[(long)mr * HWo + np[j]] = v;
                            if constexpr (O2) {
                                float v2 = v;
                                if constexpr (AD) v2 += ad[(long)mr * HWo + np[j]];
                                o2[(long)mr * HWo + np[j]] = v2;
                                ymax = fmaxf(ymax, fabsf(v2));
                            } else ymax = fmaxf(ymax, fabsf(v));
                            if constexpr (RGB) {
#pragma unroll
                                for (int c = 0; c < RGB_MAXC; ++c) pr[c][j] += wr4[c][q] * v;
                            }
                        }
                    }
                }
            }
        };
        {
            typedef std::true_type T_; typedef std::false_type F_;
            if (!fwd) value_loop(F_{}, F_{}, F_{}, F_{});
            else if (rgbc > 0) { if (o2) { if (ad) value_loop(T_{}, T_{}, T_{}, T_{}); else value_loop(T_{}, T_{}, F_{}, T_{}); } else value_loop(T_{}, F_{}, F_{}, T_{}); }
            else if (o2) { if (ad) value_loop(T_{}, T_{}, T_{}, F_{}); else value_loop(T_{}, T_{}, F_{}, F_{}); }
            else value_loop(T_{}, F_{}, F_{}, F_{});
        }
        LA_ESTAMP(8);
        if (rgbc > 0) {
#pragma unroll
            for (int c = 0; c < RGB_MAXC; ++c)
                if (c < rgbc) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const float t = pr[c][j] + __shfl_xor(pr[c][j], 32, 64);      // rows of the other half-wave
                        if (lh == 0) rgbp[(wm * RGB_MAXC + c) * NT + (wn * NJ + j) * 32 + l31] = t;
                    }
                }
            __syncthreads();
            for (int k = tid; k < rgbc * NT; k += 256) {
                const int c = k / NT, tp = k - c * NT;
                float t = rgbp[c * NT + tp];
#pragma unroll
                for (int w = 1; w < WM_; ++w) t += rgbp[(w * RGB_MAXC + c) * NT + tp];
                t += a.rgb_bias ? a.rgb_bias[c] : 0.f;
                const int tpr = a.Gx >> 5;
                const int tyb = ntile / tpr, txb = ntile - tyb * tpr;
                const long pos = ((long)b * rgbc + c) * HWo + (long)(tyb * 4 + (tp >> 5)) * a.Wout + txb * 32 + (tp & 31);
                a.rgb_pre[pos] = t;
                if (a.rgb_clamp >= 0.f) t = fminf(fmaxf(t, -a.rgb_clamp), a.rgb_clamp);
                a.rgb_img[pos] = t + (a.rgb_skip ? a.rgb_skip[pos] : 0.f);
            }
        }
        if (fwd && a.fwd_xs_out) {      // (block-uniform)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
            __syncthreads();              // (the row tables / ToRGB partials of this tile are done with)
            if (lane == 0) red[0][wid] = ymax;
            __syncthreads();
            if (tid == 0) la_xs_lower(fxs_row, fxs_seen, a.fwd_xs_mult ? a.fwd_xs_mult[b] : 1.f, fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])));
        }
        return;
    }

    // ---- generic path (ragged tiles): every access guarded
    bool pix_ok[NJ];
    long npos[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int gy, gx;
        if (TILE2D) {
            const int tpr = a.Gx >> 5;                  // tiles per row of tiles
            const int tyb = ntile / tpr, txb = ntile - tyb * tpr;
            gy = tyb * 4 + wn * NJ + j; gx = txb * 32 + l31;
            pix_ok[j] = true;                           // the halo kernel only runs on grids that tile exactly
        } else {
            const int g = ntile * NT + (wn * NJ + j) * 32 + l31;
            pix_ok[j] = g < G;
            gy = pix_ok[j] ? g / a.Gx : 0;
            gx = pix_ok[j] ? g - gy * a.Gx : 0;
        }
        const int oy = gy * a.out_sy + a.out_oy, ox = gx * a.out_sx + a.out_ox;
        npos[j] = (long)oy * wpitch + ox;
    }
    const long HWout = oplane;
    float* out_b = a.out + (long)b * a.M * HWout;

    if (a.epi == LA_EPI_BWD) {
        const float* xin_b = a.xin ? a.xin + (long)b * a.xin_bstride : nullptr;
        const float* os_b = a.out_scale ? a.out_scale + (long)b * a.oscale_stride : nullptr;
        const bool seam = a.seam_ddn_part != nullptr && xin_b != nullptr;      // fused seam of the layer that produced xin (LaConvArgs)
        const float xs_seen = (seam && a.seam_xs_out) ? la_xs_peek(a.seam_xs_out + (long)b * LA_XS_FAN + la_xs_sub()) : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * (MT / WM_) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + ml;
                const bool mok = m < a.M;
                const float sc = (os_b && mok) ? os_b[m] : 1.f;
                const float dm0 = (seam && mok && a.seam_demod) ? a.seam_demod[(long)b * a.seam_demod_stride + m] : 1.f;
                const float b0 = (seam && mok && a.seam_bias) ? a.seam_bias[m] : 0.f;
                float part = 0.f, dd = 0.f, mx = 0.f;
                float dwe[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float v = acc[i][j][r];
                    if (mok && pix_ok[j]) {
                        const float y = xin_b ? xin_b[(long)m * HWout + npos[j]] : 0.f;
                        float g = v * sc;
                        if (seam) {
                            for (int c = 0; c < a.seam_imgc && c < 4; ++c) {
                                const long o = ((long)b * a.seam_imgc + c) * HWout + npos[j];
                                float gv = a.seam_gimg[o];
                                if (a.seam_rgb_clamp >= 0.f && fabsf(a.seam_rgbpre[o]) > a.seam_rgb_clamp) gv = 0.f;
                                g += a.seam_wrgb[(long)c * a.M + m] * a.seam_srgb[(long)b * a.seam_srgb_stride + m] * gv;
                                dwe[c] += gv * y;
                            }
                            const float nz0 = a.seam_noise ? a.seam_noise[(long)b * a.seam_noise_bstride + npos[j]] * a.seam_noise_strength : 0.f;
                            const float g1 = g * la_act_bwd_from_y(y, a.seam_act, a.seam_alpha, a.seam_gain, a.seam_clamp);
                            dd += g1 * (la_act_inv(y, a.seam_act, a.seam_alpha, a.seam_gain) - b0 - nz0);
                            g = g1 * dm0;
                            mx = fmaxf(mx, fabsf(g));
                        }
                        out_b[(long)m * HWout + npos[j]] = g;
                        part += v * y;
                    }
                }
                if (a.ds_part) {
                    // sum over the 32 lanes of this half-wave (they share m)
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                    if (l31 == 0) red[wn][ml] = part;
                }
                if (seam) {
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) { dd += __shfl_xor(dd, o, 64); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
                    if (l31 == 0) { red[8 + wn][ml] = dd; red[12 + wn][ml] = mx; }
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < a.seam_imgc) {
                            float t = dwe[c];
#pragma unroll
                            for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                            if (l31 == 0) red[20 + c * 4 + wn][ml] = t;
                        }
                }
            }
        }
        if (a.ds_part || seam) {
            __syncthreads();
            if (tid < MT && m0 + tid < a.M) {
                const long slot = ((long)b * a.M + m0 + tid) * a.tiles_per_sample + ntile;
                for (int c = 0; seam && c < a.seam_imgc && c < 4; ++c) {
                    float t = red[20 + c * 4][tid];
#pragma unroll
                    for (int w = 1; w < WN_; ++w) t += red[20 + c * 4 + w][tid];
                    a.seam_dweff_part[(((long)b * a.seam_imgc + c) * a.M + m0 + tid) * a.tiles_per_sample + ntile] = t;
                }
                if (a.ds_part) {
                    float t = red[0][tid];
#pragma unroll
                    for (int w = 1; w < WN_; ++w) t += red[w][tid];
                    a.ds_part[slot] = t;
                }
                if (seam) {
                    float t = red[8][tid], mm = red[12][tid];
#pragma unroll
                    for (int w = 1; w < WN_; ++w) { t += red[8 + w][tid]; mm = fmaxf(mm, red[12 + w][tid]); }
                    a.seam_ddn_part[slot] = t;
                    if (a.seam_pmax) a.seam_pmax[slot] = mm;
                    red[12][tid] = mm;
                }
            }
            if (seam && a.seam_xs_out) {
                __syncthreads();
                if (tid < 64) {
                    float m = 0.f;
                    for (int k = tid; k < MT; k += 64)
                        if (m0 + k < a.M) m = fmaxf(m, red[12][k]);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                    if (tid == 0) la_xs_lower(a.seam_xs_out + (long)b * LA_XS_FAN + la_xs_sub(), xs_seen, a.seam_xs_mult, m);
                }
            }
        }
        return;
    }

    const bool fwd = a.epi == LA_EPI_FWD;
    const float* dm_b = (fwd && a.demod) ? a.demod + (long)b * a.demod_stride : nullptr;
    float nz[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) nz[j] = 0.f;
    if (fwd && a.noise) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (pix_ok[j]) nz[j] = a.noise[(long)b * a.noise_bstride + npos[j]] * a.noise_strength;
    }
    float* fxs_row = (fwd && a.fwd_xs_out) ? a.fwd_xs_out + (long)b * LA_XS_FAN + la_xs_sub() : nullptr;
    const float fxs_seen = (fxs_row && tid == 0) ? la_xs_peek(fxs_row) : 0.f;
    float ymax = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (MT / WM_) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= a.M) continue;
            float dmv = 1.f, bv = 0.f;
            if (fwd) {
                if (dm_b) dmv = dm_b[m];
                if (a.bias) bv = a.bias[m];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (!pix_ok[j]) continue;
                float v = acc[i][j][r];
                if (fwd) v = la_conv_epi_fwd(a, v, dmv, nz[j], bv);
                out_b[(long)m * HWout + npos[j]] = v;
                if (fwd && a.out2) {
                    const long o2 = ((long)b * a.M + m) * HWout + npos[j];
                    v += a.addend ? a.addend[o2] : 0.f;
                    a.out2[o2] = v;
                }
                ymax = fmaxf(ymax, fabsf(v));
            }
        }
    }
    if (fwd && a.fwd_xs_out) {      // (block-uniform)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
        __syncthreads();
        if (lane == 0) red[0][wid] = ymax;
        __syncthreads();
        if (tid == 0) la_xs_lower(fxs_row, fxs_seen, a.fwd_xs_mult ? a.fwd_xs_mult[b] : 1.f, fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])));
    }
}
