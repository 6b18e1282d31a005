// Implicit-GEMM convolution, fp32 MFMA 32x32x2 (see la_conv.h).  Written for gfx950 only.
//
// Tiling: workgroup = 256 threads = 4 waves in a 2(m) x 2(n) grid; block tile MT(out channels) x 128(pixels);
// each wave owns (MT/2) x 64 as TM x 2 MFMA tiles of 32x32.  K is walked in chunks of (one tap, 16 input
// channels): the A slab [16][MT] comes straight from the tap-major packed weights (float4, coalesced), the B
// slab [16][128] is gathered from the NCHW input (lanes = consecutive pixels, coalesced) with the per-(b,c)
// modulation multiplied in on the way to LDS.  The next chunk's global loads are issued before the current
// chunk's MFMAs (register-staged prefetch); fp32 MFMA takes 64 cycles per instruction so the loader has slack.
#include "la_conv.h"
#include <stdlib.h>
#include "la_conv_device.h"
#include <type_traits>

#define KC 16

int la_conv_tiles_per_sample(int Gy, int Gx) { return la_cdiv((long)Gy * Gx, NT); }

#define SPLITK_MAX_G 1156     // up to 34x34 grids (covers the 33x33 phases of the 32 -> 64 up-sampling layer)
// K slices of a split-K launch over `tiles` (flattened-pixel tiles x row tiles) workgroups, each walking nck chunks x `taps`
// (chunk, tap) steps.  The 16-bit kernels keep 2 workgroups per CU = 512 resident slots.  Cost model in microseconds, fitted to
// the kernel trace of the config-f generator (profiles/): a step costs ~1.7 us of a resident slot, a workgroup ~10 us of
// prologue + epilogue, and slicing adds a finish pass that reads ks partial copies of the output and writes it once at
// ~3.5 TB/s (+5 us of launch).  E.g. 256 tiles x 144 steps -> 2 slices; the 4 merged 33x33 phases of the 32 -> 64 layer
// (1060 tiles x 36 steps, 69 MB of output) -> no slices: their finish pass would cost more than the tail it removes.
// (The fp32 kernel keeps 3 workgroups per CU: 768 slots, rounded up as before.)
static int choose_ksplit(bool bf, long tiles, int nck, float taps, double out_bytes) {
    int ks;
    if (bf) {
        float best = 1e30f;
        ks = 1;
        for (int k = 1; k <= nck && k <= 16; ++k) {
            const int per = la_cdiv(nck, k);
            const int kk = la_cdiv(nck, per);                       // slices actually launched
            if (kk != k) continue;
            const float slots = (float)(tiles * kk) / 512.f;
            const float rounds = slots <= 1.f ? 1.f : slots;          // (dynamic dispatch: beyond one round, work / slots)
            float cost = rounds * ((float)per * taps * 1.7f + 10.f);
            if (kk > 1) cost += 5.f + (float)((kk + 1) * out_bytes / 3.5e6);
            if (cost < best - 1e-3f) { best = cost; ks = kk; }
        }
        return ks;
    }
    ks = la_cdiv(768, tiles);
    if (ks > nck) ks = nck;
    if (ks < 2) return 1;
    const int per = la_cdiv(nck, ks);
    return la_cdiv(nck, per);
}

// scratch floats of the split-K form of a launch whose nphase (>= 1) output phases have the grids Gy[p] x Gx[p]; 0 if the launch
// would not be split
long la_conv_splitk_floats_phases(int B, int M, int C, int nphase, const int* Gy, const int* Gx, int precision) {
    const bool bf = precision != LA_PREC_F32;
    const int nck = la_cdiv(C, bf ? 32 : KC);
    if (nck < 2 || nphase < 1 || (nphase > 1 && !bf)) return 0;
    const int mtiles = la_cdiv(M, M >= 128 ? 128 : 64);
    long tiles = 0, Gsum = 0;
    for (int p = 0; p < nphase; ++p) {
        const long G = (long)Gy[p] * Gx[p];
        if (G > SPLITK_MAX_G) return 0;
        tiles += la_cdiv((long)B * G, NT);
        Gsum += G;
    }
    // (upper bound over the tap counts a launch of these grids may have: fewer taps never ask for more slices ... except that
    //  cheaper steps favour fewer slices; size for the densest case, 9 taps, and for the 9/4 of transposed-conv phases)
    int ks = choose_ksplit(bf, tiles * mtiles, nck, nphase > 1 ? 2.25f : 9.f, 4.0 * B * M * (double)Gsum);
    const int ks1 = choose_ksplit(bf, tiles * mtiles, nck, 1.f, 4.0 * B * M * (double)Gsum);
    if (ks1 > ks) ks = ks1;
    return ks >= 2 ? (long)ks * B * M * Gsum : 0;
}
long la_conv_splitk_floats(int B, int M, int C, int Gy, int Gx, int precision) {
    return la_conv_splitk_floats_phases(B, M, C, 1, &Gy, &Gx, precision);
}

// SPLIT = false: one workgroup owns a (sample, 128-pixel tile, MT-channel tile) and runs the whole K loop, epilogue fused.
// SPLIT = true : pixels of all samples are flattened (tiles may straddle samples), blockIdx.z walks K slices, raw
//                accumulators go to ws[slice][b][m][g]; la_conv_splitk_finish_kernel sums the slices and applies the epilogue.
//                Used for the <= 32x32 layers, whose K loop (up to 4608 deep) would otherwise serialise on a handful of CUs.
template <int MT, bool SPLIT>
__global__ __launch_bounds__(256) void la_conv_igemm_kernel(LaConvArgs a) {
    constexpr int TM = MT / 64;            // 32-row MFMA tiles per wave along m
    constexpr int A_F4 = (KC * MT / 4) / 256;  // float4 loads of the A slab per thread (2 for MT=128, 1 for MT=64)
    __shared__ __attribute__((aligned(16))) float As[2][KC][MT];
    __shared__ __attribute__((aligned(16))) float Bs[2][KC][NT];
    __shared__ float red[4][MT];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // XCD-aware tile order (direct mode): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
    // run of pixel tiles -- vertically adjacent tiles (which share the +-1 row halos of the 3x3 taps) then hit the same L2.
    int ntile = blockIdx.x;
    if (!SPLIT && (gridDim.x & 7) == 0) ntile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int m0 = blockIdx.y * MT;
    const int G = a.Gy * a.Gx;
    const int Ntot = SPLIT ? a.B * G : G;

    // ---- loader roles
    const int n_l = tid & (NT - 1);
    const int khalf = tid >> 7;
    const int nidx_l = ntile * NT + n_l;
    const bool nvalid = nidx_l < Ntot;
    int b_l = SPLIT ? (nvalid ? nidx_l / G : 0) : (int)blockIdx.z;
    const int g_l = SPLIT ? nidx_l - b_l * G : nidx_l;
    const int gy_l = nvalid ? g_l / a.Gx : 0;
    const int gx_l = nvalid ? g_l - gy_l * a.Gx : 0;
    const int iy0 = gy_l * a.in_sy, ix0 = gx_l * a.in_sx;
    const long HWin = (long)a.Hin * a.Win;
    const float* in_b = a.in + (long)b_l * a.in_bstride;
    const float* sc_b = a.in_scale ? a.in_scale + (long)b_l * a.scale_stride : nullptr;

    const int nck = (a.C + KC - 1) / KC;
    int ck_beg = 0, ck_end = nck;
    if (SPLIT) {
        const int per = (nck + a.ksplit - 1) / a.ksplit;
        ck_beg = blockIdx.z * per;
        ck_end = ck_beg + per < nck ? ck_beg + per : nck;
    }
    const int ci_beg = ck_beg * a.ntaps, ci_end = ck_end * a.ntaps;

    float breg[KC / 2];
    float4 areg[A_F4];

    auto prefetch = [&](int ci) {
        const int cc = ci / a.ntaps;
        const int t = ci - cc * a.ntaps;
        const int c0 = cc * KC;
        const int iy = iy0 + a.tap_dy[t], ix = ix0 + a.tap_dx[t];
        const bool ok = nvalid && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const long off = (long)iy * a.Win + ix;
#pragma unroll
        for (int j = 0; j < KC / 2; ++j) {
            const int c = c0 + khalf + 2 * j;
            float v = 0.f;
            if (ok && c < a.C) {
                v = in_b[(long)c * HWin + off];
                if (sc_b) v *= sc_b[c];
            }
            breg[j] = v;
        }
        const float* wslab = a.wgt + (long)a.tap_w[t] * a.C * a.M;
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int idx = tid + 256 * j;
            const int row = idx / (MT / 4), col = (idx - row * (MT / 4)) * 4;
            const int c = c0 + row, m = m0 + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < a.C && m < a.M) v = *reinterpret_cast<const float4*>(wslab + (long)c * a.M + m);
            areg[j] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < KC / 2; ++j) Bs[buf][khalf + 2 * j][n_l] = breg[j];
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int idx = tid + 256 * j;
            const int row = idx / (MT / 4), col = (idx - row * (MT / 4)) * 4;
            *reinterpret_cast<float4*>(&As[buf][row][col]) = areg[j];
        }
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    if (ci_beg < ci_end) {
        prefetch(ci_beg);
        stage(0);
    }
    __syncthreads();
    // double-buffered LDS, one barrier per chunk: loads for chunk i+1 are in flight while chunk i feeds the MFMAs
    for (int ci = ci_beg; ci < ci_end; ++ci) {
        const int buf = (ci - ci_beg) & 1;
        if (ci + 1 < ci_end) prefetch(ci + 1);
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            float av[TM], bv[2];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = As[buf][2 * kp + lh][wm * (MT / 2) + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bs[buf][2 * kp + lh][wn * 64 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (ci + 1 < ci_end) stage(buf ^ 1);
        __syncthreads();
    }

    la_conv_epilogue<MT, SPLIT>(a, acc, red, ntile, m0, G, Ntot);
}

// Split-K finisher: sums the K slices in a fixed order and applies the same epilogue as the direct kernel (deterministic).
// One wave per (b, m) plane, four planes per workgroup (the kernel is pure latency; with 16-byte loads a wave covers a 32x32 plane
// in four steps).
template <int PPB>      // planes per workgroup: 4 (one wave each) or 1
__global__ __launch_bounds__(256) void la_conv_splitk_finish_kernel(LaConvArgs a_in) {
    __shared__ float wpart[4];
    LaConvArgs a = a_in;
    if (a_in.nphase > 0) {        // merged phases: blockIdx.z = phase (grid, output offset and partials of that phase)
        const LaConvArgs::Phase& P = a_in.ph[blockIdx.z];
        a.Gy = P.Gy; a.Gx = P.Gx; a.out_oy = P.out_oy; a.out_ox = P.out_ox;
        a.splitk_ws = a_in.splitk_ws + P.ws_off;
    }
    constexpr int T = 256 / PPB;                    // threads per plane
    const int lane = threadIdx.x & 63;
    const int tp = threadIdx.x % T;
    const int m = blockIdx.x * PPB + threadIdx.x / T;
    const int b = blockIdx.y;
    const bool live = m < a.M;
    if (!live && PPB > 1) return;
    const int G = a.Gy * a.Gx;
    const int wpitch = (a.epi == LA_EPI_RAW && a.out_pitch > 0) ? a.out_pitch : a.Wout;
    const long HWout = (a.epi == LA_EPI_RAW && a.out_plane > 0) ? a.out_plane : (long)a.Hout * a.Wout;
    const long slice = (long)a.B * a.M * G;
    const float* wsp = a.splitk_ws + ((long)b * a.M + m) * G;
    float* out_p = a.out + ((long)b * a.M + m) * HWout;
    const float dmv = (a.epi == LA_EPI_FWD && a.demod) ? a.demod[(long)b * a.demod_stride + m] : 1.f;
    const float bv = (a.epi == LA_EPI_FWD && a.bias) ? a.bias[m] : 0.f;
    const float sc = (a.epi == LA_EPI_BWD && a.out_scale) ? a.out_scale[(long)b * a.oscale_stride + m] : 1.f;
    const float* xin_p = (a.epi == LA_EPI_BWD && a.xin) ? a.xin + (long)b * a.xin_bstride + (long)m * HWout : nullptr;
    // fused seam of the layer that produced xin (LaConvArgs::seam_*): same arithmetic as the direct kernels' epilogue
    const bool seam = a.epi == LA_EPI_BWD && a.seam_ddn_part != nullptr && xin_p != nullptr;
    const float dm0 = (seam && a.seam_demod) ? a.seam_demod[(long)b * a.seam_demod_stride + m] : 1.f;
    const float b0 = (seam && a.seam_bias) ? a.seam_bias[m] : 0.f;
    float part = 0.f, dd = 0.f, mx = 0.f;
    // (the sub-slot of the consumer's scale row this plane lowers, read early: the round trip hides under the slice loads)
    float* xs_row = (seam && a.seam_xs_out && lane == 0 && (PPB > 1 || threadIdx.x == 0)) ? a.seam_xs_out + (long)b * LA_XS_FAN + la_xs_sub((int)(threadIdx.x >> 6) * 5) : nullptr;
    // (forward epilogue: the same for LaConvArgs::fwd_xs_out, one atomic per wave)
    float ymax = 0.f;
    if (a.epi == LA_EPI_FWD && a.fwd_xs_out && lane == 0) xs_row = a.fwd_xs_out + (long)b * LA_XS_FAN + la_xs_sub((int)(threadIdx.x >> 6) * 5);
    const float xs_seen = xs_row ? la_xs_peek(xs_row) : 0.f;
    const int imgc = seam ? (a.seam_imgc < 4 ? a.seam_imgc : 4) : 0;
    float we[4] = {0.f, 0.f, 0.f, 0.f}, dwe[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < imgc; ++c) we[c] = a.seam_wrgb[(long)c * a.M + m] * a.seam_srgb[(long)b * a.seam_srgb_stride + m];
    // VEC consecutive grid positions per thread and step (4: 16-byte loads / stores when the rows allow it, else 1)
    auto run = [&](auto vec_tag) {
        constexpr int VEC = decltype(vec_tag)::value;
        typedef float vf __attribute__((ext_vector_type(VEC)));
        auto ld = [](const float* p) { return *reinterpret_cast<const vf*>(p); };
        for (int g = tp * VEC; g < G; g += T * VEC) {
            // (the slices' loads in flight eight at a time, added in slice order: with one load and one add per trip the pass was an
            //  exposed L2 latency per slice -- up to 16 of them at the 4x4 and 8x8 layers)
            // (sums written element by element: a vector add is v_pk_add_f32 -- no packed-FP32 arithmetic anywhere, Makefile)
            auto add = [](vf& s, const vf& p) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[e] += p[e];
            };
            vf acc = ld(wsp + g);
            int k = 1;
            for (; k + 7 < a.ksplit; k += 8) {
                vf p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = ld(wsp + (long)(k + u) * slice + g);
#pragma unroll
                for (int u = 0; u < 8; ++u) add(acc, p[u]);
            }
            if (k + 3 < a.ksplit) {
                vf p[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) p[u] = ld(wsp + (long)(k + u) * slice + g);
#pragma unroll
                for (int u = 0; u < 4; ++u) add(acc, p[u]);
                k += 4;
            }
            for (; k < a.ksplit; ++k) { const vf p = ld(wsp + (long)k * slice + g); add(acc, p); }
            const int gy = g / a.Gx, gx = g - gy * a.Gx;
            const long pos = (long)(gy * a.out_sy + a.out_oy) * wpitch + gx * a.out_sx + a.out_ox;
            float v[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = VEC == 1 ? acc[0] : acc[e];
            if (a.epi == LA_EPI_FWD) {
                vf nz = 0.f;
                if (a.noise) nz = ld(a.noise + (long)b * a.noise_bstride + pos);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = la_conv_epi_fwd(a, v[e], dmv, nz[e] * a.noise_strength, bv);
                if (a.out2) {
                    const long o2 = ((long)b * a.M + m) * HWout + pos;
                    vf ad = 0.f;
                    if (a.addend) ad = ld(a.addend + o2);
                    vf w;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { w[e] = v[e] + ad[e]; ymax = fmaxf(ymax, fabsf(w[e])); }
                    *reinterpret_cast<vf*>(a.out2 + o2) = w;
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ymax = fmaxf(ymax, fabsf(v[e]));
                }
            } else if (a.epi == LA_EPI_BWD) {
                vf y = 0.f;
                if (xin_p) y = ld(xin_p + pos);
                vf gv[4], nz0 = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    gv[c] = 0.f;
                    if (c < imgc) {
                        const long o = ((long)b * imgc + c) * HWout + pos;
                        gv[c] = ld(a.seam_gimg + o);
                        if (a.seam_rgb_clamp >= 0.f) {
                            const vf pre = ld(a.seam_rgbpre + o);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) if (fabsf(pre[e]) > a.seam_rgb_clamp) gv[c][e] = 0.f;
                        }
                    }
                }
                if (seam && a.seam_noise) {
                    nz0 = ld(a.seam_noise + (long)b * a.seam_noise_bstride + pos);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) nz0[e] *= a.seam_noise_strength;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    part += v[e] * y[e];
                    v[e] *= sc;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < imgc) { v[e] += we[c] * gv[c][e]; dwe[c] += gv[c][e] * y[e]; }
                    if (seam) {
                        const float g1 = v[e] * la_act_bwd_from_y(y[e], a.seam_act, a.seam_alpha, a.seam_gain, a.seam_clamp);
                        dd += g1 * (la_act_inv(y[e], a.seam_act, a.seam_alpha, a.seam_gain) - b0 - nz0[e]);
                        v[e] = g1 * dm0;
                        mx = fmaxf(mx, fabsf(v[e]));
                    }
                }
            }
            vf w;
#pragma unroll
            for (int e = 0; e < VEC; ++e) w[e] = v[e];
            *reinterpret_cast<vf*>(out_p + pos) = w;
        }
    };
    {
        size_t al = (size_t)a.splitk_ws | (size_t)a.out | (size_t)a.out2 | (size_t)a.addend | (size_t)a.noise | (size_t)a.xin;
        if (seam) al |= (size_t)a.seam_noise | (size_t)(imgc ? a.seam_gimg : nullptr) | (size_t)(imgc ? a.seam_rgbpre : nullptr);
        const bool vec4 = a.out_sx == 1 && ((a.Gx | wpitch | a.out_ox) & 3) == 0 && ((HWout | a.noise_bstride | a.seam_noise_bstride | a.xin_bstride) & 3) == 0 &&
                          (al & 15) == 0 && (a_in.nphase == 0 || (a_in.ph[blockIdx.z].ws_off & 3) == 0);
        if (vec4) run(std::integral_constant<int, 4>{});
        else run(std::integral_constant<int, 1>{});
    }
    if (a.epi == LA_EPI_FWD && a.fwd_xs_out) {      // (uniform; PPB == 1: every wave of the plane lowers its own sub-slot)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
        if (xs_row) la_xs_lower(xs_row, xs_seen, a.fwd_xs_mult ? a.fwd_xs_mult[b] : 1.f, ymax);
    }
    if (a.epi == LA_EPI_BWD && (a.ds_part || seam)) {
        part = la_wave_sum(part);
        dd = la_wave_sum(dd);
#pragma unroll
        for (int c = 0; c < 4; ++c) dwe[c] = la_wave_sum(dwe[c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (PPB == 1) {
            __shared__ float wdd[4], wmx[4], wdw[4][4];
            if (lane == 0) {
                const int w = threadIdx.x >> 6;
                wpart[w] = part; wdd[w] = dd; wmx[w] = mx;
#pragma unroll
                for (int c = 0; c < 4; ++c) wdw[c][w] = dwe[c];
            }
            __syncthreads();
            part = (wpart[0] + wpart[1]) + (wpart[2] + wpart[3]);
            dd = (wdd[0] + wdd[1]) + (wdd[2] + wdd[3]);
            mx = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
#pragma unroll
            for (int c = 0; c < 4; ++c) dwe[c] = (wdw[c][0] + wdw[c][1]) + (wdw[c][2] + wdw[c][3]);
        }
        // operand scale of the output for its consumer: one atomic per plane (wave) on a sub-slot of the sample's row
        if (xs_row) la_xs_lower(xs_row, xs_seen, a.seam_xs_mult, mx);
        if (tp < a.tiles_per_sample) {
            const long slot = ((long)b * a.M + m) * a.tiles_per_sample + tp;
            for (int c = 0; c < imgc; ++c)
                a.seam_dweff_part[(((long)b * imgc + c) * a.M + m) * a.tiles_per_sample + tp] = tp == 0 ? dwe[c] : 0.f;
            if (a.ds_part) a.ds_part[slot] = tp == 0 ? part : 0.f;
            if (seam) {
                a.seam_ddn_part[slot] = tp == 0 ? dd : 0.f;
                if (a.seam_pmax) a.seam_pmax[slot] = tp == 0 ? mx : 0.f;
            }
        }
    }
}


int la_conv_launch(const LaConvArgs& a, hipStream_t stream) {
    LA_CHECK_ARG(a.in && a.wgt && a.out, "conv: null pointer");
    LA_CHECK_ARG(a.B > 0 && a.C > 0 && a.M > 0 && a.Gy > 0 && a.Gx > 0, "conv: empty shape");
    LA_CHECK_ARG(a.M % 4 == 0, "conv: M must be a multiple of 4");
    LA_CHECK_ARG(a.ntaps >= 1 && a.ntaps <= LA_CONV_MAX_TAPS, "conv: bad tap count");
    LA_CHECK_ARG(a.in_sy >= 1 && a.in_sx >= 1 && a.out_sy >= 1 && a.out_sx >= 1, "conv: bad strides");
    // last grid point must land inside the output
    LA_CHECK_ARG((a.Gy - 1) * a.out_sy + a.out_oy < a.Hout && (a.Gx - 1) * a.out_sx + a.out_ox < a.Wout &&
                     a.out_oy >= 0 && a.out_ox >= 0,
                 "conv: output grid exceeds output tensor");
    LA_CHECK_ARG((long)a.Hout * a.Wout < (1L << 31) && (long)a.Hin * a.Win < (1L << 31), "conv: plane too large");
    if (a.epi == LA_EPI_BWD && a.ds_part) {
        LA_CHECK_ARG(a.tiles_per_sample == la_conv_tiles_per_sample(a.Gy, a.Gx), "conv: tiles_per_sample mismatch");
        LA_CHECK_ARG(a.out_sy == 1 && a.out_sx == 1 && a.out_oy == 0 && a.out_ox == 0, "conv: bwd epilogue needs dense output");
    }
    int tiles = la_conv_tiles_per_sample(a.Gy, a.Gx);
    const int nphase = a.nphase;
    LA_CHECK_ARG(nphase >= 0 && nphase <= LA_CONV_MAX_PHASES, "conv: bad phase count");
    if (nphase > 0) {
        LA_CHECK_ARG(a.precision != LA_PREC_F32 && a.epi == LA_EPI_RAW && a.in_q, "conv: merged phases need a pre-split 16-bit launch without epilogue");
        tiles = 0;
        for (int p = 0; p < nphase; ++p) {
            const LaConvArgs::Phase& P = a.ph[p];
            LA_CHECK_ARG(P.Gy > 0 && P.Gx > 0 && P.ntaps >= 1 && P.ntaps <= LA_CONV_PHASE_TAPS, "conv: bad phase");
            LA_CHECK_ARG((P.Gy - 1) * a.out_sy + P.out_oy < a.Hout && (P.Gx - 1) * a.out_sx + P.out_ox < a.Wout && P.out_oy >= 0 && P.out_ox >= 0,
                         "conv: phase grid exceeds output tensor");
            const int t = la_conv_tiles_per_sample(P.Gy, P.Gx);
            if (t > tiles) tiles = t;
        }
    }
    // split-K for the small-resolution layers (see kernel comment)
    LaConvArgs as = a;
    as.ksplit = 1;
    const long G = (long)a.Gy * a.Gx;
    const bool bf = a.precision != LA_PREC_F32;
    LA_CHECK_ARG(!bf || (a.wgt_bf16 && a.wgt_bf16_term_elems > 0), "conv: split-bf16 precision needs packed bf16 weights");
    const int nck = la_cdiv(a.C, bf ? 32 : KC);
    // 32-row tiles only exist for the halo kernel (the 32-channel layers of the 1024^2 generators)
    int MTsel = a.M >= 128 ? 128 : ((a.M <= 32 && bf && !a.in_q && la_conv_bf16_uses_halo(a)) ? 32 : 64);
    {   // dev knob (kernel experiments only): LA_FORCE_MT=64 runs the >= 128-row layers on 64-row tiles
        static const int force_mt = []() { const char* e = la_dev_env("LA_FORCE_MT"); return e ? atoi(e) : 0; }();
        if (force_mt == 64 && MTsel == 128) MTsel = 64;
    }
    const int mtiles = la_cdiv(a.M, MTsel);
    // scratch: pre-split input first (split precisions only), split-K partials after it
    if (bf) { int rc = la_conv_prepare_input(as, stream); if (rc) return rc; }
    // launch profiler: algorithmic work of this launch = 2*MACs; bytes = input read once + output written once (+ weights once)
    int pslot = -1, pcls = LA_PC_CONV_F32;
    double pflops = 0, pbytes = 0;
    if (la_prof_enabled()) {
        double gt = (double)a.Gy * a.Gx * a.ntaps, gout = (double)a.Gy * a.Gx, ntw = a.ntaps;
        if (nphase > 0) {
            gt = gout = ntw = 0.0;
            for (int p = 0; p < nphase; ++p) { gt += (double)a.ph[p].Gy * a.ph[p].Gx * a.ph[p].ntaps; gout += (double)a.ph[p].Gy * a.ph[p].Gx; ntw += a.ph[p].ntaps; }
        }
        if (a.row_hi > 0 && a.Gy > 0) {      // row window: only the wanted rows are work (tiles hold 4 rows / 128 flattened pixels: counted as wanted)
            int lo = a.row_lo > 0 ? a.row_lo : 0, hi = a.row_hi < a.Gy ? a.row_hi : a.Gy;
            lo &= ~3; hi = (hi + 3) & ~3; if (hi > a.Gy) hi = a.Gy;
            double f = hi > lo ? (double)(hi - lo) / a.Gy : 0.0;
            if (a.col_hi > 0 && a.Gx > 0 && la_conv_bf16_uses_halo(a)) f *= (double)((((a.col_hi < a.Gx ? a.col_hi : a.Gx) + 31) & ~31) - (a.col_lo & ~31)) / a.Gx;
            gt *= f; gout *= f;
        }
        pflops = 2.0 * a.B * gt * a.M * (double)a.C;
        pbytes = 4.0 * ((double)a.B * a.C * a.Hin * a.Win * (a.in_bstride ? 1.0 : 1.0 / a.B) + (double)a.B * a.M * gout + ntw * a.C * a.M);
    }
    if (as.seam_xs_out) as.seam_pmax = nullptr;      // every kernel form lowers the consumer's slot row itself: no plane maxima
    as.splitk_ws = nullptr;
    long splitk_floats = 0;
    if (as.ws && as.ws_bytes >= sizeof(float)) {
        as.splitk_ws = reinterpret_cast<float*>(as.ws);
        splitk_floats = (long)(as.ws_bytes / sizeof(float));
    }
    bool small = G <= SPLITK_MAX_G && !(bf && la_conv_bf16_uses_halo(as));      // (the halo kernel claims what it can run directly)
    long Gsum = G, Gmax = G, tiles_flat = la_cdiv((long)a.B * G, NT);
    if (nphase > 0) {
        Gsum = 0; Gmax = 0; tiles_flat = 0;
        for (int p = 0; p < nphase; ++p) {
            const long Gp = (long)a.ph[p].Gy * a.ph[p].Gx;
            small = small && Gp <= SPLITK_MAX_G;
            Gsum += Gp; if (Gp > Gmax) Gmax = Gp;
            as.ph[p].tile0 = (int)tiles_flat;
            tiles_flat += la_cdiv((long)a.B * Gp, NT);
        }
        small = small && bf;       // (merged phases exist for the 16-bit kernels only)
    }
    if (as.splitk_ws && small && nck >= 2) {
        const int ntiles_flat = (int)tiles_flat;
        float taps = (float)a.ntaps;
        if (nphase > 0) { taps = 0.f; for (int p = 0; p < nphase; ++p) taps += (float)a.ph[p].ntaps * ((float)a.ph[p].Gy * a.ph[p].Gx / (float)Gsum); }
        int ks = choose_ksplit(bf, (long)ntiles_flat * mtiles, nck, taps, 4.0 * a.B * a.M * (double)Gsum);
        if (la_dev_knob(LA_KNOB_KSPLIT) > 0 && la_dev_knob(LA_KNOB_KSPLIT) <= nck) { const int per = la_cdiv(nck, la_dev_knob(LA_KNOB_KSPLIT)); ks = la_cdiv(nck, per); }
        {
            if ((long)ks * a.B * a.M * Gsum <= splitk_floats && ks >= 2) {
                as.ksplit = ks;
                long off = 0;
                for (int p = 0; p < nphase; ++p) { as.ph[p].ws_off = off; off += (long)ks * a.B * a.M * a.ph[p].Gy * a.ph[p].Gx; }
                dim3 grid(ntiles_flat, mtiles, ks);
                pslot = la_prof_open(bf ? LA_PC_CONV_SPLITK : LA_PC_CONV_F32, pflops, pbytes, stream);
                if (bf) { int rc = la_conv_bf16_dispatch(as, MTsel, grid, true, stream); if (rc) return rc; }
                else if (MTsel == 128) hipLaunchKernelGGL((la_conv_igemm_kernel<128, true>), grid, dim3(256), 0, stream, as);
                else hipLaunchKernelGGL((la_conv_igemm_kernel<64, true>), grid, dim3(256), 0, stream, as);
                const unsigned nz = nphase > 0 ? nphase : 1;
                // one wave per (b, m) plane, four planes per workgroup, at every split-K size (<= 34x34): a whole workgroup per plane
                // (the <1> form) measured 21 against 13 us on the 8 x 512 x 32^2 launches
                hipLaunchKernelGGL(la_conv_splitk_finish_kernel<4>, dim3(la_cdiv(a.M, 4), a.B, nz), dim3(256), 0, stream, as);
                // (the consumer's operand scale, LaConvArgs::seam_xs_out: the finish workgroups lower the sample's slot row themselves)
            }
        }
    }
    if (as.ksplit == 1) {
        // a windowed 16-bit launch holds the window's tiles only (rounded up to a multiple of 8: one run of tiles per XCD); the kernels
        // derive the same counts from row_lo / row_hi / col_lo / col_hi
        if (bf && a.row_hi > 0) {
            int nt = 0;
            if (la_conv_bf16_uses_halo(as)) {
                const int tpr = a.Gx >> 5, r0 = a.row_lo >> 2, r1 = ((a.row_hi < a.Gy ? a.row_hi : a.Gy) + 3) >> 2;
                int cw = tpr;
                if (a.col_hi > 0) cw = (((a.col_hi < a.Gx ? a.col_hi : a.Gx) + 31) >> 5) - (a.col_lo >> 5);
                nt = (r1 - r0) * cw;
            } else {
                for (int p = 0; p < (nphase > 0 ? nphase : 1); ++p) {
                    const int Gy = nphase > 0 ? a.ph[p].Gy : a.Gy, Gx = nphase > 0 ? a.ph[p].Gx : a.Gx;
                    const int tall = la_cdiv(Gy * Gx, NT), t0 = (a.row_lo * Gx) / NT;
                    int t1 = la_cdiv((a.row_hi < Gy ? a.row_hi : Gy) * Gx, NT);
                    if (t1 > tall) t1 = tall;
                    if (t1 - t0 > nt) nt = t1 - t0;
                }
            }
            const int n8 = (nt + 7) & ~7;
            if (nt > 0 && n8 < tiles) tiles = n8;
        }
#ifdef LA_DEV
        as.dbg_stamp = la_dev_knob(LA_KNOB_HALO_STAMP);
#endif
        dim3 grid(tiles, mtiles, a.B * (nphase > 0 ? nphase : 1));
        if (bf) pcls = la_conv_bf16_uses_halo(as) ? LA_PC_CONV_HALO : LA_PC_CONV_FLAT;
        pslot = la_prof_open(pcls, pflops, pbytes, stream);
        if (bf) { int rc = la_conv_bf16_dispatch(as, MTsel, grid, false, stream); if (rc) return rc; }
        else if (MTsel == 128) hipLaunchKernelGGL((la_conv_igemm_kernel<128, false>), grid, dim3(256), 0, stream, as);
        else hipLaunchKernelGGL((la_conv_igemm_kernel<64, false>), grid, dim3(256), 0, stream, as);
    }
    LA_CHECK_LAUNCH();
    la_prof_close(pslot, stream);
    return LA_OK;
}
