// upfirdn2d: zero-insert, pad/crop, 2-D FIR, decimate, gain  (+ optional fused modconv epilogue).
// Semantics follow models/stylegan3/torch_utils/ops/upfirdn2d.py:167-211 (ref) / upfirdn2d.cu:29-92.
// HBM-bound: each output reads <= ceil(fh/up)*ceil(fw/up) inputs (L1/L2 hits), one coalesced store.
#include "la_upfirdn2d.h"
#include <atomic>
#include <stdlib.h>

#define FIR_MAX 8

struct FirArgs {
    const float* in;
    float* out;
    int P, C;          // planes = B*C
    int Hin, Win, Hout, Wout;
    int upx, upy, dnx, dny, padx0, pady0;
    int fw, fh;
    float f[FIR_MAX * FIR_MAX];  // effective correlation kernel (already flipped as needed, gain folded in)
    int epi;                     // 0 plain, 1 modconv epilogue, 2 activation backward: out = fir(in) * act'(yref) (4x4 stride-1 scalar kernel)
    const float* yref;           // epi 2: saved output of the activation, same shape as `out`
    const float* demod; int demod_stride;   // [B][stride]
    const float* noise; long noise_bstride; float noise_strength;
    const float* bias;
    int act; float alpha, gain, clamp;
    const float* addend;         // optional same-shape tensor added to the result (skip connection), epi 0 only
    float* pmax;                 // optional [P][la_fir4x4_segments(Hout, Wout)]: partial max |out| per plane, 4x4 stride-1 kernel only
    int in_pitch; long in_plane; // 0 = dense; row pitch / plane stride of `in` in floats (vector kernel: multiples of 4)
    int in_xhalf;                // > 0: COLUMN-PLANAR rows -- the even columns of a row at [0, ceil(Win/2)), the odd ones from in_xhalf on
    float* xs_out;               // optional (4x4 stride-1 kernels): slot rows [B][LA_XS_FAN] of the fp16 operand scale of `out` for the
    const float* xs_mult;        //   contraction that consumes it: every wave lowers its sample's row to pow2(xs_mult[b] * its max |out|)
    int row_lo, row_hi;          // row window (planar vector kernel; 0 / 0 = all): only output rows [row_lo, row_hi) are computed and written
    int col_lo, col_hi;          // ... and only the 4-column groups that hold a column of [col_lo, col_hi) (0 / 0 = all)
};

__global__ __launch_bounds__(256) void la_upfirdn2d_kernel(FirArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.Wout || y >= a.Hout) return;
    // contributing input rows: iy*upy = y*dny + ta - pady0, ta in [0, fh)
    const int by = y * a.dny - a.pady0, bx = x * a.dnx - a.padx0;
    // smallest iy with iy*up >= by  (floor division that is safe for negatives)
    int iy_lo = (by >= 0) ? (by + a.upy - 1) / a.upy : -((-by) / a.upy);
    int ix_lo = (bx >= 0) ? (bx + a.upx - 1) / a.upx : -((-bx) / a.upx);
    const long HWin = (long)a.Hin * a.Win, HWout = (long)a.Hout * a.Wout;
    for (int p = blockIdx.z; p < a.P; p += gridDim.z) {
        const float* ip = a.in + (long)p * HWin;
        float v = 0.f;
        for (int iy = iy_lo; iy * a.upy - by < a.fh; ++iy) {
            if (iy < 0 || iy >= a.Hin) continue;
            const int ta = iy * a.upy - by;
            for (int ix = ix_lo; ix * a.upx - bx < a.fw; ++ix) {
                if (ix < 0 || ix >= a.Win) continue;
                const int tb = ix * a.upx - bx;
                v += ip[(long)iy * a.Win + ix] * a.f[ta * a.fw + tb];
            }
        }
        const long o = (long)p * HWout + (long)y * a.Wout + x;
        if (a.epi == 1) {
            const int b = p / a.C, c = p - b * a.C;
            if (a.demod) v *= a.demod[(long)b * a.demod_stride + c];
            if (a.noise) v += a.noise[(long)b * a.noise_bstride + (long)y * a.Wout + x] * a.noise_strength;
            if (a.bias) v += a.bias[c];
            v = la_act_fwd(v, a.act, a.alpha, a.gain, a.clamp);
        } else if (a.addend) {
            v += a.addend[o];
        }
        a.out[o] = v;
    }
}

// Specialisation for the hot case up = down = 1 with a 4x4 filter (FIR after the transposed conv, and its adjoint):
// each thread produces 8 vertically consecutive outputs of one column from an 11 x 4 register window, taps fully unrolled.
// Lanes = 64 consecutive columns -> every load / store is a coalesced 256-B row segment; the +-3 halo re-reads hit L1/L2.
#define FIR_ROWS 8
template <int EPI>
__device__ __forceinline__ float la_fir4x4_plane(const FirArgs& a, const float (&f)[16], const bool (&xok)[4], int p, int x, int y0, int ix0,
                                                int iy0, long HWin, long HWout) {
    const float* ip = a.in + (long)p * HWin;
    float acc[FIR_ROWS];
#pragma unroll
    for (int r = 0; r < FIR_ROWS; ++r) acc[r] = 0.f;
#pragma unroll
    for (int wr = 0; wr < FIR_ROWS + 3; ++wr) {
        const int iy = iy0 + wr;
        float v[4];
        const bool yok = iy >= 0 && iy < a.Hin;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (yok && xok[j]) ? ip[(long)iy * a.Win + ix0 + j] : 0.f;
#pragma unroll
        for (int r = 0; r < FIR_ROWS; ++r) {
            const int ta = wr - r;          // filter row used by output row r
            if (ta >= 0 && ta < 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r] += v[j] * f[ta * 4 + j];
            }
        }
    }
    float dm = 1.f, bv = 0.f;
    if (EPI == 1) {
        const int b = p / a.C, c = p - b * a.C;
        if (a.demod) dm = a.demod[(long)b * a.demod_stride + c];
        if (a.bias) bv = a.bias[c];
    }
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < FIR_ROWS; ++r) {
        const int y = y0 + r;
        if (y >= a.Hout) break;
        float v = acc[r];
        const long pos = (long)y * a.Wout + x;
        if (EPI == 1) {
            v = v * dm + bv;
            if (a.noise) v += a.noise[(long)(p / a.C) * a.noise_bstride + pos] * a.noise_strength;
            v = la_act_fwd(v, a.act, a.alpha, a.gain, a.clamp);
        } else if (EPI == 2) {
            v *= la_act_bwd_from_y(a.yref[(long)p * HWout + pos], a.act, a.alpha, a.gain, a.clamp);
        } else if (a.addend) {
            v += a.addend[(long)p * HWout + pos];
        }
        a.out[(long)p * HWout + pos] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    return mx;
}

template <int EPI>
__global__ __launch_bounds__(256) void la_fir4x4_s1_kernel(FirArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * FIR_ROWS;
    const bool live = x < a.Wout && y0 < a.Hout;
    if (!live && !a.pmax && !a.xs_out) return;          // (with plane maxima / a scale slot, dead lanes wait for the reduction)
    const long HWin = (long)a.Hin * a.Win, HWout = (long)a.Hout * a.Wout;
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = a.f[i];
    const int ix0 = x - a.padx0, iy0 = y0 - a.pady0;
    bool xok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xok[j] = (ix0 + j) >= 0 && (ix0 + j) < a.Win;
    for (int p = blockIdx.z; p < a.P; p += gridDim.z) {
        float mx = 0.f;
        if (live) mx = la_fir4x4_plane<EPI>(a, f, xok, p, x, y0, ix0, iy0, HWin, HWout);
        if (a.pmax || a.xs_out) {      // one partial maximum per (plane, workgroup): plain store, reduced by the consumer (no atomics) ...
            __shared__ float wmax[4];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            __syncthreads();
            if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
            __syncthreads();
            if (threadIdx.x == 0) {
                const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
                if (a.pmax) a.pmax[(long)p * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x] = m;
                if (a.xs_out) {      // ... or lowers a sub-slot of the sample's scale row for the consuming contraction (FirArgs::xs_out)
                    const int bs = p / a.C;
                    float* row = a.xs_out + (long)bs * LA_XS_FAN + la_xs_sub(p);
                    la_xs_lower(row, la_xs_peek(row), a.xs_mult ? a.xs_mult[bs] : 1.f, m);
                }
            }
        }
    }
}

static int fir_fill(FirArgs& a, const float* in, float* out, int B, int C, int Hin, int Win, const float* f_host,
                    int fh, int fw, int upx, int upy, int dnx, int dny, int padx0, int padx1, int pady0, int pady1,
                    int flip_filter, float gain, int* Hout, int* Wout) {
    LA_CHECK_ARG(in && out && f_host, "upfirdn2d: null pointer");
    // (up to 8 x 8 taps, or one separable pass of up to 32: 1 x fw / fh x 1 -- the two-pass form of upfirdn2d.py:188-201 for 1-D filters)
    LA_CHECK_ARG(fh >= 1 && fw >= 1 && fh * fw <= FIR_MAX * FIR_MAX && fh <= 32 && fw <= 32, "upfirdn2d: filter larger than 8x8 (or than 32 taps in one separable pass)");
    LA_CHECK_ARG(upx >= 1 && upy >= 1 && dnx >= 1 && dny >= 1, "upfirdn2d: bad up/down factor");
    LA_CHECK_ARG(B >= 1 && C >= 1 && Hin >= 1 && Win >= 1, "upfirdn2d: empty input");
    const int upW = Win * upx + padx0 + padx1, upH = Hin * upy + pady0 + pady1;
    LA_CHECK_ARG(upW >= fw && upH >= fh, "upfirdn2d: upsampled image smaller than the filter");
    *Wout = (upW - fw + dnx) / dnx;   // upfirdn2d.cpp:35-36
    *Hout = (upH - fh + dny) / dny;
    a.in = in; a.out = out; a.P = B * C; a.C = C; a.pmax = nullptr; a.xs_out = nullptr; a.xs_mult = nullptr; a.row_lo = a.row_hi = 0; a.col_lo = a.col_hi = 0;
    a.in_pitch = Win; a.in_plane = (long)Hin * Win; a.in_xhalf = 0;
    a.Hin = Hin; a.Win = Win; a.Hout = *Hout; a.Wout = *Wout;
    a.upx = upx; a.upy = upy; a.dnx = dnx; a.dny = dny; a.padx0 = padx0; a.pady0 = pady0;
    a.fw = fw; a.fh = fh;
    // the op is a true convolution unless flip_filter: correlation kernel = flipped filter (upfirdn2d.py:198-199)
    for (int i = 0; i < fh; ++i)
        for (int j = 0; j < fw; ++j)
            a.f[i * fw + j] = gain * (flip_filter ? f_host[i * fw + j] : f_host[(fh - 1 - i) * fw + (fw - 1 - j)]);
    a.epi = 0; a.demod = nullptr; a.noise = nullptr; a.bias = nullptr; a.addend = nullptr; a.yref = nullptr;
    a.demod_stride = 0; a.noise_bstride = 0; a.noise_strength = 0.f;
    a.act = LA_ACT_LINEAR; a.alpha = 0.f; a.gain = 1.f; a.clamp = -1.f;
    return LA_OK;
}

// Vector form of the 4x4 stride-1 pad-1 FIR (+ layer epilogue) for a SEPARABLE filter f = fy (x) fx (setup_filter's outer product)
// on COLUMN-PLANAR input rows (FirArgs::in_xhalf): the transposed stride-2 conv that produces the (2h+1)-wide intermediate writes
// each output phase as contiguous runs (even columns | odd columns) instead of 4-byte stores at stride 8 (128 -> 313 us on the
// 8x128x257^2 launch), and this kernel re-interleaves on the fly.
//     out[y][x] = sum_{a,b} fy[a] fx[b] * in[y + a - 1][x + b - 1]
// Work item = (plane, ROWS-row strip, 4-column group), groups fastest, so that small planes share a workgroup; per input row two
// 8-byte loads (even / odd plane) + three neighbour dwords (L1 hits), 7 columns -> 4 horizontal results, then a vertical 4-tap
// pass over a rolling window of the last four horizontal rows with two input rows in flight: 8 MACs per output (the scalar kernel
// above: 16 MACs and 5.5 dword loads per output).  Measured at 8x128x256^2: 133 us; an 8-column form with 16-byte loads (113
// registers) 147 us; the same kernel on column-interleaved rows with three 16-byte loads per row 127 us.
template <int EPI, int ROWS>
__global__ __launch_bounds__(256) void la_fir4x4_s1p_kernel(FirArgs a, float4 fx, float4 fy) {
    __shared__ unsigned smx[64];                   // xs_out: max |out| per sample of this workgroup (bit patterns: non-negative floats order like unsigned ints)
    // Work items = (plane, row strip, 4-column group) of the WANTED part of the output only (round 5): with a row / column window
    // (FirArgs::row_lo / col_lo) the grid holds the strips and column groups that contain a wanted row / column -- round 4 kept the
    // whole plane's items and let those outside return, which left the launch as long as the whole-frame one (123 us for 56 % of the
    // bytes at 256^2: the live lanes were spread thinly over the same number of waves).
    const int w4 = a.Wout >> 2;
    const int strips = (a.Hout + ROWS - 1) / ROWS;
    int s_lo = 0, s_hi = strips, g_lo = 0, g_hi = w4;
    const int ybase = a.row_hi > 0 ? a.row_lo : 0;      // strips are counted from the window's first row: no nearly empty strip at its top
    if (a.row_hi > 0) s_hi = ((a.row_hi < a.Hout ? a.row_hi : a.Hout) - ybase + ROWS - 1) / ROWS;
    if (a.col_hi > 0) { g_lo = a.col_lo >> 2; g_hi = ((a.col_hi < a.Wout ? a.col_hi : a.Wout) + 3) >> 2; }
    const int gw = g_hi - g_lo;
    const int per_plane = gw * (s_hi - s_lo);
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int p = (int)(gid / per_plane);
    const bool live = p < a.P;
    if (!live && !a.xs_out) return;
    const int b = (live ? p : a.P - 1) / a.C;
    float omax = 0.f;
    if (a.xs_out) {
        if (threadIdx.x < 64) smx[threadIdx.x] = 0u;
        __syncthreads();
    }
    if (live) {
        const int within = (int)(gid - (long)p * per_plane);
        const int strip = s_lo + within / gw, xg = g_lo + (within - (within / gw) * gw);
        const int x0 = xg * 4, q2 = xg * 2;
        // row window (FirArgs::row_lo, 0 / 0 = all): the strip shrinks to the wanted rows, a strip without one does nothing
        const int ys = ybase + strip * ROWS;
        const int y0 = ys;
        const long HWout = (long)a.Hout * a.Wout;
        const int ne = (a.Win + 1) >> 1, no = a.Win >> 1;
        const bool okm = q2 > 0, oke = q2 + 2 < ne, oko = q2 + 2 < no;
        struct Row { float2 e, o; float om1, e2, o2; };
        const float* ip = a.in + (long)p * a.in_plane + q2;
        auto load_row = [&](int iy) {
            Row r;
            r.e = r.o = make_float2(0.f, 0.f);
            r.om1 = r.e2 = r.o2 = 0.f;
            if (iy >= 0 && iy < a.Hin) {
                const float* rp = ip + (long)iy * a.in_pitch;
                r.e = *reinterpret_cast<const float2*>(rp);
                r.o = *reinterpret_cast<const float2*>(rp + a.in_xhalf);
                if (okm) r.om1 = rp[a.in_xhalf - 1];
                if (oke) r.e2 = rp[2];
                if (oko) r.o2 = rp[a.in_xhalf + 2];
            }
            return r;
        };
        auto hpass = [&](const Row& r) {
            // image columns x0 - 1 .. x0 + 5
            const float c[7] = {r.om1, r.e.x, r.o.x, r.e.y, r.o.y, r.e2, r.o2};
            float4 h;
            h.x = fx.x * c[0] + fx.y * c[1] + fx.z * c[2] + fx.w * c[3];
            h.y = fx.x * c[1] + fx.y * c[2] + fx.z * c[3] + fx.w * c[4];
            h.z = fx.x * c[2] + fx.y * c[3] + fx.z * c[4] + fx.w * c[5];
            h.w = fx.x * c[3] + fx.y * c[4] + fx.z * c[5] + fx.w * c[6];
            return h;
        };
        float dm = 1.f, bv = 0.f;
        if (EPI == 1) {
            const int c = p - b * a.C;
            if (a.demod) dm = a.demod[(long)b * a.demod_stride + c];
            if (a.bias) bv = a.bias[c];
        }
        const float* nzp = (EPI == 1 && a.noise) ? a.noise + (long)b * a.noise_bstride : nullptr;
        const Row r0 = load_row(y0 - 1), r1 = load_row(y0), r2 = load_row(y0 + 1);
        Row n0 = load_row(y0 + 2), n1 = load_row(y0 + 3);
        float4 h0 = hpass(r0), h1 = hpass(r1), h2 = hpass(r2);
        int y1 = ys + ROWS < a.Hout ? ys + ROWS : a.Hout;
        if (a.row_hi > 0 && a.row_hi < y1) y1 = a.row_hi;
#pragma unroll 4
        for (int y = y0; y < y1; ++y) {
            const float4 h3 = hpass(n0);
            n0 = n1;
            n1 = load_row(y + 4);
            float4 o;
            o.x = fy.x * h0.x + fy.y * h1.x + fy.z * h2.x + fy.w * h3.x;
            o.y = fy.x * h0.y + fy.y * h1.y + fy.z * h2.y + fy.w * h3.y;
            o.z = fy.x * h0.z + fy.y * h1.z + fy.z * h2.z + fy.w * h3.z;
            o.w = fy.x * h0.w + fy.y * h1.w + fy.z * h2.w + fy.w * h3.w;
            h0 = h1; h1 = h2; h2 = h3;
            const long pos = (long)y * a.Wout + x0;
            if (EPI == 1) {
                float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nzp) {
                    nz = *reinterpret_cast<const float4*>(nzp + pos);
                    nz.x *= a.noise_strength; nz.y *= a.noise_strength; nz.z *= a.noise_strength; nz.w *= a.noise_strength;
                }
                o.x = la_act_fwd(o.x * dm + bv + nz.x, a.act, a.alpha, a.gain, a.clamp);
                o.y = la_act_fwd(o.y * dm + bv + nz.y, a.act, a.alpha, a.gain, a.clamp);
                o.z = la_act_fwd(o.z * dm + bv + nz.z, a.act, a.alpha, a.gain, a.clamp);
                o.w = la_act_fwd(o.w * dm + bv + nz.w, a.act, a.alpha, a.gain, a.clamp);
            } else if (EPI == 2) {      // activation backward from the saved output at this position (FirArgs::yref)
                const float4 yr = *reinterpret_cast<const float4*>(a.yref + (long)p * HWout + pos);
                o.x *= la_act_bwd_from_y(yr.x, a.act, a.alpha, a.gain, a.clamp);
                o.y *= la_act_bwd_from_y(yr.y, a.act, a.alpha, a.gain, a.clamp);
                o.z *= la_act_bwd_from_y(yr.z, a.act, a.alpha, a.gain, a.clamp);
                o.w *= la_act_bwd_from_y(yr.w, a.act, a.alpha, a.gain, a.clamp);
            } else if (a.addend) {
                const float4 ad = *reinterpret_cast<const float4*>(a.addend + (long)p * HWout + pos);
                o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
            }
            *reinterpret_cast<float4*>(a.out + (long)p * HWout + pos) = o;
            omax = fmaxf(omax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        }
    }
    if (a.xs_out) {
        // operand scale of `out` for its consumer (FirArgs::xs_out): per-sample maxima of the workgroup through LDS (the workgroup's 256
        // work items are consecutive planes, i.e. a handful of consecutive samples), then one thread per sample lowers a sub-slot of
        // that sample's row
        const int b0 = (int)(((long)blockIdx.x * 256) / per_plane) / a.C;      // first sample of the workgroup
        const int bi = b - b0;
        if (live && omax > 0.f) {
            if (bi < 64) atomicMax(&smx[bi], __float_as_uint(omax));
            else { float* row = a.xs_out + (long)b * LA_XS_FAN + la_xs_sub(threadIdx.x); la_xs_lower(row, la_xs_peek(row), a.xs_mult ? a.xs_mult[b] : 1.f, omax); }
        }
        __syncthreads();
        const int bt = b0 + (int)threadIdx.x;
        if (threadIdx.x < 64 && smx[threadIdx.x] != 0u && bt * a.C < a.P) {
            float* row = a.xs_out + (long)bt * LA_XS_FAN + la_xs_sub();
            la_xs_lower(row, la_xs_peek(row), a.xs_mult ? a.xs_mult[bt] : 1.f, __uint_as_float(smx[threadIdx.x]));
        }
    }
}

// f (4x4, effective correlation taps) == fy (x) fx ?  (rank one, as setup_filter's outer product is; tolerance 1e-6 relative)
static bool fir_separable(const float* f, float* fx, float* fy) {
    int pi = 0, pj = 0;
    float best = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (fabsf(f[i * 4 + j]) > best) { best = fabsf(f[i * 4 + j]); pi = i; pj = j; }
    if (best == 0.f) return false;
    for (int j = 0; j < 4; ++j) fx[j] = f[pi * 4 + j];
    for (int i = 0; i < 4; ++i) fy[i] = f[i * 4 + pj] / f[pi * 4 + pj];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (fabsf(fy[i] * fx[j] - f[i * 4 + j]) > 1e-6f * best) return false;
    return true;
}

int la_fir4x4_segments(int Hout, int Wout) { return la_cdiv(Wout, 64) * la_cdiv(Hout, 4 * FIR_ROWS); }

// 4x4 FIR around a factor-2 resampling of few, possibly large planes (the image pyramid of the skip architecture: upsample2d
// forward, upfirdn2d.py:342-348, and its adjoint-by-decimation in the backward pass).  Same taps, same accumulation order (rows,
// then columns, ascending) as the generic kernel above, without its per-output tap search; work item = 2 x 4 (up) / 1 x 2 (down)
// outputs, flat over all planes.
__global__ __launch_bounds__(256) void la_fir4x4_up2_kernel(FirArgs a) {      // up 2, pad0 2: out [2H][2W]
    __shared__ unsigned smx[64];                                 // xs_out: max |out| per sample of this workgroup (as la_fir4x4_s1p_kernel)
    const int wq = a.Wout >> 2;                                  // 4-column groups per output row
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_plane = (long)a.Hin * wq;
    const int pp = (int)(gid / per_plane);
    const bool live = pp < a.P;
    if (!live && !a.xs_out) return;
    if (a.xs_out) {
        if (threadIdx.x < 64) smx[threadIdx.x] = 0u;
        __syncthreads();
    }
    const int p = live ? pp : a.P - 1;
    float omax = 0.f;
    if (live) {
    const int within = (int)(gid - (long)p * per_plane);
    const int i = within / wq, q = within - i * wq;
    const int j0 = 2 * q;
    const float* ip = a.in + (long)p * a.Hin * a.Win;
    float v[3][4];                                               // in[i-1 .. i+1][j0-1 .. j0+2], zero outside
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = i - 1 + r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ix = j0 - 1 + c;
            v[r][c] = (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) ? ip[(long)iy * a.Win + ix] : 0.f;
        }
    }
    const long HWout = (long)a.Hout * a.Wout;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                            // output column 4q + k = 2 (j0 + k/2) + (k & 1)
            const int px = k & 1, jj = k >> 1;
            float s = 0.f;
#pragma unroll
            for (int ua = 0; ua < 2; ++ua) {                     // taps ta = py + 2 ua at input row i + py + ua - 1
                const int ta = py + 2 * ua, r = py + ua;
#pragma unroll
                for (int ub = 0; ub < 2; ++ub) {
                    const int tb = px + 2 * ub, c = jj + px + ub;
                    s += v[r][c] * a.f[ta * 4 + tb];
                }
            }
            o[k] = s;
        }
        const long pos = (long)(2 * i + py) * a.Wout + 4 * q;
        if (a.addend) {
            const float4 ad = *reinterpret_cast<const float4*>(a.addend + (long)p * HWout + pos);
            o[0] += ad.x; o[1] += ad.y; o[2] += ad.z; o[3] += ad.w;
        }
        *reinterpret_cast<float4*>(a.out + (long)p * HWout + pos) = make_float4(o[0], o[1], o[2], o[3]);
        omax = fmaxf(omax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
    }
    if (a.xs_out) {      // operand scale of `out` for its consumer: per-sample maxima of the workgroup through LDS, one lowering per sample
        const int b = p / a.C, b0 = (int)(((long)blockIdx.x * 256) / per_plane) / a.C;
        const int bi = b - b0;
        if (live && omax > 0.f) {
            if (bi < 64) atomicMax(&smx[bi], __float_as_uint(omax));
            else { float* row = a.xs_out + (long)b * LA_XS_FAN + la_xs_sub(threadIdx.x); la_xs_lower(row, la_xs_peek(row), a.xs_mult ? a.xs_mult[b] : 1.f, omax); }
        }
        __syncthreads();
        const int bt = b0 + (int)threadIdx.x;
        if (threadIdx.x < 64 && smx[threadIdx.x] != 0u && bt * a.C < a.P) {
            float* row = a.xs_out + (long)bt * LA_XS_FAN + la_xs_sub();
            la_xs_lower(row, la_xs_peek(row), a.xs_mult ? a.xs_mult[bt] : 1.f, __uint_as_float(smx[threadIdx.x]));
        }
    }
}

__global__ __launch_bounds__(256) void la_fir4x4_down2_kernel(FirArgs a) {    // down 2, pad0 1: out [H/2][W/2]
    const int wq = a.Wout >> 1;                                  // 2-column groups per output row
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_plane = (long)a.Hout * wq;
    const int p = (int)(gid / per_plane);
    if (p >= a.P) return;
    const int within = (int)(gid - (long)p * per_plane);
    const int y = within / wq, q = within - y * wq;
    const float* ip = a.in + (long)p * a.Hin * a.Win;
    float v[4][6];                                               // in[2y-1 .. 2y+2][4q-1 .. 4q+4]
    const bool vec = (a.Win & 3) == 0 && (((size_t)a.in) & 15) == 0;      // (Win = 2 Wout is even; rows 16-byte aligned when Win % 4 == 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = 2 * y - 1 + r;
        const bool rok = iy >= 0 && iy < a.Hin;
        const float* rp = ip + (long)(rok ? iy : 0) * a.Win + 4 * q;
        if (vec) {      // columns 4q .. 4q+3 as one 16-byte load, the two neighbours as dwords
            const float4 m = rok ? *reinterpret_cast<const float4*>(rp) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[r][1] = m.x; v[r][2] = m.y; v[r][3] = m.z; v[r][4] = m.w;
            v[r][0] = (rok && q > 0) ? rp[-1] : 0.f;
            v[r][5] = (rok && 4 * q + 4 < a.Win) ? rp[4] : 0.f;
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int ix = 4 * q - 1 + c;
                v[r][c] = (rok && ix >= 0 && ix < a.Win) ? rp[c - 1] : 0.f;
            }
        }
    }
    float o[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float s = 0.f;
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) s += v[ta][2 * k + tb] * a.f[ta * 4 + tb];
        o[k] = s;
    }
    const long HWout = (long)a.Hout * a.Wout;
    const long pos = (long)y * a.Wout + 2 * q;
    if (a.addend) { o[0] += a.addend[(long)p * HWout + pos]; o[1] += a.addend[(long)p * HWout + pos + 1]; }
    *reinterpret_cast<float2*>(a.out + (long)p * HWout + pos) = make_float2(o[0], o[1]);
}

// ------------------------------------------------------------------------------------------------------------
// The image-gradient pyramid of a synthesis backward pass in ONE launch (round 5; was one la_fir4x4_down2_kernel launch per block:
// six 5 us links on the serial chain of a step at 256^2).  The gradient of the image at level k-1 is the adjoint of upsample2d applied
// to the gradient at level k (upfirdn2d.py:255-266 via :342-348: flipped 4x4 taps, decimate 2, pad (1,1,1,1), gain 4) and depends on
// nothing else, so one workgroup per (sample, channel) plane walks the levels: the first level is read from global memory, every
// later one from the LDS copy of the level above (ping-pong), each level is also written to its block's g_img.  (The caller keeps
// the top level's own launch -- 256 workgroups against one per plane here -- and hands this kernel the levels from 128^2 down.)
// Arithmetic per output = la_fir4x4_down2_kernel's (taps in ascending row, column order).  Levels whose OUTPUT exceeds 128^2
// (generators above 256^2) keep their own launches.
struct PyrArgs { const float* top; float* out[12]; int nlev, R0; float f[16]; };      // out[l]: [planes][R0 >> (l + 1)]^2
__global__ __launch_bounds__(1024) void la_imgrad_pyramid_kernel(PyrArgs a) {
    extern __shared__ float lds[];
    const int p = blockIdx.x;
    int Rin = a.R0;
    float* bufA = lds;                                  // levels 0, 2, 4, ..  ([R0/2]^2 floats)
    float* bufB = lds + (size_t)(a.R0 >> 1) * (a.R0 >> 1);      // levels 1, 3, ..  ([R0/4]^2 floats)
    for (int l = 0; l < a.nlev; ++l) {
        const int Ro = Rin >> 1;
        const float* gin = l == 0 ? a.top + (size_t)p * Rin * Rin : nullptr;
        const float* lin = l == 0 ? nullptr : ((l & 1) ? bufA : bufB);
        float* lout = (l & 1) ? bufB : bufA;
        float* gout = a.out[l] + (size_t)p * Ro * Ro;
        for (int o = threadIdx.x; o < Ro * Ro; o += 1024) {
            const int y = o / Ro, x = o - y * Ro;
            float s = 0.f;
#pragma unroll
            for (int ta = 0; ta < 4; ++ta) {
                const int iy = 2 * y - 1 + ta;
                const bool rok = iy >= 0 && iy < Rin;
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) {
                    const int ix = 2 * x - 1 + tb;
                    float v = 0.f;
                    if (rok && ix >= 0 && ix < Rin) v = gin ? gin[(size_t)iy * Rin + ix] : lin[iy * Rin + ix];
                    s += v * a.f[ta * 4 + tb];
                }
            }
            lout[o] = s;
            gout[o] = s;
        }
        __syncthreads();
        Rin = Ro;
    }
}

// outs[l] = gradient planes at resolution R0 >> (l + 1), l = 0 .. nlev-1; needs (R0/2)^2 * 4 <= 64 KB (R0 <= 256)
int la_image_grad_pyramid(const float* g_top, float* const* outs, int nlev, int planes, int R0, const float* f_host, hipStream_t stream) {
    LA_CHECK_ARG(g_top && outs && nlev >= 1 && nlev <= 12 && planes >= 1 && R0 >= 2 && R0 <= 256 && (R0 >> nlev) >= 1, "image_grad_pyramid: bad arguments");
    FirArgs fa; int ho, wo;
    int rc = fir_fill(fa, g_top, outs[0], planes, 1, R0, R0, f_host, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1, 1, 4.f, &ho, &wo);      // (the taps as the per-level launches build them)
    if (rc) return rc;
    PyrArgs a;
    a.top = g_top; a.nlev = nlev; a.R0 = R0;
    for (int l = 0; l < 12; ++l) a.out[l] = l < nlev ? outs[l] : nullptr;
    for (int k = 0; k < 16; ++k) a.f[k] = fa.f[k];
    const size_t ldsb = ((size_t)(R0 >> 1) * (R0 >> 1) + (size_t)(R0 >> 2) * (R0 >> 2)) * sizeof(float);
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (ldsb > 65536 && !attr_done[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&la_imgrad_pyramid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) {
            la_set_error("image_grad_pyramid: hipFuncSetAttribute failed"); return LA_ERR_HIP;
        }
        attr_done[dev].store(true, std::memory_order_release);
    }
    const int slot = la_prof_open(LA_PC_FIR, 2.0 * 16 * planes * (double)(R0 / 2) * (R0 / 2) * 4.0 / 3.0, 4.0 * planes * (double)R0 * R0 * 1.67, stream);
    hipLaunchKernelGGL(la_imgrad_pyramid_kernel, dim3(planes), dim3(1024), ldsb, stream, a);
    la_prof_close(slot, stream);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

static int fir_launch_inner(const FirArgs& a, hipStream_t stream);
static int fir_launch(const FirArgs& a, hipStream_t stream) {
    // launch profiler: one read of the input planes + one write of the output planes
    double f = 1.0;      // (row window: the wanted rows only)
    if (a.row_hi > 0 && a.in_xhalf > 0) { const int lo = a.row_lo > 0 ? a.row_lo : 0, hi = a.row_hi < a.Hout ? a.row_hi : a.Hout; f = hi > lo ? (double)(hi - lo) / a.Hout : 0.0; }
    if (a.col_hi > 0 && a.in_xhalf > 0) f *= (double)((((a.col_hi < a.Wout ? a.col_hi : a.Wout) + 3) & ~3) - (a.col_lo & ~3)) / a.Wout;
    const int slot = la_prof_open(LA_PC_FIR, f * 2.0 * a.fw * a.fh * (double)a.P * a.Hout * a.Wout,
                                  f * 4.0 * a.P * ((double)a.Hin * a.Win + (double)a.Hout * a.Wout), stream);
    const int rc = fir_launch_inner(a, stream);
    la_prof_close(slot, stream);
    return rc;
}
static int fir_launch_inner(const FirArgs& a, hipStream_t stream) {
    const bool s1 = a.upx == 1 && a.upy == 1 && a.dnx == 1 && a.dny == 1 && a.fw == 4 && a.fh == 4;
    float fx[4], fy[4];
    if (a.in_xhalf > 0) {      // column-planar rows: only the planar vector kernel reads them
        LA_CHECK_ARG(s1 && !a.pmax && a.padx0 == 1 && a.pady0 == 1 && a.Wout % 4 == 0 && a.Win == a.Wout + 1 &&
                         a.in_pitch % 4 == 0 && a.in_plane % 4 == 0 && a.in_xhalf % 4 == 0 && a.in_xhalf >= (a.Win + 1) / 2 &&
                         a.in_pitch >= a.in_xhalf + a.Win / 2 && (a.noise_bstride % 4) == 0 &&
                         (((size_t)a.in | (size_t)a.out | (size_t)a.noise | (size_t)a.addend | (size_t)a.yref) & 15) == 0 && fir_separable(a.f, fx, fy),
                     "upfirdn2d: column-planar input needs the 4x4 pad-1 separable FIR on aligned planes (W % 4 == 0, no plane maxima)");
        // rows per thread: 16 where that still leaves every SIMD >= 4 waves' worth of threads, else 4 (short serial chains on small planes)
        // (work items of the wanted rows / columns only: the kernel derives the same strip and column-group ranges from the window)
        auto live_items = [&](int rows) {
            int s_lo = 0, s_hi = la_cdiv(a.Hout, rows), g_lo = 0, g_hi = a.Wout / 4;
            if (a.row_hi > 0) s_hi = la_cdiv((a.row_hi < a.Hout ? a.row_hi : a.Hout) - a.row_lo, rows);
            if (a.col_hi > 0) { g_lo = a.col_lo >> 2; g_hi = ((a.col_hi < a.Wout ? a.col_hi : a.Wout) + 3) >> 2; }
            return (long)a.P * (g_hi - g_lo) * (s_hi - s_lo);
        };
        int rows = live_items(16) >= 256l * 4 * 4 * 64 ? 16 : 4;
        if (const char* e = la_dev_env("LA_FIR_ROWS")) rows = atoi(e) == 16 ? 16 : 4;      // (dev knob)
        const long items = live_items(rows);
        LA_CHECK_ARG(items > 0, "upfirdn2d: empty window");
        LA_CHECK_ARG(items < (1l << 38), "upfirdn2d: too many planes");
        dim3 g((unsigned)((items + 255) / 256));
        const float4 vx = make_float4(fx[0], fx[1], fx[2], fx[3]), vy = make_float4(fy[0], fy[1], fy[2], fy[3]);
        if (a.epi == 1) { if (rows == 4) hipLaunchKernelGGL((la_fir4x4_s1p_kernel<1, 4>), g, dim3(256), 0, stream, a, vx, vy); else hipLaunchKernelGGL((la_fir4x4_s1p_kernel<1, 16>), g, dim3(256), 0, stream, a, vx, vy); }
        else if (a.epi == 2) { if (rows == 4) hipLaunchKernelGGL((la_fir4x4_s1p_kernel<2, 4>), g, dim3(256), 0, stream, a, vx, vy); else hipLaunchKernelGGL((la_fir4x4_s1p_kernel<2, 16>), g, dim3(256), 0, stream, a, vx, vy); }
        else { if (rows == 4) hipLaunchKernelGGL((la_fir4x4_s1p_kernel<0, 4>), g, dim3(256), 0, stream, a, vx, vy); else hipLaunchKernelGGL((la_fir4x4_s1p_kernel<0, 16>), g, dim3(256), 0, stream, a, vx, vy); }
        LA_CHECK_LAUNCH();
        return LA_OK;
    }
    LA_CHECK_ARG(a.in_pitch == a.Win && a.in_plane == (long)a.Hin * a.Win, "upfirdn2d: a padded input layout needs the column-planar vector 4x4 kernel");
    if (a.upx == 1 && a.upy == 1 && a.dnx == 1 && a.dny == 1 && a.fw == 4 && a.fh == 4) {
        dim3 g(la_cdiv(a.Wout, 64), la_cdiv(a.Hout, 4 * FIR_ROWS), a.P < 4096 ? a.P : 4096);
        LA_CHECK_ARG(g.y <= 65535, "upfirdn2d: output too tall");
        if (a.epi == 1) hipLaunchKernelGGL(la_fir4x4_s1_kernel<1>, g, dim3(256), 0, stream, a);
        else if (a.epi == 2) hipLaunchKernelGGL(la_fir4x4_s1_kernel<2>, g, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(la_fir4x4_s1_kernel<0>, g, dim3(256), 0, stream, a);
        LA_CHECK_LAUNCH();
        return LA_OK;
    }
    // the image pyramid's factor-2 resamplers (dense input, 4x4 taps, aligned outputs)
    const bool plain = a.fw == 4 && a.fh == 4 && a.epi == 0 && !a.pmax && a.in_pitch == a.Win && a.in_plane == (long)a.Hin * a.Win;
    if (plain && a.upx == 2 && a.upy == 2 && a.dnx == 1 && a.dny == 1 && a.padx0 == 2 && a.pady0 == 2 && a.Wout == 2 * a.Win && a.Hout == 2 * a.Hin &&
        a.Wout % 4 == 0 && (((size_t)a.out | (size_t)a.addend) & 15) == 0) {
        const long items = (long)a.P * a.Hin * (a.Wout / 4);
        hipLaunchKernelGGL(la_fir4x4_up2_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, a);
        LA_CHECK_LAUNCH();
        return LA_OK;
    }
    if (plain && a.upx == 1 && a.upy == 1 && a.dnx == 2 && a.dny == 2 && a.padx0 == 1 && a.pady0 == 1 && 2 * a.Wout == a.Win && 2 * a.Hout == a.Hin &&
        a.Wout % 2 == 0 && (((size_t)a.out) & 7) == 0) {
        const long items = (long)a.P * a.Hout * (a.Wout / 2);
        hipLaunchKernelGGL(la_fir4x4_down2_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, a);
        LA_CHECK_LAUNCH();
        return LA_OK;
    }
    LA_CHECK_ARG(!a.xs_out && a.epi != 2, "upfirdn2d: this shape runs on the generic kernel, which has no operand-scale / activation-backward tail");
    dim3 grid(la_cdiv(a.Wout, 64), la_cdiv(a.Hout, 4), a.P < 1024 ? a.P : 1024);
    LA_CHECK_ARG(grid.y <= 65535, "upfirdn2d: output too tall");
    hipLaunchKernelGGL(la_upfirdn2d_kernel, grid, dim3(256), 0, stream, a);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

int la_upfirdn2d_ex(const float* in, float* out, int B, int C, int Hin, int Win, const float* f_host, int fh, int fw,
                    int upx, int upy, int dnx, int dny, int padx0, int padx1, int pady0, int pady1, int flip_filter,
                    float gain, const float* addend, hipStream_t stream, float* pmax, const LaFirTail* tail) {
    FirArgs a; int ho, wo;
    int rc = fir_fill(a, in, out, B, C, Hin, Win, f_host, fh, fw, upx, upy, dnx, dny, padx0, padx1, pady0, pady1,
                      flip_filter, gain, &ho, &wo);
    if (rc) return rc;
    a.addend = addend;
    if (tail) {
        const bool s1 = upx == 1 && upy == 1 && dnx == 1 && dny == 1 && fw == 4 && fh == 4;
        const bool up2 = upx == 2 && upy == 2 && dnx == 1 && dny == 1 && fw == 4 && fh == 4 && padx0 == 2 && pady0 == 2;
        LA_CHECK_ARG(!tail->yref || (s1 && !addend), "upfirdn2d: the fused activation backward exists for the 4x4 stride-1 kernel");
        LA_CHECK_ARG(!tail->xs_out || s1 || up2, "upfirdn2d: the operand-scale hand-over exists for the 4x4 stride-1 and up-2 kernels");
        if (tail->yref) { a.epi = 2; a.yref = tail->yref; a.act = tail->act; a.alpha = tail->alpha; a.gain = tail->gain; a.clamp = tail->clamp; }
        a.xs_out = tail->xs_out; a.xs_mult = nullptr;
        if (tail->in_pitch > 0) {      // padded / column-planar input rows (the transposed conv's intermediate): the planar vector kernel
            LA_CHECK_ARG(s1, "upfirdn2d: a padded input layout needs the 4x4 stride-1 kernel");
            a.in_pitch = tail->in_pitch; a.in_plane = tail->in_plane; a.in_xhalf = tail->in_xhalf;
        }
    }
    if (pmax && upx == 1 && upy == 1 && dnx == 1 && dny == 1 && fw == 4 && fh == 4) a.pmax = pmax;
    else LA_CHECK_ARG(!pmax, "upfirdn2d: plane maxima are produced by the 4x4 stride-1 kernel only");
    return fir_launch(a, stream);
}

int la_upfirdn2d_modconv_epilogue(const float* in, float* out, int B, int C, int Hin, int Win, const float* f_host,
                                  int fh, int fw, int padx0, int padx1, int pady0, int pady1, float fir_gain,
                                  const float* demod, int demod_stride, const float* noise, long noise_bstride,
                                  float noise_strength, const float* bias, int act, float alpha, float gain,
                                  float clamp, hipStream_t stream, float* pmax, int in_pitch, long in_plane, int in_xhalf, float* xs_out,
                                  const float* xs_mult, int row_lo, int row_hi, int col_lo, int col_hi) {
    FirArgs a; int ho, wo;
    int rc = fir_fill(a, in, out, B, C, Hin, Win, f_host, fh, fw, 1, 1, 1, 1, padx0, padx1, pady0, pady1, 0, fir_gain,
                      &ho, &wo);
    if (rc) return rc;
    a.epi = 1; a.demod = demod; a.demod_stride = demod_stride; a.noise = noise; a.noise_bstride = noise_bstride;
    a.noise_strength = noise_strength; a.bias = bias; a.act = act; a.alpha = alpha; a.gain = gain; a.clamp = clamp;
    if (fw == 4 && fh == 4) a.pmax = pmax;
    else LA_CHECK_ARG(!pmax, "upfirdn2d: plane maxima are produced by the 4x4 stride-1 kernel only");
    if (in_pitch > 0) { a.in_pitch = in_pitch; a.in_plane = in_plane; a.in_xhalf = in_xhalf; }
    LA_CHECK_ARG(!xs_out || (fw == 4 && fh == 4), "upfirdn2d: the operand-scale hand-over exists for the 4x4 stride-1 kernels only");
    a.xs_out = xs_out; a.xs_mult = xs_mult;
    if (in_xhalf > 0) { a.row_lo = row_lo; a.row_hi = row_hi; a.col_lo = col_lo; a.col_hi = col_hi; }      // (the planar vector kernel honours the window; the others compute everything)
    return fir_launch(a, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Adjoint of the FIR that follows a transposed stride-2 conv (pad 2, flipped taps; upfirdn2d.py:255-266), writing its result
// ALREADY in the form the stride-2 backward contraction reads: [b][chunk][pixel][128-byte record], each element scaled by xscale[b]
// and split into two fp16 terms, the record holding the h terms of the chunk's 32 channels followed by their l terms.  One pass replaces FIR adjoint + plane maxima +
// pre-split copy (three passes over a [B][C][(2h+1)^2] tensor).
// The power-of-two operand scale must be known before the first element is written, so it comes from a BOUND on the output,
// |out| <= gain * sum|f| * max|in| (la_xscale_pmax with that factor), not from the output's exact maximum.  A looser scale costs
// nothing measurable: the 2-term split keeps 22 significand bits of every element whatever the scale; only the absolute floor
// (fp16 subnormals, 2^-24 in scaled units) moves, from <= 2^-39 to <= 2^-36 of the sample maximum for a bound 8x too large.
// Workgroup = 8 rows x 64 columns x 32 channels; thread = 4 rows x 4 columns x 4 channels from 16-byte row loads (1.3 loads per
// output; the plain FIR kernel issues 5.5), results transposed through LDS (16-byte slots XOR-swizzled by the pixel quad; two rounds
// of 4 rows, 32 KB) so that every store is a full 128-byte [pixel][32 channels] line.
struct FirPackArgs {
    const float* in;       // [B][C][H][W], W % 4 == 0, 16-byte aligned planes
    unsigned* out;         // [B][nck][Hz*Wz][32]
    const float* xscale;   // [B], or slot rows [B][LA_XS_FAN] (xs_fan = LA_XS_FAN)
    int xs_fan;
    int B, C, H, W, Hz, Wz, nck, pad;
    float f[16];           // effective correlation taps (flip and gain folded in)
    int in_lo, in_hi;      // valid rows of `in` (0 / 0 = all): the others read as zeros (a windowed producer left older contents there)
    int in_c0, in_c1;      // valid columns of `in` likewise (multiples of 4; 0 / 0 = all)
    int out_lo, out_hi;    // row window of the output (0 / 0 = all): 8-row workgroup tiles without a wanted row write nothing
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void la_fir4x4_adj_pack_kernel(FirPackArgs a) {
    __shared__ uint4 tile[256 * 8];                       // [4 rows x 64 cols][8 slots of 4 channels]: half the rows per round (32 KB)
    const int tid = threadIdx.x;
    const int xg = tid & 15, rs = (tid >> 4) & 1, cg = tid >> 5;
    const int b = blockIdx.z / a.nck, ck = blockIdx.z - b * a.nck;
    const int Xb = blockIdx.x * 64, Yb = blockIdx.y * 8;
    if (a.out_hi > 0 && (Yb + 8 <= a.out_lo || Yb >= a.out_hi)) return;      // row window of the output: nothing wanted in these 8 rows
    const int vin0 = a.in_hi > 0 ? a.in_lo : 0, vin1 = a.in_hi > 0 ? a.in_hi : a.H;
    const int X0 = Xb + xg * 4, Y0 = Yb + rs * 4;
    const float xs = la_xs_get(a.xscale, b, a.xs_fan);
    unsigned pk[4][4][4];                                 // [row][col][channel of the group]: {h | l << 16}
    const int vc0 = a.in_c1 > 0 ? a.in_c0 : 0, vc1 = a.in_c1 > 0 ? a.in_c1 : a.W;
    const bool lo_ok = X0 - 4 >= vc0 && X0 - 4 < vc1, mid_ok = X0 >= vc0 && X0 < vc1, hi_ok = X0 + 4 >= vc0 && X0 + 4 < vc1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = ck * 32 + cg * 4 + j;
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
        if (c < a.C) {
            const float* ip = a.in + ((long)b * a.C + c) * a.H * a.W;
#pragma unroll
            for (int wr = 0; wr < 7; ++wr) {
                const int iy = Y0 - a.pad + wr;
                float v[8];                               // v[t] = in[iy][X0 - 2 + t], t = 0..6
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = 0.f;
                if (iy >= vin0 && iy < vin1) {
                    const float* rp = ip + (long)iy * a.W;
                    if (lo_ok) { const float4 q = *reinterpret_cast<const float4*>(rp + X0 - 4); v[0] = q.z; v[1] = q.w; }
                    if (mid_ok) { const float4 q = *reinterpret_cast<const float4*>(rp + X0); v[2] = q.x; v[3] = q.y; v[4] = q.z; v[5] = q.w; }
                    if (hi_ok) v[6] = rp[X0 + 4];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ta = wr - r;
                    if (ta >= 0 && ta < 4) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int tb = 0; tb < 4; ++tb) acc[r][k] += v[k + tb] * a.f[ta * 4 + tb];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float s = acc[r][k] * xs;
                const _Float16 h = (_Float16)s;
                const _Float16 l = (_Float16)(s - (float)h);
                pk[r][k][j] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
            }
    }
    // two rounds through LDS (rows 0-1 then rows 2-3 of every thread): 32 KB per workgroup instead of 64, twice the waves per CU
    const long plane = (long)a.Hz * a.Wz;
    uint4* op = reinterpret_cast<uint4*>(a.out) + ((long)b * a.nck + ck) * plane * 8;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = half * 2 + r2;
                const int pl = (rs * 2 + r2) * 64 + xg * 4 + k;
                tile[pl * 8 + (cg ^ ((pl >> 2) & 7))] = make_uint4(pk[r][k][0], pk[r][k][1], pk[r][k][2], pk[r][k][3]);
            }
        __syncthreads();
        // read-out: the record of a pixel is [h of its 32 channels | l of its 32 channels] (la_conv_bf16_kernel's pieces), so a thread
        // takes two neighbouring channel groups (8 channels, 32 bytes of {h | l} pairs) and stores their h and l halves as 16-byte slots
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid;
            const int pl = idx >> 2, sp = idx & 3;
            const int lr = pl >> 6;
            const int Y = Yb + (lr >> 1) * 4 + half * 2 + (lr & 1), X = Xb + (pl & 63);
            const int sw = (pl >> 2) & 7;
            const uint4 e0 = tile[pl * 8 + ((2 * sp) ^ sw)], e1 = tile[pl * 8 + ((2 * sp + 1) ^ sw)];
            if (Y < a.Hz && X < a.Wz) {
                uint4* rec = op + ((long)Y * a.Wz + X) * 8;
                rec[sp] = make_uint4(__builtin_amdgcn_perm(e0.y, e0.x, 0x05040100u), __builtin_amdgcn_perm(e0.w, e0.z, 0x05040100u),
                                     __builtin_amdgcn_perm(e1.y, e1.x, 0x05040100u), __builtin_amdgcn_perm(e1.w, e1.z, 0x05040100u));
                rec[4 + sp] = make_uint4(__builtin_amdgcn_perm(e0.y, e0.x, 0x07060302u), __builtin_amdgcn_perm(e0.w, e0.z, 0x07060302u),
                                         __builtin_amdgcn_perm(e1.y, e1.x, 0x07060302u), __builtin_amdgcn_perm(e1.w, e1.z, 0x07060302u));
            }
        }
    }
}

// in [B][C][H][W] -> q [B][ceil(C/32)][(H+1)*(W+1)][32] packed fp16 pairs of  xscale[b] * (FIR adjoint of `in`)
int la_fir4x4_adjoint_pack_f16(const float* in, unsigned* q, const float* xscale, int xs_fan, int B, int C, int H, int W, const float* f_host,
                               float gain, hipStream_t stream, int flip_taps, int in_lo, int in_hi, int out_lo, int out_hi, int in_c0, int in_c1) {
    LA_CHECK_ARG(in && q && xscale && f_host, "fir_adjoint_pack: null pointer");
    LA_CHECK_ARG(W % 4 == 0 && (((size_t)in | (size_t)q) & 15) == 0, "fir_adjoint_pack: rows must be 16-byte aligned");
    FirPackArgs a;
    a.in = in; a.out = q; a.xscale = xscale; a.xs_fan = xs_fan; a.B = B; a.C = C; a.H = H; a.W = W; a.Hz = H + 1; a.Wz = W + 1; a.nck = la_cdiv(C, 32);
    a.in_lo = in_lo; a.in_hi = in_hi; a.out_lo = out_lo; a.out_hi = out_hi; a.in_c0 = in_c0; a.in_c1 = in_c1;
    LA_CHECK_ARG(in_c0 % 4 == 0 && in_c1 % 4 == 0, "fir_adjoint_pack: the column mask is in groups of 4");
    a.pad = 2;         // adjoint of pad (1,1,1,1): fw - 1 - pad = 2 per side (upfirdn2d.py:255-266)
    // adjoint = correlation with the flipped filter = flip_filter of the forward op negated; the forward (flip_filter = False)
    // correlates with the flipped taps, so the adjoint correlates with the taps as given
    // (flip_taps: the FORWARD op with pad (2,2,2,2) -- upfirdn2d(x, f, padding=2), a true convolution, i.e. correlation with the flipped
    //  taps, upfirdn2d.py:198-199 -- has the same geometry: the discriminator's stride-2 conv reads its (H+1)^2 pre-filtered input this way)
    for (int i = 0; i < 16; ++i) a.f[i] = gain * f_host[flip_taps ? 15 - i : i];
    dim3 grid(la_cdiv(a.Wz, 64), la_cdiv(a.Hz, 8), B * a.nck);
    LA_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "fir_adjoint_pack: grid too large");
    const double wf = out_hi > 0 ? (double)((out_hi < a.Hz ? out_hi : a.Hz) - out_lo) / a.Hz : 1.0;
    const int slot = la_prof_open(LA_PC_FIR, wf * 32.0 * B * C * (double)a.Hz * a.Wz, wf * 4.0 * B * C * ((double)H * W + (double)a.Hz * a.Wz), stream);
    hipLaunchKernelGGL(la_fir4x4_adj_pack_kernel, grid, dim3(256), 0, stream, a);
    la_prof_close(slot, stream);
    LA_CHECK_LAUNCH();
    return LA_OK;
}

extern "C" int la_upfirdn2d_out_size(int in_size, int up, int down, int pad0, int pad1, int taps) {
    return (in_size * up + pad0 + pad1 - taps + down) / down;
}

extern "C" int la_upfirdn2d_f32(const float* x, const float* f_host, float* y, int N, int C, int H, int W, int fh,
                                int fw, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0,
                                int pady1, int flip, float gain, hipStream_t stream) {
    return la_upfirdn2d_ex(x, y, N, C, H, W, f_host, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip,
                           gain, nullptr, stream);
}
