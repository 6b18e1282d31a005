// Modulated 3x3 convolution of a SynthesisLayer, forward and backward-to-(input, style), composed from the
// implicit-GEMM contraction (la_conv.hip) and the FIR kernels (la_upfirdn2d.hip).
// Reference semantics: SG2 `modulated_conv2d` + `bias_act` inside SynthesisLayer.forward, with the
// resampling algebra of models/stylegan3/torch_utils/ops/conv2d_resample.py:82-86,112-134 (Appendix B).
#include "la_modconv.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "la_conv.h"
#include "la_upfirdn2d.h"

static void base_args(LaConvArgs& a) {
    memset(&a, 0, sizeof(a));
    a.in_sy = a.in_sx = a.out_sy = a.out_sx = 1;
    a.clamp = -1.f; a.gain = 1.f; a.act = LA_ACT_LINEAR;
}

extern "C" int la_pack_conv_weights_f32(const float* w, float* wf, float* wb, float* wsq, int cout, int cin, int ktaps,
                                        hipStream_t stream) {
    return la_pack_conv_weights(w, wf, wb, wsq, cout, cin, ktaps, stream);
}

static void fwd_shape(LaConvArgs& a, int precision, int B, int cin, int cout, int res) {
    a.precision = precision;
    a.B = B; a.C = cin; a.M = cout; a.Hin = a.Win = a.Hout = a.Wout = a.Gy = a.Gx = res;
    a.ntaps = 9;
    for (int t = 0; t < 9; ++t) { a.tap_dy[t] = t / 3 - 1; a.tap_dx[t] = t % 3 - 1; a.tap_w[t] = t; }
}

// one row tile of the halo kernel = all output channels (tiles of 128, 64 or -- for <= 32 channels -- 32 rows, la_conv_launch)
bool la_modconv3x3_fwd_fuses_rgb(int precision, int B, int cin, int cout, int res) {
    static const bool off = la_dev_env("LA_NO_RGB_FUSE") != nullptr;      // dev knob
    if (off || precision == LA_PREC_F32 || (cout != 128 && cout != 64 && cout != 32)) return false;
    LaConvArgs a; base_args(a);
    fwd_shape(a, precision, B, cin, cout, res);
    return la_conv_bf16_uses_halo(a);
}

int la_modconv3x3_fwd_ex(const float* x, long x_bstride, const float* in_pmax, int in_nseg, const float* wf, const void* wq, int precision, const float* s,
                         int s_stride, const float* d, int d_stride, const float* noise, long noise_bstride, float noise_strength,
                         const float* bias, int act, float alpha, float gain, float clamp, float* y, void* ws, size_t ws_bytes, int B,
                         int cin, int cout, int res, hipStream_t stream, const float* xscale, const LaRgbFuse* rgb, float* xs_out,
                         const float* xs_mult, int row_lo, int row_hi, int col_lo, int col_hi) {
    LA_CHECK_ARG(x && wf && y, "modconv_fwd: null pointer");
    LA_CHECK_ARG(col_lo >= 0 && (col_hi == 0 || (row_hi > 0 && col_hi > col_lo && col_hi <= res)), "modconv_fwd: bad column window");
    LA_CHECK_ARG(row_lo >= 0 && (row_hi == 0 || (row_hi > row_lo && row_hi <= res)), "modconv_fwd: bad row window");
    LaConvArgs a; base_args(a);
    a.row_lo = row_lo; a.row_hi = row_hi;
    a.col_lo = col_lo; a.col_hi = col_hi;      // (round 4 took the arguments and never passed them on: the forward conv1 of the top block computed every tile column)
    a.fwd_xs_out = xs_out; a.fwd_xs_mult = xs_mult;
    if (rgb) {
        LA_CHECK_ARG(rgb->imgc >= 1 && rgb->imgc <= 4 && rgb->w && rgb->s && rgb->rgb_pre && rgb->img && la_modconv3x3_fwd_fuses_rgb(precision, B, cin, cout, res),
                     "modconv_fwd: this launch cannot carry the fused ToRGB (la_modconv3x3_fwd_fuses_rgb)");
        a.rgb_imgc = rgb->imgc; a.rgb_w = rgb->w; a.rgb_s = rgb->s; a.rgb_s_stride = rgb->s_stride; a.rgb_bias = rgb->bias;
        a.rgb_skip = rgb->skip; a.rgb_pre = rgb->rgb_pre; a.rgb_img = rgb->img; a.rgb_clamp = rgb->clamp;
    }
    if (precision == LA_PREC_F16X2 && xscale) { a.acc_scale_x = xscale; a.acc_scale_fan = LA_XS_FAN; }      // preset operand scale (slot rows of the caller): no absmax / plane-maxima pass
    a.in = x; a.in_bstride = x_bstride; a.wgt = wf; a.out = y; a.in_pmax = in_pmax; a.in_pmax_nseg = in_nseg;
    a.in_scale = s; a.scale_stride = s_stride;
    a.ws = ws; a.ws_bytes = ws_bytes;
    a.precision = precision; a.wgt_bf16 = wq; a.wgt_bf16_term_elems = la_conv_bf16_pack_elems(cout, cin, 9);
    a.B = B; a.C = cin; a.M = cout; a.Hin = a.Win = a.Hout = a.Wout = a.Gy = a.Gx = res;
    a.ntaps = 9;
    for (int t = 0; t < 9; ++t) { a.tap_dy[t] = t / 3 - 1; a.tap_dx[t] = t % 3 - 1; a.tap_w[t] = t; }
    a.epi = LA_EPI_FWD;
    a.demod = d; a.demod_stride = d_stride;
    a.noise = noise; a.noise_bstride = noise_bstride; a.noise_strength = noise_strength;
    a.bias = bias; a.act = act; a.alpha = alpha; a.gain = gain; a.clamp = clamp;
    return la_conv_launch(a, stream);
}

extern "C" int la_modconv3x3_fwd_f32(const float* x, long x_bstride, const float* wf, const void* wq, int precision, const float* s, int s_stride, const float* d,
                                     int d_stride, const float* noise, long noise_bstride, float noise_strength, const float* bias,
                                     int act, float alpha, float gain, float clamp, float* y, void* ws, size_t ws_bytes, int B, int cin, int cout, int res,
                                     hipStream_t stream) {
    return la_modconv3x3_fwd_ex(x, x_bstride, nullptr, 0, wf, wq, precision, s, s_stride, d, d_stride, noise, noise_bstride, noise_strength, bias, act,
                                alpha, gain, clamp, y, ws, ws_bytes, B, cin, cout, res, stream);
}

int la_modconv3x3_up2_fwd_ex(const float* x, long x_bstride, const float* wf, const void* wq, int precision, const float* s, int s_stride,
                             const float* d, int d_stride, const float* noise, long noise_bstride, float noise_strength,
                             const float* bias, int act, float alpha, float gain, float clamp, const float* fir_host,
                             float* scratch, float* y, float* y_pmax, void* ws, size_t ws_bytes, int B, int cin, int cout, int res,
                             hipStream_t stream, const float* xscale, int scratch_pitch, int scratch_xhalf, float* xs_out, const float* xs_mult,
                             int row_lo, int row_hi, int col_lo, int col_hi) {
    LA_CHECK_ARG(x && wf && y && scratch && fir_host, "modconv_up2_fwd: null pointer");
    LA_CHECK_ARG(row_lo >= 0 && (row_hi == 0 || (row_hi > row_lo && row_hi <= res)), "modconv_up2_fwd: bad row window");
    if (scratch_xhalf == 0) row_lo = row_hi = 0;      // (only the planar FIR kernel honours a window)
    if (row_hi == 0) col_lo = col_hi = 0;
    LA_CHECK_ARG(scratch_pitch == 0 || scratch_pitch >= res + 1, "modconv_up2_fwd: scratch pitch smaller than a row");
    LA_CHECK_ARG((scratch_xhalf == 0 && scratch_pitch == 0) || (scratch_xhalf >= res / 2 + 1 && scratch_pitch >= scratch_xhalf + res / 2),
                 "modconv_up2_fwd: bad column-planar scratch layout");
    LA_CHECK_ARG(res >= 2 && res % 2 == 0, "modconv_up2_fwd: output resolution must be even");
    // transposed stride-2 conv as 4 output phases: row Y = 2*qy + py receives taps ky with (Y - ky) even
    const int hin = res / 2;
    LaConvArgs a; base_args(a);
    a.in = x; a.in_bstride = x_bstride; a.wgt = wf; a.out = scratch;
    a.in_scale = s; a.scale_stride = s_stride;
    a.ws = ws; a.ws_bytes = ws_bytes;
    a.precision = precision; a.wgt_bf16 = wq; a.wgt_bf16_term_elems = la_conv_bf16_pack_elems(cout, cin, 9);
    a.B = B; a.C = cin; a.M = cout; a.Hin = a.Win = hin; a.Hout = a.Wout = res + 1;
    a.out_sy = a.out_sx = 2; a.epi = LA_EPI_RAW;
    if (scratch_pitch > 0) { a.out_pitch = scratch_pitch; a.out_plane = (long)scratch_pitch * (res + 1); }      // padded (2h+1)-wide rows
    if (scratch_xhalf > 0) { a.out_sx = 1; a.Wout = scratch_pitch; }      // column-planar rows: phase px writes the contiguous run from px * xhalf
    if (precision == LA_PREC_F16X2 && xscale) { a.acc_scale_x = xscale; a.acc_scale_fan = LA_XS_FAN; }      // preset operand scale (slot rows of the caller): no absmax pass
    if (row_hi > 0) {
        // FIR output row y reads intermediate rows y - 1 .. y + 2; intermediate row Y = 2 q + py belongs to row q of phase py
        const int zlo = row_lo - 1 > 0 ? row_lo - 1 : 0, zhi = row_hi + 2 < res + 1 ? row_hi + 2 : res + 1;
        a.row_lo = zlo >> 1; a.row_hi = ((zhi - 1) >> 1) + 1;
        // the input rows those phase rows read: nothing else is copied into the pre-split operand, and the contraction reads the rest as zeros
        la_modconv3x3_up2_fwd_rows(res, row_lo, row_hi, &a.in_row_lo, &a.in_row_hi);
    }
    if (precision != LA_PREC_F32) {
        // split the (modulated) input once for the four phase launches
        int rc = la_conv_prepare_input(a, stream);
        if (rc) return rc;
    }
    // 16-bit kernels: the four phases in ONE launch -- above the split-K sizes each phase launch would end in a nearly empty
    // round, at the split-K sizes (<= 34x34 phase grids) four launches + four finish passes become one of each
    const bool merged = precision != LA_PREC_F32;
    int np = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            a.out_oy = py; a.out_ox = scratch_xhalf > 0 ? px * scratch_xhalf : px;
            a.Gy = py ? hin : hin + 1; a.Gx = px ? hin : hin + 1;
            int nt = 0;
            for (int ky = py; ky < 3; ky += 2)
                for (int kx = px; kx < 3; kx += 2) {
                    a.tap_dy[nt] = -(ky / 2); a.tap_dx[nt] = -(kx / 2); a.tap_w[nt] = ky * 3 + kx; ++nt;
                }
            a.ntaps = nt;
            if (merged) {
                LaConvArgs::Phase& P = a.ph[np++];
                P.Gy = a.Gy; P.Gx = a.Gx; P.out_oy = py; P.out_ox = a.out_ox; P.ntaps = nt;
                for (int t = 0; t < nt; ++t) { P.tap_dy[t] = a.tap_dy[t]; P.tap_dx[t] = a.tap_dx[t]; P.tap_w[t] = a.tap_w[t]; }
                continue;
            }
            int rc = la_conv_launch(a, stream);
            if (rc) return rc;
        }
    if (merged) {
        a.nphase = np;
        a.out_oy = a.out_ox = 0; a.Gy = a.Gx = hin + 1; a.ntaps = 4;      // launch-wide fields = the largest phase (checks only)
        int rc = la_conv_launch(a, stream);
        if (rc) return rc;
    }
    // FIR with pad (1,1,1,1) and gain up^2 = 4 (conv2d_resample.py:119-126), then the layer epilogue
    return la_upfirdn2d_modconv_epilogue(scratch, y, B, cout, res + 1, res + 1, fir_host, 4, 4, 1, 1, 1, 1, 4.f, d, d_stride,
                                         noise, noise_bstride, noise_strength, bias, act, alpha, gain, clamp, stream, y_pmax,
                                         scratch_pitch, (long)scratch_pitch * (res + 1), scratch_xhalf, xs_out, xs_mult, row_lo, row_hi, col_lo, col_hi);
}

// input rows [in_lo, in_hi) (of the res/2-row input) that la_modconv3x3_up2_fwd_ex reads for the row window [row_lo, row_hi) of y: the phase
// rows of the window (above) plus the tap row above
void la_modconv3x3_up2_fwd_rows(int res, int row_lo, int row_hi, int* in_lo, int* in_hi) {
    const int hin = res / 2;
    const int zlo = row_lo - 1 > 0 ? row_lo - 1 : 0, zhi = row_hi + 2 < res + 1 ? row_hi + 2 : res + 1;
    const int qlo = zlo >> 1, qhi = ((zhi - 1) >> 1) + 1;
    // (the flattened 128-pixel tiles at the ends of the window reach into phase rows outside it; what they compute there from rows the
    //  producer did not deliver lands in intermediate rows that the FIR never reads)
    int lo = qlo - 1, hi = qhi;
    *in_lo = lo > 0 ? lo : 0; *in_hi = hi < hin ? hi : hin;
}

extern "C" int la_modconv3x3_up2_fwd_f32(const float* x, long x_bstride, const float* wf, const void* wq, int precision, const float* s, int s_stride,
                                         const float* d, int d_stride, const float* noise, long noise_bstride,
                                         float noise_strength, const float* bias, int act, float alpha, float gain,
                                         float clamp, const float* fir_host, float* scratch, float* y, void* ws, size_t ws_bytes, int B, int cin,
                                         int cout, int res, hipStream_t stream) {
    return la_modconv3x3_up2_fwd_ex(x, x_bstride, wf, wq, precision, s, s_stride, d, d_stride, noise, noise_bstride, noise_strength, bias, act,
                                    alpha, gain, clamp, fir_host, scratch, y, nullptr, ws, ws_bytes, B, cin, cout, res, stream, nullptr, 0);
}

static void set_seam(LaConvArgs& a, const LaSeamFuse* seam) {
    if (!seam) return;
    a.seam_demod = seam->demod; a.seam_demod_stride = seam->demod_stride; a.seam_bias = seam->bias;
    a.seam_noise = seam->noise; a.seam_noise_bstride = seam->noise_bstride; a.seam_noise_strength = seam->noise_strength;
    a.seam_act = seam->act; a.seam_alpha = seam->alpha; a.seam_gain = seam->gain; a.seam_clamp = seam->clamp;
    a.seam_ddn_part = seam->ddn_part; a.seam_pmax = seam->pmax; a.seam_xs_out = seam->xs_out; a.seam_xs_mult = seam->xs_mult;
    a.seam_imgc = seam->imgc; a.seam_gimg = seam->g_img; a.seam_rgbpre = seam->rgb_pre; a.seam_rgb_clamp = seam->rgb_clamp;
    a.seam_wrgb = seam->wrgb; a.seam_srgb = seam->s_rgb; a.seam_srgb_stride = seam->s_rgb_stride; a.seam_dweff_part = seam->dweff_part;
}

int la_modconv3x3_bwd_ex(const float* gz, const float* in_pmax, int in_nseg, const float* wb, const void* wq, int precision, const float* s, int s_stride,
                         const float* xin, long xin_bstride, float* gx, float* ds_part, void* ws, size_t ws_bytes, int B, int cin, int cout,
                         int res, hipStream_t stream, const LaSeamFuse* seam, const float* xscale, const LaBwdRows* rows) {
    LA_CHECK_ARG(gz && wb && gx, "modconv_bwd: null pointer");
    LA_CHECK_ARG(!seam || (precision != LA_PREC_F32 && xin && seam->ddn_part), "modconv_bwd: the fused seam needs a 16-bit contraction and xin");
    LaConvArgs a; base_args(a);
    a.in = gz; a.in_bstride = (long)cout * res * res; a.wgt = wb; a.out = gx; a.in_pmax = in_pmax; a.in_pmax_nseg = in_nseg;
    a.ws = ws; a.ws_bytes = ws_bytes;
    a.precision = precision; a.wgt_bf16 = wq; a.wgt_bf16_term_elems = la_conv_bf16_pack_elems(cin, cout, 9);
    a.B = B; a.C = cout; a.M = cin; a.Hin = a.Win = a.Hout = a.Wout = a.Gy = a.Gx = res;
    a.ntaps = 9;
    for (int t = 0; t < 9; ++t) { a.tap_dy[t] = 1 - t / 3; a.tap_dx[t] = 1 - t % 3; a.tap_w[t] = t; }
    if (precision == LA_PREC_F16X2 && xscale) { a.acc_scale_x = xscale; a.acc_scale_fan = LA_XS_FAN; a.in_pmax = nullptr; }      // preset operand scale (slot rows)
    a.epi = LA_EPI_BWD;
    a.out_scale = s; a.oscale_stride = s_stride;
    a.xin = xin; a.xin_bstride = xin_bstride;
    a.ds_part = ds_part; a.tiles_per_sample = la_conv_tiles_per_sample(res, res);
    set_seam(a, seam);
    if (rows && precision != LA_PREC_F32) {
        LA_CHECK_ARG(rows->in_lo >= 0 && rows->in_hi <= res && rows->out_lo >= 0 && rows->out_hi <= res, "modconv_bwd: bad row windows");
        a.in_row_lo = rows->in_lo; a.in_row_hi = rows->in_hi; a.row_lo = rows->out_lo; a.row_hi = rows->out_hi;
        if (rows->out_hi > 0) { a.col_lo = rows->out_c0; a.col_hi = rows->out_c1; }      // (gz is valid in every column here: no input column mask)
    }
    return la_conv_launch(a, stream);
}

extern "C" int la_modconv3x3_bwd_f32(const float* gz, const float* wb, const void* wq, int precision, const float* s, int s_stride, const float* xin,
                                     long xin_bstride, float* gx, float* ds_part, void* ws, size_t ws_bytes, int B, int cin, int cout, int res,
                                     hipStream_t stream) {
    return la_modconv3x3_bwd_ex(gz, nullptr, 0, wb, wq, precision, s, s_stride, xin, xin_bstride, gx, ds_part, ws, ws_bytes, B, cin, cout, res, stream);
}

int la_modconv3x3_up2_bwd_ex(const float* gz, const float* gz_pmax, int gz_nseg, const float* wb, const void* wq, int precision, const float* s,
                             int s_stride, const float* xin, long xin_bstride, const float* fir_host, float* scratch, float* gx,
                             float* ds_part, void* ws, size_t ws_bytes, int B, int cin, int cout, int res, hipStream_t stream,
                             const LaSeamFuse* seam, const float* xscale_in, const LaBwdRows* rows) {
    LA_CHECK_ARG(gz && wb && gx && scratch && fir_host, "modconv_up2_bwd: null pointer");
    LA_CHECK_ARG(!seam || (precision != LA_PREC_F32 && xin && seam->ddn_part && (seam->imgc == 0 || (seam->g_img && seam->wrgb && seam->s_rgb && seam->dweff_part))),
                 "modconv_up2_bwd: the fused seam needs a 16-bit contraction, xin and its output buffers");
    const int hin = res / 2;
    LaConvArgs a; base_args(a);
    set_seam(a, seam);
    a.wgt = wb; a.out = gx;
    a.precision = precision; a.wgt_bf16 = wq; a.wgt_bf16_term_elems = la_conv_bf16_pack_elems(cin, cout, 9);
    a.B = B; a.C = cout; a.M = cin; a.Hin = a.Win = res + 1; a.Hout = a.Wout = a.Gy = a.Gx = hin;
    a.in_sy = a.in_sx = 2; a.ntaps = 9;
    for (int t = 0; t < 9; ++t) { a.tap_dy[t] = t / 3; a.tap_dx[t] = t % 3; a.tap_w[t] = t; }
    a.epi = LA_EPI_BWD;
    a.out_scale = s; a.oscale_stride = s_stride;
    a.xin = xin; a.xin_bstride = xin_bstride;
    a.ds_part = ds_part; a.tiles_per_sample = la_conv_tiles_per_sample(hin, hin);
    a.in_bstride = (long)cout * (res + 1) * (res + 1);
    const size_t qbytes = (size_t)B * la_cdiv(cout, 32) * 32 * (res + 1) * (res + 1) * 4;
    const size_t fused_need = 512 + ((qbytes + 255) & ~(size_t)255);
    if (precision == LA_PREC_F16X2 && (xscale_in || (gz_pmax && gz_nseg >= 1)) && res % 4 == 0 && ws && ws_bytes > fused_need && (((size_t)ws | (size_t)gz) & 15) == 0) {
        // fp16 mode with the plane maxima of gz at hand (left by the seam kernel): ONE pass turns gz into the contraction's
        // operand -- FIR adjoint (pad 2, flipped taps, gain 4; upfirdn2d.py:255-266) + operand scale + fp16 split + channel
        // interleave.  The scale comes from the bound |adjoint(gz)| <= 4 * sum(f) * max|gz| = 4 * max|gz| (see la_upfirdn2d.hip).
        const float* xscale = xscale_in;      // (slot rows, already final: the producer of gz lowered them with the same bound, la_modconv_up2_bwd_xs_mult)
        int xs_fan = LA_XS_FAN;
        unsigned* q = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + 512);
        int rc;
        if (!xscale) {
            float* xs = static_cast<float*>(ws);
            if ((rc = la_conv_xscale_from_pmax(gz_pmax, gz_nseg, nullptr, 0, la_modconv_up2_bwd_xs_mult(fir_host), xs, B, cout, stream))) return rc;
            xscale = xs; xs_fan = 1;
        }
        int zlo = 0, zhi = 0;
        if (rows && rows->in_hi > 0) {      // rows of the adjoint that can be non-zero: 4 taps, pad 2
            LA_CHECK_ARG(rows->in_lo >= 0 && rows->in_hi <= res && rows->out_lo >= 0 && rows->out_hi <= hin, "modconv_up2_bwd: bad row windows");
            zlo = rows->in_lo - 2 > 0 ? rows->in_lo - 2 : 0; zhi = rows->in_hi + 2 < res + 1 ? rows->in_hi + 2 : res + 1;
            a.in_row_lo = zlo; a.in_row_hi = zhi;
        }
        if (rows) { a.row_lo = rows->out_lo; a.row_hi = rows->out_hi; }
        if ((rc = la_fir4x4_adjoint_pack_f16(gz, q, xscale, xs_fan, B, cout, res, res, fir_host, 4.f, stream, 0, rows ? rows->in_lo : 0, rows ? rows->in_hi : 0, zlo, zhi,
                                             rows ? rows->in_c0 : 0, rows ? rows->in_c1 : 0))) return rc;
        a.in = gz;                       // (not read: the launch takes its operand from in_q)
        a.in_q = q; a.acc_scale_x = xscale; a.acc_scale_fan = xs_fan;
        a.ws = static_cast<char*>(ws) + fused_need; a.ws_bytes = ws_bytes - fused_need;
        return la_conv_launch(a, stream);
    }
    // adjoint of [pad (1,1,1,1) -> FIR]: pad fw-1-pad = 2 per side, flipped filter, same gain (upfirdn2d.py:255-266)
    // (fp16 mode: the FIR kernel also leaves the plane maxima of the scratch at the head of ws, so the contraction below
    //  needs no absmax pass)
    float* pmax = nullptr;
    const int nseg = la_fir4x4_segments(res + 1, res + 1);
    const size_t pm_bytes = ((size_t)B * cout * nseg * sizeof(float) + 255) & ~(size_t)255;
    if (precision == LA_PREC_F16X2 && ws && ws_bytes > pm_bytes) {
        pmax = static_cast<float*>(ws);
        ws = static_cast<char*>(ws) + pm_bytes;
        ws_bytes -= pm_bytes;
    }
    int rc = la_upfirdn2d_ex(gz, scratch, B, cout, res, res, fir_host, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, 1, 4.f, nullptr, stream, pmax);
    if (rc) return rc;
    a.in = scratch; a.in_pmax = pmax; a.in_pmax_nseg = nseg;
    a.ws = ws; a.ws_bytes = ws_bytes;
    return la_conv_launch(a, stream);
}

// bound factor of the fused FIR-adjoint operand: |adjoint(gz)| <= 4 * sum|f| * max|gz|
float la_modconv_up2_bwd_xs_mult(const float* fir_host) {
    float fsum = 0.f;
    for (int i = 0; i < 16; ++i) fsum += fabsf(fir_host[i]);
    return 4.f * fsum;
}

extern "C" int la_modconv3x3_up2_bwd_f32(const float* gz, const float* wb, const void* wq, int precision, const float* s, int s_stride, const float* xin,
                                         long xin_bstride, const float* fir_host, float* scratch, float* gx,
                                         float* ds_part, void* ws, size_t ws_bytes, int B, int cin, int cout,
                                         int res, hipStream_t stream) {
    return la_modconv3x3_up2_bwd_ex(gz, nullptr, 0, wb, wq, precision, s, s_stride, xin, xin_bstride, fir_host, scratch, gx, ds_part, ws, ws_bytes,
                                    B, cin, cout, res, stream);
}

extern "C" int la_modconv_ds_tiles(int grid_res) { return la_conv_tiles_per_sample(grid_res, grid_res); }

// scratch bytes for a layer's forward and backward launches at any contraction precision:
// pre-split bf16 copy of the launch input (split-bf16 only) + split-K slice partials (<= 34x34 grids).
extern "C" size_t la_modconv_workspace_bytes(int B, int cin, int cout, int res, int up) {
    size_t need = 0;
    const int hin = up ? res / 2 : res;
    for (int prec = 0; prec <= 3; ++prec) {
        long f, b;
        size_t qf = 0, qb = 0;
        if (up) {
            if (prec == LA_PREC_F32) f = la_conv_splitk_floats(B, cout, cin, hin + 1, hin + 1, prec);   // one phase per launch
            else {      // four phases' partials side by side
                const int gy[4] = {hin + 1, hin + 1, hin, hin}, gx[4] = {hin + 1, hin, hin + 1, hin};
                f = la_conv_splitk_floats_phases(B, cout, cin, 4, gy, gx, prec);
            }
            b = la_conv_splitk_floats(B, cin, cout, hin, hin, prec);
            if (prec) { qf = la_conv_presplit_bytes(B, cin, hin, hin); qb = la_conv_presplit_bytes(B, cout, res + 1, res + 1); }
        } else {
            f = la_conv_splitk_floats(B, cout, cin, res, res, prec);
            b = la_conv_splitk_floats(B, cin, cout, res, res, prec);
            if (prec) { qf = la_conv_presplit_bytes(B, cin, res, res); qb = la_conv_presplit_bytes(B, cout, res, res); }
        }
        const size_t nf = ((qf + 255) & ~(size_t)255) + (size_t)f * 4, nb = ((qb + 255) & ~(size_t)255) + (size_t)b * 4;
        if (nf > need) need = nf;
        if (nb > need) need = nb;
    }
    // head of the workspace in la_modconv3x3_up2_bwd (fp16 mode): plane maxima of the FIR-adjoint scratch
    need += ((size_t)B * (cin > cout ? cin : cout) * la_fir4x4_segments(res + 1, res + 1) * sizeof(float) + 255) & ~(size_t)255;
    return need;
}

// split-bf16 weight packs: bytes for one direction (forward: transpose = 0, backward: transpose = 1), nterm terms
extern "C" size_t la_modconv_bf16_pack_bytes(int cin, int cout, int transpose, int nterm) {
    (void)nterm;      // one pack serves every split precision (3 bf16 terms + 2 fp16 terms + the fp16 weight scale)
    return la_conv_split_pack_bytes(transpose ? cin : cout, transpose ? cout : cin, 9);
}

extern "C" int la_pack_conv_weights_bf16_f32(const float* w, void* out, int cout, int cin, int ktaps, int transpose,
                                             int nterm, hipStream_t stream) {
    return la_pack_conv_weights_bf16(w, out, cout, cin, ktaps, transpose, nterm, stream);
}
