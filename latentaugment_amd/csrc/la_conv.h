// Implicit-GEMM convolution on the fp32 MFMA path (v_mfma_f32_32x32x2_f32), gfx950.
//
// One kernel family serves every dense contraction of the synthesis pass (DESIGN.md "conv_igemm"):
//   F1  forward 3x3 same-resolution conv          (conv2d_resample.py:132-134)
//   F2  forward transposed stride-2 conv, one launch per output phase (conv2d_resample.py:112-125)
//   B1  backward-data of F1 (flipped taps, transposed weight slab)
//   B2  backward-data of F2 (stride-2 gather over the (2h+1)^2 gradient)
// expressed as  out[b, m, g*os+oo] = sum_{t, c} wgt[tap_w[t]][c][m] * scale[b][c] * in[b, c, g*is + tap_d[t]]
#pragma once
#include "la_common.h"

#define LA_EPI_RAW 0
#define LA_EPI_FWD 1
#define LA_EPI_BWD 2
#define LA_CONV_MAX_TAPS 9
#define LA_CONV_MAX_PHASES 4
#define LA_CONV_PHASE_TAPS 4

// contraction arithmetic: exact fp32 MFMA, or fp32 operands split into 3 / 2 bf16 terms on the bf16 MFMA (la_conv_bf16.hip)
#define LA_PREC_F32 0
#define LA_PREC_BF16X3 1
#define LA_PREC_BF16X2 2
#define LA_PREC_F16X2 3     // fp32 operands scaled by a power of two and split into 2 fp16 terms, 3 fp16 MFMAs per product

struct LaConvArgs {
    const float* in;         // [B][C][Hin][Win]; in_bstride == 0 broadcasts one sample over the batch
    const float* wgt;        // [slabs][C][M]
    float* out;              // [B][M][Hout][Wout]
    const float* in_scale;   // [B][scale_stride] or null: modulate-on-load  x * s[b][c]
    long in_bstride;
    int scale_stride;
    // Launch input = in * act'(in_mask_y) * in_gain, applied while the pre-split copy is made (16-bit flat / split-K launches only: the
    // activation backward in front of a backward contraction without a sweep of its own, bias_act.py:170 with grad = 1).  in_mask_y:
    // saved output of the activation, same shape as `in`, or null (factor in_gain only; 0 is read as 1).  The operand scale must then
    // be preset (acc_scale_x) from a bound of the product, e.g. max |in| * max slope.
    const float* in_mask_y;
    int in_mask_act; float in_mask_alpha, in_mask_gain, in_mask_clamp;
    float in_gain;
    int B, C, M, Hin, Win, Hout, Wout;
    int Gy, Gx;              // output grid per sample handled by this launch
    int in_sy, in_sx;
    int out_sy, out_sx, out_oy, out_ox;
    int out_pitch; long out_plane;   // LA_EPI_RAW only, 0 = dense: row pitch / plane stride of `out` in floats (padded scratch rows, 16-byte aligned)
    int ntaps;
    int tap_dy[LA_CONV_MAX_TAPS], tap_dx[LA_CONV_MAX_TAPS], tap_w[LA_CONV_MAX_TAPS];
    int epi;
    // LA_EPI_FWD:  y = clamp(act(acc*demod[b][m] + noise*strength + bias[m]) * gain)
    const float* demod;
    int demod_stride;
    const float* noise;      // [Hout][Wout] (noise_bstride 0) or [B][Hout][Wout]
    long noise_bstride;
    float noise_strength;
    const float* bias;
    int act;
    float alpha, gain, clamp;
    const float* addend;     // optional [B][M][Hout][Wout]: out2 = y + addend (residual sum), y itself still goes to `out`
    float* out2;
    // LA_EPI_FWD, optional: the fp16 operand scale of this launch's output for the contraction that consumes it -- slot rows
    // [B][LA_XS_FAN] (holding LA_XS_INIT before) that every producing workgroup lowers to pow2(mult[b] * its max |y|) (la_xs_lower,
    // la_common.h; y = out2 where there is one); fwd_xs_mult [B] or null (1): e.g. max_c |style| of the consuming layer
    float* fwd_xs_out;
    const float* fwd_xs_mult;
    // LA_EPI_BWD:  out = acc * out_scale[b][m];  ds_part[b][m][tile] = sum_pixels acc * xin[b][m][pixel]
    const float* out_scale;
    int oscale_stride;
    const float* xin;
    long xin_bstride;
    float* ds_part;          // [B][M][tiles_per_sample]
    int tiles_per_sample;
    // LA_EPI_BWD with seam_ddn_part != null (16-bit kernels and the split-K finish pass only): the backward "seam" of the layer
    // that PRODUCED xin is applied to the outgoing gradient in the same epilogue -- xin is that layer's saved output y, which the
    // epilogue loads anyway for ds_part -- instead of a separate pass over y and the gradient (la_seam_bwd_kernel<0>):
    //   g = acc * out_scale;  g1 = g * act'(y);  seam_ddn_part[b][m][tile] = sum_px g1 * (act^-1(y) - seam_bias[m] - noise*strength);
    //   out = g1 * seam_demod[b][m];  seam_pmax[b][m][tile] = max_px |out|   (plane maxima for the next contraction's operand scale)
    const float* seam_demod; int seam_demod_stride;
    const float* seam_bias;
    const float* seam_noise; long seam_noise_bstride; float seam_noise_strength;
    int seam_act; float seam_alpha, seam_gain, seam_clamp;
    float* seam_ddn_part;    // [B][M][tiles_per_sample]
    float* seam_pmax;        // [B][M][tiles_per_sample] or null
    float* seam_xs_out;      // [B][LA_XS_FAN] or null: slot rows of the fp16 operand scale of `out` for its consumer = pow2 scale of
    float seam_xs_mult;      //   seam_xs_mult * max|out| over the sample, final when the launch has run: every kernel form (direct epilogues,
                             //   split-K finish pass) lowers the row itself (la_xs_lower, la_common.h; it must hold LA_XS_INIT before)
    // ... and, with seam_imgc > 0, the ToRGB backward of the block whose conv1 output xin is (la_seam_bwd_kernel<imgc>):
    //   g += sum_c seam_wrgb[c][m] * seam_srgb[b][m] * gr_c,  gr_c = seam_gimg[b][c][px] where |seam_rgbpre[b][c][px]| <= seam_rgb_clamp;
    //   seam_dweff_part[b][c][m][tile] = sum_px gr_c * y
    int seam_imgc;
    const float* seam_gimg; const float* seam_rgbpre; float seam_rgb_clamp;
    const float* seam_wrgb; const float* seam_srgb; int seam_srgb_stride;
    float* seam_dweff_part;  // [B][imgc][M][tiles_per_sample]
    // Fused ToRGB of the block (LA_EPI_FWD, 16-bit halo launches whose ONE row tile holds every output channel, M == tile rows):
    //   rgb_pre[b][c][px] = sum_m rgb_w[c][m] * rgb_s[b][m] * out[b][m][px] + rgb_bias[c];  rgb_img = clamp(rgb_pre) + rgb_skip
    // (la_torgb_fwd_kernel's arithmetic on the epilogue's registers: the block's conv1 output is not streamed a second time)
    int rgb_imgc;            // 0: none
    const float* rgb_w; const float* rgb_s; int rgb_s_stride; const float* rgb_bias;
    const float* rgb_skip;   // [B][imgc][Hout*Wout] or null
    float* rgb_pre; float* rgb_img; float rgb_clamp;
    // optional caller-provided scratch (la_conv_workspace_bytes): [fp16 scale header | pre-split input | split-K slice partials]
    void* ws;
    size_t ws_bytes;
    // filled in by la_conv_launch
    float* splitk_ws;
    int ksplit;
    const float* in_pmax;          // optional [B][C][in_pmax_nseg]: partial max |in| of every plane, left by the kernel that produced `in`
    int in_pmax_nseg;              //   (the fp16 operand scale then needs no absmax pass)
    const void* in_q;              // split paths: input already split by la_conv_prepare_input (flat / split-K kernels), or NULL
    // split-bf16 path (precision != LA_PREC_F32): weights pre-split by la_pack_conv_weights_bf16
    int precision;
    const void* wgt_bf16;          // split pack (la_conv_split_pack_bytes): 3 bf16 terms, 2 fp16 terms, fp16 weight scale
    long wgt_bf16_term_elems;      // elements per term: slabs * ceil(C/32) * M * 32
    // LA_PREC_F16X2: acc is divided by xscale[b] * wscale (exact powers of two) before the epilogue; set by la_conv_prepare_input
    const float* acc_scale_x;      // [B] (acc_scale_fan <= 1), or slot rows [B][LA_XS_FAN] left by the producer of `in` (acc_scale_fan = LA_XS_FAN)
    int acc_scale_fan;
    const float* acc_scale_w;      // [1]
    // Merged output phases (transposed stride-2 conv, 16-bit direct kernel only): nphase > 0 runs all phases in ONE launch,
    // blockIdx.z = phase * B + sample, each phase with its own grid / output offset / tap table (<= 4 taps).  One launch of
    // ~4x the workgroups instead of four launches that each end in a nearly empty last round.  Phase grids at the split-K sizes
    // (<= 34x34) run the same way through the split-K kernel: blockIdx.x walks the phases' flattened-pixel tiles back to back,
    // one finish launch serves all phases (blockIdx.z = phase).
    // Row window (16-bit direct kernels; a hint, 0 / 0 = all rows): only output-grid rows [row_lo, row_hi) are wanted -- pixel tiles that
    // hold none of them return at once and write nothing (merged phases: the window counts rows of every phase's own grid).  The
    // caller guarantees that nobody reads the rows left out (la_synth.hip: the loop steps of a criterion that sees a crop only).
    int row_lo, row_hi;
    // ... and a column window on top of it (halo kernel only; 0 / 0 = all columns): the wanted tiles are then the rectangle of 4 x 32 tiles
    // that holds rows [row_lo, row_hi) x columns [col_lo, col_hi); needs a row window
    int col_lo, col_hi;
    // Valid rows of the INPUT (16-bit direct kernels; 0 / 0 = all): rows outside [in_row_lo, in_row_hi) read as zeros, as rows outside the
    // image do -- the backward contractions behind a windowed producer, whose other rows hold older contents of a shared buffer while the
    // gradient there is exactly zero.  (Flat kernel: rows of the pre-split operand's grid.)
    int in_row_lo, in_row_hi;
#ifdef LA_DEV
    int dbg_stamp;                 // development build: per-wave segment clocks of the MF 5 halo kernel (la_conv_bf16.hip, LA_STAMP)
#endif
    int nphase;
    struct Phase {
        int Gy, Gx, out_oy, out_ox, ntaps; int tap_dy[LA_CONV_PHASE_TAPS], tap_dx[LA_CONV_PHASE_TAPS], tap_w[LA_CONV_PHASE_TAPS];
        int tile0;           // split-K form (filled by la_conv_launch): first flattened-pixel tile of the phase on blockIdx.x ...
        long ws_off;         // ... and the float offset of its slice partials inside splitk_ws
    } ph[LA_CONV_MAX_PHASES];
};

long la_conv_bf16_pack_elems(int M, int C, int ktaps);   // elements per term
int la_pack_conv_weights_bf16(const float* w, void* out, int cout, int cin, int ktaps, int transpose, int nterm,
                              hipStream_t stream, float scale = 1.f, int m_pad = 0);
int la_conv_bf16_dispatch(const LaConvArgs& as, int MTsel, dim3 grid, bool split, hipStream_t stream);

// scratch floats that let every <= 32x32 launch of a (B, M) problem use split-K: slices * B * M * G, G <= 1024
long la_conv_splitk_floats(int B, int M, int C, int Gy, int Gx, int precision);
long la_conv_splitk_floats_phases(int B, int M, int C, int nphase, const int* Gy, const int* Gx, int precision);
// bytes of the pre-split copy of an input [B][C][Hin][Win] (split paths; sized for the larger, 8 B/element format)
size_t la_conv_presplit_bytes(int B, int C, int Hin, int Win);
// bytes of one weight pack serving every split precision
size_t la_conv_split_pack_bytes(int M, int C, int ktaps);
// If a.precision needs a pre-split input and a.in_q is not set: split a.in (* a.in_scale) into the head of a.ws, point
// a.in_q / a.acc_scale_x at it and advance a.ws / a.ws_bytes past it.  Callers that launch several phases over one input
// call this once.  bf16: {hi | mid << 16, lo} (8 B / element);  fp16: per-sample power-of-two scale, {hi | lo << 16} (4 B).
int la_conv_prepare_input(LaConvArgs& a, hipStream_t stream);
// Activation backward fused with the plane maxima of its result (pass 1 of the fp16 operand scale of the consuming contraction):
// dx [B][C][HW] = dy * act'(yref) (dx may alias dy), pm [B][C][la_conv_act_grad_segments(HW)] -> LaConvArgs::in_pmax / in_pmax_nseg.
// Replaces la_bias_act_grad_f32 (bias_act.py:170, grad = 1) where the result feeds a contraction.
int la_conv_act_grad_segments(long HW);
int la_conv_act_grad_pmax(const float* dy, const float* yref, float* dx, float* pm, int B, int C, long HW, int act, float alpha, float gain,
                          float clamp, hipStream_t stream);
int la_conv_xscale_from_pmax(const float* pmax, int nseg, const float* scale, int scale_stride, float mult, float* xscale, int B, int C,
                             hipStream_t stream);
int la_absmax_bits(const float* w, long n, unsigned* amax_bits, hipStream_t stream);      // max |w| as a float bit pattern (zero it first)
bool la_conv_bf16_uses_halo(const LaConvArgs& a);     // fp32-input halo kernel (no pre-split copy needed)

// number of pixel tiles per sample for a launch (the ds_part leading dimension)
int la_conv_tiles_per_sample(int Gy, int Gx);
int la_conv_launch(const LaConvArgs& a, hipStream_t stream);
