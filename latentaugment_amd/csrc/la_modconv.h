// Modulated 3x3 conv entry points (C ABI; citations in include/latentaug_hip.h).
#pragma once
#include <stddef.h>
#include "la_common.h"
#include "la_style.h"

// Internal variants with plane maxima passed between producer and consumer (fp16 operand scale without an absmax pass):
//   in_pmax  [B][C][in_nseg]: partial max |input| per plane, written by the kernel that produced the input
//   y_pmax   [B][cout][la_fir4x4_segments(res, res)]: partial max |y| per plane, written by the FIR epilogue of the up-sampling layer
// ToRGB of the block fused into the epilogue of its conv1 (LaConvArgs::rgb_*): possible where one row tile of the halo kernel holds
// every output channel (la_modconv3x3_fwd_fuses_rgb); weights [imgc][cout], styles [B][s_stride] (already * weight_gain), bias [imgc],
// skip [B][imgc][res^2] or null, outputs rgb_pre / img [B][imgc][res^2]
struct LaRgbFuse {
    int imgc;
    const float* w; const float* s; int s_stride; const float* bias; const float* skip;
    float* rgb_pre; float* img; float clamp;
};
bool la_modconv3x3_fwd_fuses_rgb(int precision, int B, int cin, int cout, int res);
int la_modconv3x3_fwd_ex(const float* x, long x_bstride, const float* in_pmax, int in_nseg, const float* wf, const void* wq, int precision, const float* s,
                         int s_stride, const float* d, int d_stride, const float* noise, long noise_bstride, float noise_strength,
                         const float* bias, int act, float alpha, float gain, float clamp, float* y, void* ws, size_t ws_bytes, int B,
                         int cin, int cout, int res, hipStream_t stream, const float* xscale = nullptr, const LaRgbFuse* rgb = nullptr,
                         float* xs_out = nullptr, const float* xs_mult = nullptr, int row_lo = 0, int row_hi = 0, int col_lo = 0, int col_hi = 0);
// row_lo / row_hi, col_lo / col_hi (0 / 0 = all): row (and column) window of the output, LaConvArgs::row_lo / col_lo -- a hint: rows outside it may or may not be written
// xs_out / xs_mult (optional): the operand scale of y for the contraction that consumes it (LaConvArgs::fwd_xs_out / fwd_xs_mult)
int la_modconv3x3_up2_fwd_ex(const float* x, long x_bstride, const float* wf, const void* wq, int precision, const float* s, int s_stride,
                             const float* d, int d_stride, const float* noise, long noise_bstride, float noise_strength,
                             const float* bias, int act, float alpha, float gain, float clamp, const float* fir_host,
                             float* scratch, float* y, float* y_pmax, void* ws, size_t ws_bytes, int B, int cin, int cout, int res,
                             hipStream_t stream, const float* xscale = nullptr, int scratch_pitch = 0, int scratch_xhalf = 0,
                             float* xs_out = nullptr, const float* xs_mult = nullptr, int row_lo = 0, int row_hi = 0, int col_lo = 0, int col_hi = 0);
// col_lo / col_hi: the FIR writes these columns of the row window only (the transposed conv computes whole rows)
// row_lo / row_hi (0 / 0 = all; column-planar scratch only): row window of y -- the FIR writes exactly these rows, the transposed conv the
// rows of its intermediate they read.  la_modconv3x3_up2_fwd_rows: the input rows such a call reads.
void la_modconv3x3_up2_fwd_rows(int res, int row_lo, int row_hi, int* in_lo, int* in_hi);
// scratch_pitch / scratch_xhalf (floats; both 0 = dense (res+1)-wide rows, or both set): COLUMN-PLANAR rows of the transposed-conv
// intermediate -- the even output columns of a row at [0, res/2 + 1), the odd ones from scratch_xhalf on (a multiple of 4,
// scratch_pitch >= scratch_xhalf + res/2) -- so that every output phase of the transposed conv stores contiguous runs and the FIR
// runs its vector kernel (the scratch then holds B * cout * (res+1) * scratch_pitch floats)
// xscale (optional, [B]): preset power-of-two fp16 operand scale of the input (e.g. from the clamp bound of the producing layer)
// Backward seam of the layer that produced `xin`, applied inside the backward contraction's epilogue (LaConvArgs::seam_*):
// its demod / bias / noise / activation, and where its demod-gradient partials and plane maxima go ([B][cin][la_modconv_ds_tiles(res)]).
struct LaSeamFuse {
    const float* demod; int demod_stride;
    const float* bias;
    const float* noise; long noise_bstride; float noise_strength;
    int act; float alpha, gain, clamp;
    float* ddn_part;
    float* pmax;
    float* xs_out; float xs_mult;      // optional [B]: running operand scale of the epilogue's output for its consumer (la_xs_lower)
    // ToRGB backward of that block (imgc > 0): image gradient, ToRGB pre-clamp output, weights [imgc][C], styles, partial outputs
    int imgc;
    const float* g_img; const float* rgb_pre; float rgb_clamp;
    const float* wrgb; const float* s_rgb; int s_rgb_stride;
    float* dweff_part;       // [B][imgc][cin][la_modconv_ds_tiles(res)]
};
// Row windows of a backward launch (la_synth.hip; null = whole planes): rows [in_lo, in_hi) of gz are valid -- the others hold older
// contents of a shared buffer where the gradient is exactly zero, and read as zeros -- and only rows [out_lo, out_hi) of gx are wanted
// (0 / 0 = all; tiles outside write nothing but zero their style-gradient partials).  16-bit direct kernels; other forms ignore them.
struct LaBwdRows { int in_lo, in_hi, out_lo, out_hi; int in_c0, in_c1, out_c0, out_c1; };      // (.. and columns: valid columns of gz, column window of gx; 0 / 0 = all)
int la_modconv3x3_bwd_ex(const float* gz, const float* in_pmax, int in_nseg, const float* wb, const void* wq, int precision, const float* s, int s_stride,
                         const float* xin, long xin_bstride, float* gx, float* ds_part, void* ws, size_t ws_bytes, int B, int cin, int cout,
                         int res, hipStream_t stream, const LaSeamFuse* seam = nullptr, const float* xscale = nullptr, const LaBwdRows* rows = nullptr);
// xscale (optional, [B]): the fp16 operand scale of gz, already final when this launch starts (left by the producer of gz through
// LaSeamFuse::xs_out / LaSeamArgs::xs_out) -- no plane-maxima reduction launch

// gz_pmax [B][cout][gz_nseg]: partial max |gz| per plane (left by the seam kernel); with it the fp16 mode builds the contraction's
// operand in one fused pass (FIR adjoint + scale + split + interleave)
int la_modconv3x3_up2_bwd_ex(const float* gz, const float* gz_pmax, int gz_nseg, const float* wb, const void* wq, int precision, const float* s,
                             int s_stride, const float* xin, long xin_bstride, const float* fir_host, float* scratch, float* gx,
                             float* ds_part, void* ws, size_t ws_bytes, int B, int cin, int cout, int res, hipStream_t stream,
                             const LaSeamFuse* seam = nullptr, const float* xscale = nullptr, const LaBwdRows* rows = nullptr);
// rows (fused fp16 path only): in = valid rows of gz (res rows), out = window of gx (res/2 rows); the FIR adjoint then writes the rows of
// its (res+1)-row result that can be non-zero, [in_lo - 2, in_hi + 2), and the contraction reads the others as zeros      // seam of the block BELOW (its conv1 output is xin), incl. its ToRGB backward

float la_modconv_up2_bwd_xs_mult(const float* fir_host);      // the `mult` of the operand scale an up layer's backward expects (LaSeamFuse::xs_mult)

extern "C" {
int la_pack_conv_weights_f32(const float* w, float* wf, float* wb, float* wsq, int cout, int cin, int ktaps, hipStream_t);
int la_modconv3x3_fwd_f32(const float* x, long x_bstride, const float* wf, const void* wq, int precision, const float* s, int s_stride, const float* d,
                          int d_stride, const float* noise, long noise_bstride, float noise_strength, const float* bias,
                          int act, float alpha, float gain, float clamp, float* y, void* ws, size_t ws_bytes, int B, int cin, int cout, int res,
                          hipStream_t stream);
int la_modconv3x3_up2_fwd_f32(const float* x, long x_bstride, const float* wf, const void* wq, int precision, const float* s, int s_stride,
                              const float* d, int d_stride, const float* noise, long noise_bstride, float noise_strength,
                              const float* bias, int act, float alpha, float gain, float clamp, const float* fir_host,
                              float* scratch, float* y, void* ws, size_t ws_bytes, int B, int cin, int cout, int res, hipStream_t stream);
int la_modconv3x3_bwd_f32(const float* gz, const float* wb, const void* wq, int precision, const float* s, int s_stride, const float* xin,
                          long xin_bstride, float* gx, float* ds_part, void* ws, size_t ws_bytes, int B, int cin, int cout, int res, hipStream_t);
int la_modconv3x3_up2_bwd_f32(const float* gz, const float* wb, const void* wq, int precision, const float* s, int s_stride, const float* xin,
                              long xin_bstride, const float* fir_host, float* scratch, float* gx, float* ds_part, void* ws, size_t ws_bytes, int B,
                              int cin, int cout, int res, hipStream_t stream);
int la_modconv_ds_tiles(int grid_res);
size_t la_modconv_workspace_bytes(int B, int cin, int cout, int res, int up);
size_t la_modconv_bf16_pack_bytes(int cin, int cout, int transpose, int nterm);
int la_pack_conv_weights_bf16_f32(const float* w, void* out, int cout, int cin, int ktaps, int transpose, int nterm,
                                  hipStream_t stream);
}
