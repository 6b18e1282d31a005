// Internal (C++) entry points of the FIR resampling kernels; the public C ABI is in include/latentaug_hip.h.
#pragma once
#include "la_common.h"

// generic op, optional same-shape addend (skip connection add fused into the store)
// optional tail of the 4x4 kernels: activation backward of the layer whose saved output `yref` has the shape of `out` (stride-1 kernel:
// out = fir(in) * act'(yref), bias_act.py:170 with grad = 1) and / or the fp16 operand scale of `out` for the contraction that consumes
// it (slot rows [B][LA_XS_FAN], la_common.h; stride-1 and up-2 kernels)
struct LaFirTail {
    const float* yref; int act; float alpha, gain, clamp;
    float* xs_out;
    int in_pitch = 0; long in_plane = 0; int in_xhalf = 0;      // (4x4 stride-1 only) padded / column-planar input rows, as la_upfirdn2d_modconv_epilogue takes them
};
int la_upfirdn2d_ex(const float* in, float* out, int B, int C, int Hin, int Win, const float* f_host, int fh, int fw,
                    int upx, int upy, int dnx, int dny, int padx0, int padx1, int pady0, int pady1, int flip_filter,
                    float gain, const float* addend, hipStream_t stream, float* pmax = nullptr, const LaFirTail* tail = nullptr);
// pmax (optional, [B*C][la_fir4x4_segments(Hout, Wout)]): partial max |out| of every plane, one per workgroup -- lets the
// contraction that consumes `out` skip its own absmax pass (fp16 operand scale).
int la_fir4x4_segments(int Hout, int Wout);

// FIR (up=down=1) followed by the modulated-conv epilogue: *demod[b][c] + noise*strength + bias[c] -> act -> clamp.
// Used after the transposed stride-2 conv of an up-sampling SynthesisLayer (conv2d_resample.py:126).
int la_upfirdn2d_modconv_epilogue(const float* in, float* out, int B, int C, int Hin, int Win, const float* f_host,
                                  int fh, int fw, int padx0, int padx1, int pady0, int pady1, float fir_gain,
                                  const float* demod, int demod_stride, const float* noise, long noise_bstride,
                                  float noise_strength, const float* bias, int act, float alpha, float gain,
                                  float clamp, hipStream_t stream, float* pmax = nullptr, int in_pitch = 0, long in_plane = 0, int in_xhalf = 0,
                                  float* xs_out = nullptr, const float* xs_mult = nullptr, int row_lo = 0, int row_hi = 0, int col_lo = 0, int col_hi = 0);
// row_lo / row_hi (column-planar input only; 0 / 0 = all): only output rows [row_lo, row_hi) are computed and written
// xs_out / xs_mult (optional): slot rows [B][LA_XS_FAN] of the fp16 operand scale of `out` for the contraction that consumes it
// (la_common.h): lowered by the producing workgroups to pow2(xs_mult[b] * max |out|)
// in_pitch / in_plane (floats, 0 = dense): padded row pitch / plane stride of `in` (multiples of 4 select the vector kernel)
// in_xhalf (> 0, needs in_pitch): column-planar rows -- even columns of the image at [0, ceil(Win/2)), odd columns from in_xhalf on

// FIR adjoint of an up-sampling layer written straight into the stride-2 backward contraction's operand format (fp16 mode):
// q [B][ceil(C/32)][(H+1)*(W+1)][32 channels] = {h | l << 16} of xscale[b] * adjoint(in); see la_upfirdn2d.hip
int la_fir4x4_adjoint_pack_f16(const float* in, unsigned* q, const float* xscale, int xs_fan, int B, int C, int H, int W, const float* f_host,
                               float gain, hipStream_t stream, int flip_taps = 0, int in_lo = 0, int in_hi = 0, int out_lo = 0, int out_hi = 0, int in_c0 = 0, int in_c1 = 0);
// in_lo / in_hi: valid rows of `in` (the others read as zeros); out_lo / out_hi: row window of the (H+1)-row output (0 / 0 = all)
// flip_taps = 1: the forward 4x4 FIR with pad 2 (same geometry: (H+1) x (W+1) outputs) instead of the adjoint of the pad-1 FIR

// The image-gradient pyramid of a synthesis backward pass in one launch: outs[l] [planes][R0 >> (l+1)]^2 = adjoint of upsample2d applied l + 1
// times to g_top [planes][R0]^2 (per level exactly la_upfirdn2d_ex(.., down 2, pad (1,1,1,1), flip, gain 4)); R0 <= 256.
int la_image_grad_pyramid(const float* g_top, float* const* outs, int nlev, int planes, int R0, const float* f_host, hipStream_t stream);
