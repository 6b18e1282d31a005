// Internal (C++) interface of la_style.hip.
#pragma once
#include "la_common.h"

#define LA_MAX_STYLE_LAYERS 40

// One row block per affine layer (conv layers first, then ToRGB layers); rows = that layer's in_channels.
struct LaStyleTable {
    int nlayers;
    int total_rows;
    int row_start[LA_MAX_STYLE_LAYERS + 1];   // row_start[nlayers] == total_rows
    int widx[LA_MAX_STYLE_LAYERS];            // which w of ws[b][.] feeds the layer
    float post_gain[LA_MAX_STYLE_LAYERS];     // 1 for conv layers, 1/sqrt(cin) for ToRGB (weight_gain folded into the style)
    const float* aw[LA_MAX_STYLE_LAYERS];     // affine.weight [cin][wdim]
    const float* ab[LA_MAX_STYLE_LAYERS];     // affine.bias   [cin]
};

struct LaDemodTable {
    int nlayers;
    int total_rows;
    int row_start[LA_MAX_STYLE_LAYERS + 1];
    int cin[LA_MAX_STYLE_LAYERS];
    int s_off[LA_MAX_STYLE_LAYERS];           // offset of the layer's styles inside a row of s_all
    const float* wsq[LA_MAX_STYLE_LAYERS];    // [cout][cin]
};

struct LaSeamArgs {
    const float* y;          // [B][C][HW] saved post-activation output of the layer
    const float* gx_next;    // [B][C][HW] or null
    float* gz;               // [B][C][HW] (may alias gx_next)
    long HW;
    int C;
    const float* demod; int demod_stride;
    const float* bias;
    const float* noise; long noise_bstride; float noise_strength;
    int act; float alpha, gain, clamp;
    float* ddn_part;         // [B][C][slabs]
    // ToRGB part (imgc > 0)
    const float* g_img;      // [B][imgc][HW] gradient w.r.t. the image at this resolution
    const float* rgb_pre;    // [B][imgc][HW] ToRGB output before its clamp
    float rgb_clamp;
    const float* wrgb;       // [imgc][C]
    const float* s_rgb; int s_stride;   // styles of the ToRGB layer (already * weight_gain)
    float* dweff_part;       // [B][imgc][C][slabs]
    float* pmax_out;         // optional [B][C][slabs]: partial max |gz| per plane (one per workgroup)
    float* xs_out; float xs_mult;      // optional slot rows [B][LA_XS_FAN] (la_common.h): fp16 operand scale of gz for its consumer, pow2 scale of xs_mult * max|gz|
    long p_lo, p_hi;         // pixel window (multiples of 4; 0 / 0 = the whole plane): only pixels [p_lo, p_hi) of every plane are read and written --
                             // the incoming gradient is zero outside them (la_synth.hip: row windows); the slabs share the window
};

int la_pack_conv_weights(const float* w, float* wf, float* wb, float* wsq, int cout, int cin, int ktaps, hipStream_t,
                         float scale = 1.f, int wb_ld = 0);   // wb_ld > cin: backward slab rows padded with zero columns
int la_affine_forward(const LaStyleTable& t, const float* ws, long ws_bstride, long ws_lstride, int B, int wdim,
                      float* s_all, hipStream_t);
// xs[l][b] = power-of-two fp16 operand scale of conv layer l's forward contraction from the bound  bound[l] * max_i |s[b][i]|
// start of a pass: resets the slot rows xs [nlayers][B][LA_XS_FAN] (layer 0: the scale of the constant input `cst`, cst_n floats) and
// xs_bwd (optional) and writes xs_mult [nlayers][B] = max_i |style| of every layer (la_style.hip)
int la_xscale_from_bounds(const LaDemodTable& t, const float* s_all, int s_stride, const float* cst, int cst_n, float* xs, float* xs_mult, int B,
                          hipStream_t, float* xs_bwd = nullptr);
int la_demod_forward(const LaDemodTable& t, const float* s_all, int s_stride, int B, float* d_all, hipStream_t);
// mask (optional, planes above 64x64): x is a gradient still to be taken through an activation -- x[b][i][p] * act'(mask.y[b][i][p]) is what
// the 1x1 reads (the discriminator's FromRGB backward: one stream of the gradient and the saved output instead of a sweep + a stream)
struct LaTorgbMask { const float* y; int act; float alpha, gain, clamp; };
int la_torgb_forward(const float* x, const float* wrgb, const float* s, int s_stride, const float* bias,
                     const float* skip, float* rgb_pre, float* img, int B, int C, int imgc, int H, int W, float clamp,
                     hipStream_t, const LaTorgbMask* mask = nullptr, int row_lo = 0, int row_hi = 0, const float* skip_lo = nullptr,
                     const float* fir_host = nullptr);      // skip_lo + fir_host: the block below's image, up-sampled inside the kernel
// row_lo / row_hi (planes above 64x64; 0 / 0 = all): only these rows of rgb_pre / img are computed and written
int la_seam_slabs(long HW);
int la_seam_backward(const LaSeamArgs& a, int B, int imgc, hipStream_t);
int la_style_backward_conv(float* ds_part, int ntiles, float* ddn_part, int nslabs, const float* d,
                           int d_stride, const float* s, int s_stride, const float* wsq, int cin, int cout, int B,
                           float* ds_out, int ds_stride, hipStream_t);
int la_style_backward_rgb(const float* dweff_part, int nslabs, const float* wrgb, int C, int imgc, int B,
                          float* ds_out, int ds_stride, hipStream_t);
// Style-gradient finish of ALL layers of one backward pass in three launches (row sums of every partial buffer, conv layers,
// ToRGB layers) instead of three small launches per layer: the per-layer partial buffers are kept until the end of the pass.
#define LA_FIN_MAX_CONV 24
#define LA_FIN_MAX_RGB 12
struct LaStyleFinish {
    int nconv, nrgb, imgc, d_stride, s_stride, ds_stride;
    struct Conv { float* ds_part; float* ddn_part; const float* d; const float* s; const float* wsq; float* ds_out; int ntiles, nslabs, cin, cout, blk0; } conv[LA_FIN_MAX_CONV];
    struct Rgb { float* dweff_part; const float* wrgb; float* ds_out; int nslabs, C, blk0; } rgb[LA_FIN_MAX_RGB];
};
int la_style_backward_all(const LaStyleFinish& f, int B, hipStream_t stream);
int la_affine_bwd_chunks(const LaStyleTable& t);   // part scratch = chunks * B * wdim floats
int la_affine_backward(const LaStyleTable& t, const float* ds_all, int B, int wdim, float* dws, int num_ws, float* part,
                       hipStream_t);
