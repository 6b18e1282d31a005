// StyleGAN2 synthesis network (architecture 'skip') forward + backward-to-latent as a sequence of HIP launches.
// Mirrors the call  G.synthesis(ws, noise_mode=...)  at augments/utils/util_latent_aug.py:227,488 and the autograd
// backward of it that loss.backward() at :275 performs, restricted to d/d(ws) (G is frozen: :480).
// Formulation: non-fused modulated conv (x*s -> shared-weight contraction -> *demod), see DESIGN.md.
#include "la_synth.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "la_conv.h"
#include "la_modconv.h"
#include "la_style.h"
#include "la_upfirdn2d.h"

#define MAX_BLOCKS 12

struct ConvLayer {
    int cin, cout, res, up, widx;
    const float *affine_w, *affine_b, *weight, *bias, *noise_const;
    float noise_strength;
    float *wf, *wb, *wsq;   // packed (fp32)
    void *wqf, *wqb;        // packed split-bf16 (3 terms), forward / backward
    float* y;               // saved output [maxB][cout][res*res]
    float *dsp, *ddnp;      // style-gradient partials of this layer (kept until the one-pass finish at the end of a backward pass)
    int s_off, d_off, style_idx;
    const float* noise_used;   // set by forward (null when the term vanishes)
    long noise_bstride;
};

struct RgbLayer {
    int cin, res, widx;
    const float *affine_w, *affine_b, *weight, *bias;
    int s_off, style_idx;
    float *rgb_pre, *img, *g_img;
    float* dwep;            // ToRGB weight-gradient partials [maxB][imgc][cin][slabs]
};

struct la_synth {
    int R, imgc, wdim, nblocks, num_ws, maxB;
    int channels[MAX_BLOCKS];
    float clamp;
    const float* cst;   // b4.const [C4][4][4]
    int nconv;
    ConvLayer conv[2 * MAX_BLOCKS];
    RgbLayer rgb[MAX_BLOCKS];
    float fir[16];
    LaStyleTable st;
    LaDemodTable dt;
    int S, Dt;
    float *s_all, *d_all, *ds_all;
    float *zT, *G0, *G1, *aff_part;
    void* cws;
    size_t cws_bytes;
    float* pmax;             // [maxB][max channels]: plane maxima handed from a producing kernel to the next contraction (fp16 mode)
    float* pmax3;            // the same for a block's conv1 output when ITS seam (incl. ToRGB backward) is fused into the epilogue of the
                             // up-sampling layer's backward contraction of the block above
    float* pmax2;            // [maxB][cout][tiles]: plane maxima of an up-sampling layer's gz when its seam is fused into the epilogue of
                             // the conv1 backward contraction above it
    float* xs_fwd;           // [nconv][B][LA_XS_FAN] slot rows: fp16 operand scales of the forward contractions (from the clamp bound, one launch per pass)
    float* xs_bwd;           // [nconv][B][LA_XS_FAN] slot rows: running fp16 operand scales of the backward contractions' inputs, lowered by the producing kernels
    float* xs_mult;          // [nconv][B]: max_i |style| of every conv layer = what the producer of its input multiplies its max |x| with
    int lastB;
    int precision;
    float* final_img;   // where the last forward put the full-resolution image
    int win_lo, win_hi;  // row window of the image the next forward passes are asked for (la_synth_set_row_window; 0 / 0 = the whole image)
    int wcol_lo, wcol_hi;      // ... and its column window (la_synth_set_col_window; top block only)
    int fw_c0, fw_c1;          // column window (32-column tiles) of the top block's conv1 output in the LAST forward pass (0 / 0 = all)
    int fw_lo[2 * MAX_BLOCKS], fw_hi[2 * MAX_BLOCKS];      // row windows of the conv outputs in the LAST forward pass (0 / 0 = whole plane)
};

static size_t align_up(size_t v) { return (v + 63) & ~(size_t)63; }

struct Carver {
    char* base; size_t off; size_t cap;
    float* take(size_t nfloats) {
        size_t bytes = align_up(nfloats * sizeof(float));
        float* p = base ? (float*)(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

// column-planar intermediate of an up layer (la_modconv3x3_up2_fwd_ex): floats from the start of a row to its odd-column run
static inline int zt_xhalf(int res) { return (res / 2 + 1 + 3) & ~3; }
static int layout(la_synth* h, void* workspace, size_t cap, size_t* need) {
    Carver c{(char*)workspace, 0, cap};
    const size_t mb = h->maxB;
    size_t gmax = 0, ztmax = 0, dsp = 0, ddn = 0, dwe = 0;
    size_t skf = 0;
    for (int k = 0; k < h->nconv; ++k) {
        ConvLayer& L = h->conv[k];
        const size_t wn = (size_t)L.cin * L.cout;
        L.wf = c.take(9 * wn); L.wb = c.take(9 * wn); L.wsq = c.take(wn);
        L.wqf = c.take((la_modconv_bf16_pack_bytes(L.cin, L.cout, 0, 3) + 3) / 4);
        L.wqb = c.take((la_modconv_bf16_pack_bytes(L.cin, L.cout, 1, 3) + 3) / 4);
        const size_t hw = (size_t)L.res * L.res;
        L.y = c.take(mb * L.cout * hw);
        if (mb * L.cout * hw > gmax) gmax = mb * L.cout * hw;
        if (mb * L.cin * hw > gmax && !L.up) gmax = mb * L.cin * hw;
        if (L.up) {      // transposed-conv intermediate, column-planar rows (zt_xhalf)
            const size_t z = mb * L.cout * (size_t)(L.res + 1) * (size_t)(2 * zt_xhalf(L.res));
            if (z > ztmax) ztmax = z;
        }
        const int gin = L.up ? L.res / 2 : L.res;   // grid of the backward-data conv
        {
            const size_t sl0 = (size_t)la_seam_slabs((long)hw), t1 = (size_t)la_conv_tiles_per_sample(L.res, L.res);
            L.dsp = c.take(mb * L.cin * (size_t)la_conv_tiles_per_sample(gin, gin));
            L.ddnp = c.take(mb * L.cout * (sl0 > t1 ? sl0 : t1));
        }
        const size_t t = mb * L.cin * (size_t)la_conv_tiles_per_sample(gin, gin);
        if (t > dsp) dsp = t;
        const size_t sl = mb * L.cout * (size_t)la_seam_slabs((long)hw);
        if (sl > ddn) ddn = sl;
        const size_t sk = la_modconv_workspace_bytes((int)mb, L.cin, L.cout, L.res, L.up);
        if (sk > skf) skf = sk;
    }
    for (int k = 0; k < h->nblocks; ++k) {
        RgbLayer& T = h->rgb[k];
        const size_t hw = (size_t)T.res * T.res;
        T.rgb_pre = c.take(mb * h->imgc * hw);
        T.img = c.take(mb * h->imgc * hw);
        T.g_img = c.take(mb * h->imgc * hw);
        {
            const size_t sl0 = (size_t)la_seam_slabs((long)hw), t1 = (size_t)la_conv_tiles_per_sample(T.res, T.res);
            T.dwep = c.take(mb * h->imgc * T.cin * (sl0 > t1 ? sl0 : t1));
        }
        const size_t sl = mb * h->imgc * T.cin * (size_t)la_seam_slabs((long)hw);
        if (sl > dwe) dwe = sl;
    }
    h->s_all = c.take(mb * h->S);
    h->ds_all = c.take(mb * h->S);
    h->d_all = c.take(mb * h->Dt);
    h->zT = c.take(ztmax);
    h->G0 = c.take(gmax);
    h->G1 = c.take(gmax);
    h->aff_part = c.take((size_t)la_affine_bwd_chunks(h->st) * mb * h->wdim);
    h->cws = c.take((skf + 3) / 4);
    h->cws_bytes = skf;
    {
        // plane-maxima hand-off buffer: [B][C][segments] of the largest producer (FIR epilogue of an up layer, seam kernel)
        size_t pmx = 0;
        for (int k = 0; k < h->nconv; ++k) {
            const ConvLayer& L = h->conv[k];
            const size_t a = (size_t)L.cout * (L.up ? la_fir4x4_segments(L.res, L.res) : 1), b2 = (size_t)L.cout * la_seam_slabs((long)L.res * L.res);
            if (a > pmx) pmx = a;
            if (b2 > pmx) pmx = b2;
        }
        h->pmax = c.take(mb * pmx);
    }
    {
        size_t f2 = 0;
        for (int k = 0; k < h->nconv; ++k) {
            const ConvLayer& L = h->conv[k];
            if (L.up) { const size_t n = mb * L.cout * (size_t)la_conv_tiles_per_sample(L.res, L.res); if (n > f2) f2 = n; }
        }
        h->pmax2 = c.take(f2 ? f2 : 16);
        size_t f3 = 0;
        for (int k = 0; k < h->nconv; ++k) {
            const ConvLayer& L = h->conv[k];
            if (!L.up) { const size_t n = mb * L.cout * (size_t)la_conv_tiles_per_sample(L.res, L.res); if (n > f3) f3 = n; }
        }
        h->pmax3 = c.take(f3 ? f3 : 16);
    }
    h->xs_fwd = c.take((size_t)h->nconv * mb * LA_XS_FAN);
    h->xs_bwd = c.take((size_t)h->nconv * mb * LA_XS_FAN);
    h->xs_mult = c.take((size_t)h->nconv * mb + 16);
    *need = c.off;
    return LA_OK;
}

static int describe(la_synth* h, int R, int imgc, int wdim, const int* channels, int maxB) {
    LA_CHECK_ARG(R >= 4 && (R & (R - 1)) == 0 && R <= 4096, "synth: resolution must be a power of two >= 4");
    LA_CHECK_ARG(imgc >= 1 && imgc <= 4, "synth: img_channels must be 1..4");
    LA_CHECK_ARG(wdim >= 1 && maxB >= 1, "synth: bad w_dim / batch");
    memset(h, 0, sizeof(*h));
    h->R = R; h->imgc = imgc; h->wdim = wdim; h->maxB = maxB;
    int nb = 0;
    for (int r = 4; r <= R; r *= 2) ++nb;
    LA_CHECK_ARG(nb <= MAX_BLOCKS, "synth: too many blocks");
    h->nblocks = nb;
    int nconv = 0, widx = 0, S = 0, Dt = 0, nst = 0;
    for (int k = 0; k < nb; ++k) {
        const int res = 4 << k;
        const int co = channels[k];
        LA_CHECK_ARG(co >= 4 && co % 4 == 0, "synth: channel counts must be multiples of 4");
        h->channels[k] = co;
        if (k > 0) {
            ConvLayer& L = h->conv[nconv++];
            L.cin = channels[k - 1]; L.cout = co; L.res = res; L.up = 1; L.widx = widx++;
        }
        ConvLayer& L1 = h->conv[nconv++];
        L1.cin = co; L1.cout = co; L1.res = res; L1.up = 0; L1.widx = widx++;
        RgbLayer& T = h->rgb[k];
        T.cin = co; T.res = res; T.widx = widx;   // shares its w with the next block's conv0
    }
    h->nconv = nconv;
    h->num_ws = widx + 1;
    LA_CHECK_ARG(nconv + nb <= LA_MAX_STYLE_LAYERS, "synth: too many style layers");
    // style rows: conv layers then rgb layers
    for (int k = 0; k < nconv; ++k) {
        ConvLayer& L = h->conv[k];
        L.s_off = S; L.d_off = Dt; L.style_idx = nst;
        h->st.row_start[nst] = S; h->st.widx[nst] = L.widx; h->st.post_gain[nst] = 1.f;
        h->dt.row_start[k] = Dt; h->dt.cin[k] = L.cin; h->dt.s_off[k] = S;
        S += L.cin; Dt += L.cout; ++nst;
    }
    for (int k = 0; k < nb; ++k) {
        RgbLayer& T = h->rgb[k];
        T.s_off = S; T.style_idx = nst;
        h->st.row_start[nst] = S; h->st.widx[nst] = T.widx; h->st.post_gain[nst] = 1.f / sqrtf((float)T.cin);
        S += T.cin; ++nst;
    }
    h->st.nlayers = nst; h->st.total_rows = S; h->st.row_start[nst] = S;
    h->dt.nlayers = nconv; h->dt.total_rows = Dt; h->dt.row_start[nconv] = Dt;
    h->S = S; h->Dt = Dt;
    return LA_OK;
}

extern "C" int la_synth_num_ws(int img_resolution) {
    int nb = 0;
    for (int r = 4; r <= img_resolution; r *= 2) ++nb;
    return 2 * nb;   // 2*log2(R) - 2
}

extern "C" int la_synth_num_params(int img_resolution) {
    int nb = 0;
    for (int r = 4; r <= img_resolution; r *= 2) ++nb;
    return 1 + 5 + 4 + (nb - 1) * 14;
}

extern "C" size_t la_synth_workspace_bytes(int img_resolution, int img_channels, int w_dim, const int* channels,
                                           int max_batch) {
    la_synth* h = (la_synth*)malloc(sizeof(la_synth));
    if (!h) return 0;
    size_t need = 0;
    if (describe(h, img_resolution, img_channels, w_dim, channels, max_batch) == LA_OK) layout(h, nullptr, 0, &need);
    free(h);
    return need;
}

extern "C" int la_synth_create(int img_resolution, int img_channels, int w_dim, const int* channels, float conv_clamp,
                               const float* const* params, int nparams, const float* noise_strength, int nlayers,
                               const float* fir_host, int fir_h, int fir_w, int max_batch, void* workspace,
                               size_t workspace_bytes, hipStream_t stream, la_synth** out) {
    LA_CHECK_ARG(params && noise_strength && fir_host && workspace && out, "synth_create: null pointer");
    LA_CHECK_ARG(fir_h == 4 && fir_w == 4, "synth_create: resample filter must be 4x4 (setup_filter([1,3,3,1]))");
    la_synth* h = (la_synth*)malloc(sizeof(la_synth));
    LA_CHECK_ARG(h, "synth_create: out of host memory");
    int rc = describe(h, img_resolution, img_channels, w_dim, channels, max_batch);
    if (rc) { free(h); return rc; }
    if (nparams != la_synth_num_params(img_resolution) || nlayers != h->nconv) {
        free(h); la_set_error("synth_create: parameter list length mismatch"); return LA_ERR_ARG;
    }
    for (int i = 0; i < nparams; ++i)
        if (!params[i]) { free(h); la_set_error("synth_create: null parameter tensor"); return LA_ERR_ARG; }
    size_t need = 0;
    layout(h, workspace, workspace_bytes, &need);
    if (need > workspace_bytes) { free(h); la_set_error("synth_create: workspace too small"); return LA_ERR_WORKSPACE; }
    h->clamp = conv_clamp;
    h->win_lo = h->win_hi = 0; h->wcol_lo = h->wcol_hi = 0; h->fw_c0 = h->fw_c1 = 0;
    // every buffer starts as zeros: a forward pass restricted to a row window (la_synth_set_row_window) leaves the other rows of its saved
    // activations as they were, and the backward pass multiplies them with gradients that are exactly zero there -- they must be finite
    if (hipMemsetAsync(workspace, 0, need, stream) != hipSuccess) { free(h); la_set_error("synth_create: clearing the workspace failed"); return LA_ERR_HIP; }
    memcpy(h->fir, fir_host, sizeof(float) * 16);
    int p = 0, ci = 0;
    for (int k = 0; k < h->nblocks; ++k) {
        if (k == 0) h->cst = params[p++];
        const int nl = (k == 0) ? 1 : 2;
        for (int q = 0; q < nl; ++q) {
            ConvLayer& L = h->conv[ci];
            L.affine_w = params[p++]; L.affine_b = params[p++]; L.weight = params[p++]; L.bias = params[p++];
            L.noise_const = params[p++];
            L.noise_strength = noise_strength[ci];
            h->st.aw[L.style_idx] = L.affine_w; h->st.ab[L.style_idx] = L.affine_b;
            h->dt.wsq[ci] = L.wsq;
            rc = la_pack_conv_weights(L.weight, L.wf, L.wb, L.wsq, L.cout, L.cin, 9, stream);
            if (!rc) rc = la_pack_conv_weights_bf16_f32(L.weight, L.wqf, L.cout, L.cin, 9, 0, 3, stream);
            if (!rc) rc = la_pack_conv_weights_bf16_f32(L.weight, L.wqb, L.cout, L.cin, 9, 1, 3, stream);
            if (rc) { free(h); return rc; }
            ++ci;
        }
        RgbLayer& T = h->rgb[k];
        T.affine_w = params[p++]; T.affine_b = params[p++]; T.weight = params[p++]; T.bias = params[p++];
        h->st.aw[T.style_idx] = T.affine_w; h->st.ab[T.style_idx] = T.affine_b;
    }
    h->lastB = 0;
    *out = h;
    return LA_OK;
}

extern "C" void la_synth_destroy(la_synth* h) { free(h); }

// contraction arithmetic of every modulated conv of the engine: 0 fp32 MFMA, 1 split-bf16 x3 (fp32-class), 2 split-bf16 x2
extern "C" int la_synth_set_precision(la_synth* h, int precision) {
    LA_CHECK_ARG(h && precision >= 0 && precision <= 3, "synth_set_precision: precision must be 0..3");
    h->precision = precision;
    return LA_OK;
}
extern "C" int la_synth_get_precision(const la_synth* h) { return h ? h->precision : -1; }
// Kept for ABI compatibility (rounds 1-3 selected between an a-priori bound and data maxima for the fp16 operand scale of the forward
// contractions): since round 4 every forward scale is derived from the data of the pass by the kernel that produces the tensor
// (la_xscale_bound_kernel, la_style.hip) -- there is nothing to select, the launch sequence does not depend on this call.
extern "C" int la_synth_set_operand_scale(la_synth* h, int from_data) {
    LA_CHECK_ARG(h && (from_data == 0 || from_data == 1), "synth_set_operand_scale: 0 or 1");
    return LA_OK;
}

// Row window of the IMAGE that the following forward passes must deliver (rows [row_lo, row_hi) of the img_resolution rows; 0, 0 = all):
// the latent-optimisation loop whose criteria read a centre crop only (la_latent_opt.hip).  The 16-bit forward kernels of the blocks at
// >= 64^2 then compute only the rows that window depends on (3x3 taps, FIR taps and the up-sampling geometry followed down the
// blocks, tile granularities included: everything that IS computed is exact); the rest of every buffer keeps older contents.  The
// backward pass is unchanged: image-gradient rows outside the window must be zero (they are for a criterion that reads the window only).
extern "C" int la_synth_set_row_window(la_synth* h, int row_lo, int row_hi) {
    LA_CHECK_ARG(h && row_lo >= 0 && (row_hi == 0 ? row_lo == 0 : (row_hi > row_lo && row_hi <= h->R)), "synth_set_row_window: bad window");
    h->win_lo = row_lo; h->win_hi = row_hi;
    return LA_OK;
}

// Column window of the image on top of the row window (0, 0 = all columns).  Only the TOP block follows it -- its conv1 (forward: the 32-column
// tiles that hold the window; backward: the same tiles, the gradient of its input is non-zero one column further at most), the FIR that
// feeds it and the FIR adjoint behind it (which reads the other columns as zeros); every other kernel computes whole rows.
extern "C" int la_synth_set_col_window(la_synth* h, int col_lo, int col_hi) {
    LA_CHECK_ARG(h && col_lo >= 0 && (col_hi == 0 ? col_lo == 0 : (col_hi > col_lo && col_hi <= h->R)), "synth_set_col_window: bad window");
    h->wcol_lo = col_lo; h->wcol_hi = col_hi;
    return LA_OK;
}

#ifdef LA_DEV
// development build: the transposed-conv intermediate of the LAST up-sampling layer that ran (column-planar rows; scripts/exp_overlap_*.py)
extern "C" const float* la_synth_dev_zt(const la_synth* h) { return h ? h->zT : nullptr; }
#endif
extern "C" const float* la_synth_image(const la_synth* h) { return h ? h->final_img : nullptr; }
extern "C" const float* la_synth_block_image(const la_synth* h, int k) { return (h && k >= 0 && k < h->nblocks) ? h->rgb[k].img : nullptr; }
extern "C" const float* la_synth_layer_output(const la_synth* h, int k) { return (h && k >= 0 && k < h->nconv) ? h->conv[k].y : nullptr; }
extern "C" const float* la_synth_styles(const la_synth* h) { return h ? h->s_all : nullptr; }
extern "C" const float* la_synth_style_grads(const la_synth* h) { return h ? h->ds_all : nullptr; }
extern "C" int la_synth_style_rows(const la_synth* h) { return h ? h->S : 0; }

extern "C" int la_synth_forward(la_synth* h, const float* ws, long ws_bstride, long ws_lstride, int B, int noise_mode,
                                const float* const* noises, float* img_out, hipStream_t stream) {
    LA_CHECK_ARG(h && ws, "synth_forward: null pointer");
    LA_CHECK_ARG(B >= 1 && B <= h->maxB, "synth_forward: batch exceeds the max_batch the workspace was sized for");
    LA_CHECK_ARG(noise_mode >= 0 && noise_mode <= 2, "synth_forward: noise_mode must be 0 (none), 1 (const), 2 (explicit)");
    LA_CHECK_ARG(noise_mode != 2 || noises, "synth_forward: explicit noise requested but no noise tensors given");
    int rc;
    if ((rc = la_affine_forward(h->st, ws, ws_bstride, ws_lstride, B, h->wdim, h->s_all, stream))) return rc;
    if ((rc = la_demod_forward(h->dt, h->s_all, h->S, B, h->d_all, stream))) return rc;
    h->lastB = B;
    const bool f16 = h->precision == LA_PREC_F16X2;
    // scratch layout of the up layers: column-planar rows (dev knob LA_NO_ZT_PITCH: dense interleaved rows + the scalar FIR kernel)
    static const bool zt_dense = la_dev_env("LA_NO_ZT_PITCH") != nullptr;
    // fp16 operand scales: the slot rows of every layer start the pass at LA_XS_INIT; the kernel that PRODUCES a layer's input lowers
    // that layer's row (forward: xs_fwd, la_xs_lower with max|style| of the consuming layer; backward: xs_bwd)
    if (f16 && (rc = la_xscale_from_bounds(h->dt, h->s_all, h->S, h->cst, h->channels[0] * 16, h->xs_fwd, h->xs_mult, B, stream, h->xs_bwd))) return rc;
    auto fwd_row = [&](int conv_index) { return (f16 && conv_index < h->nconv) ? h->xs_fwd + (long)conv_index * B * LA_XS_FAN : nullptr; };
    auto fwd_mult = [&](int conv_index) { return (f16 && conv_index < h->nconv) ? h->xs_mult + (long)conv_index * B : nullptr; };
    // row windows of the conv outputs for an image row window (la_synth_set_row_window), top block downwards until a window is the
    // whole plane: conv1 in 4-row tiles around what the image / the block above need, conv0 (FIR output) one row more on either side,
    // the block below what conv0's transposed conv reads (la_modconv3x3_up2_fwd_rows) and what the image up-sampling reads
    int wlo[2 * MAX_BLOCKS], whi[2 * MAX_BLOCKS], ilo[MAX_BLOCKS], ihi[MAX_BLOCKS];      // (ilo / ihi: image rows a block has to deliver)
    for (int i = 0; i < h->nconv; ++i) wlo[i] = whi[i] = 0;
    for (int i = 0; i < MAX_BLOCKS; ++i) ilo[i] = ihi[i] = 0;
    if (h->win_hi > 0 && h->precision != LA_PREC_F32 && !zt_dense) {
        int img_lo = h->win_lo, img_hi = h->win_hi, up_lo = 0, up_hi = 0;      // needs of block k: image rows, rows read by block k + 1
        for (int k = h->nblocks - 1, c1 = h->nconv - 1; k >= 1; --k, c1 -= 2) {
            const int res = 4 << k;
            if (res < 64) break;
            int lo = img_lo, hi = img_hi;
            if (up_hi > 0) { lo = up_lo < lo ? up_lo : lo; hi = up_hi > hi ? up_hi : hi; }
            lo &= ~3; hi = (hi + 3) & ~3;
            if (hi > res) hi = res;
            if (lo <= 0 && hi >= res) break;                      // everything is needed from here down
            wlo[c1] = lo; whi[c1] = hi;
            ilo[k] = img_lo; ihi[k] = img_hi;
            const int l0 = lo - 1 > 0 ? lo - 1 : 0, h0 = hi + 1 < res ? hi + 1 : res;
            wlo[c1 - 1] = l0; whi[c1 - 1] = h0;
            la_modconv3x3_up2_fwd_rows(res, l0, h0, &up_lo, &up_hi);
            img_lo = (img_lo - 2) >> 1; if (img_lo < 0) img_lo = 0;
            img_hi = (img_hi >> 1) + 1; if (img_hi > res / 2) img_hi = res / 2;
        }
    }
    for (int i = 0; i < h->nconv; ++i) { h->fw_lo[i] = wlo[i]; h->fw_hi[i] = whi[i]; }
    // column window: the top block's conv1 in whole 32-column tiles, the FIR in front of it one column more on either side.  Needs a tile
    // column to spare on both sides (the gradient of conv1's input reaches one column beyond the image window: it must stay inside the tiles)
    int c1lo = 0, c1hi = 0, c0lo = 0, c0hi = 0;
    h->fw_c0 = h->fw_c1 = 0;
    if (h->wcol_hi > 0 && whi[h->nconv - 1] > 0 && h->R >= 64) {
        c1lo = h->wcol_lo & ~31; c1hi = (h->wcol_hi + 31) & ~31; if (c1hi > h->R) c1hi = h->R;
        if (h->wcol_lo - 1 >= c1lo && h->wcol_hi + 1 <= c1hi && (c1lo > 0 || c1hi < h->R)) {
            c0lo = c1lo - 1 > 0 ? c1lo - 1 : 0; c0hi = c1hi + 1 < h->R ? c1hi + 1 : h->R;
            h->fw_c0 = c1lo; h->fw_c1 = c1hi;
        } else c1lo = c1hi = 0;
    }
    int ci = 0;
    const float* x = h->cst;
    long x_bstride = 0;
    for (int k = 0; k < h->nblocks; ++k) {
        const int res = 4 << k;
        const int nl = (k == 0) ? 1 : 2;
        RgbLayer& T = h->rgb[k];
        float* const rgb_dst = (k == h->nblocks - 1 && img_out) ? img_out : T.img;
        // ToRGB inside the epilogue of the block's conv1 where one row tile of the halo kernel holds all its channels (the top blocks:
        // <= 128 channels at >= 64^2) -- conv1's output is then not streamed a second time
        const bool fuse_rgb = la_modconv3x3_fwd_fuses_rgb(h->precision, B, h->conv[ci + nl - 1].cin, h->conv[ci + nl - 1].cout, res);
        const float* skip = nullptr;
        if (k > 0 && fuse_rgb) {      // (the skip image has to exist before conv1 runs)
            RgbLayer& P = h->rgb[k - 1];
            if ((rc = la_upfirdn2d_ex(P.img, T.g_img, B, h->imgc, res / 2, res / 2, h->fir, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1, 0, 4.f, nullptr, stream)))
                return rc;
            skip = T.g_img;
        }
        for (int q = 0; q < nl; ++q, ++ci) {
            ConvLayer& L = h->conv[ci];
            L.noise_used = nullptr; L.noise_bstride = 0;
            if (L.noise_strength != 0.f) {
                if (noise_mode == 1) L.noise_used = L.noise_const;
                else if (noise_mode == 2) { L.noise_used = noises[ci]; L.noise_bstride = (long)res * res; LA_CHECK_ARG(L.noise_used, "synth_forward: missing noise tensor"); }
            }
            const float sq2 = sqrtf(2.f);
            LaRgbFuse rf;
            rf.imgc = h->imgc; rf.w = T.weight; rf.s = h->s_all + T.s_off; rf.s_stride = h->S; rf.bias = T.bias; rf.skip = skip;
            rf.rgb_pre = T.rgb_pre; rf.img = rgb_dst; rf.clamp = h->clamp;
            if (!L.up) {
                rc = la_modconv3x3_fwd_ex(x, x_bstride, nullptr, 0, L.wf, L.wqf, h->precision, h->s_all + L.s_off, h->S, h->d_all + L.d_off, h->Dt,
                                          L.noise_used, L.noise_bstride, L.noise_strength, L.bias, LA_ACT_LRELU, 0.2f, sq2,
                                          h->clamp, L.y, h->cws, h->cws_bytes, B, L.cin, L.cout, res, stream, fwd_row(ci),
                                          (fuse_rgb && q == nl - 1) ? &rf : nullptr, fwd_row(ci + 1), fwd_mult(ci + 1), wlo[ci], whi[ci],
                                          ci == h->nconv - 1 ? c1lo : 0, ci == h->nconv - 1 ? c1hi : 0);
            } else {
                rc = la_modconv3x3_up2_fwd_ex(x, x_bstride, L.wf, L.wqf, h->precision, h->s_all + L.s_off, h->S, h->d_all + L.d_off, h->Dt,
                                              L.noise_used, L.noise_bstride, L.noise_strength, L.bias, LA_ACT_LRELU, 0.2f,
                                              sq2, h->clamp, h->fir, h->zT, L.y, nullptr, h->cws, h->cws_bytes, B, L.cin,
                                              L.cout, res, stream, fwd_row(ci),
                                              zt_dense ? 0 : 2 * zt_xhalf(res), zt_dense ? 0 : zt_xhalf(res), fwd_row(ci + 1), fwd_mult(ci + 1),
                                              wlo[ci], whi[ci], ci == h->nconv - 2 ? c0lo : 0, ci == h->nconv - 2 ? c0hi : 0);
            }
            if (rc) return rc;
            x = L.y; x_bstride = (long)L.cout * res * res;
        }
        if (!fuse_rgb) {
            // img = upsample2d(img_prev, f) + torgb(x): up 2, pad (2,1,2,1), gain 4 (upfirdn2d.py:342-348) -- the up-sampled image of the block
            // below is computed inside the ToRGB kernel (la_up2_quad; round 4: a launch of its own per block, parked in g_img)
            const float* skip_lo = k > 0 ? h->rgb[k - 1].img : nullptr;
            const bool iw = ihi[k] > 0 && (long)res * res > 4096;      // (windowed block: only the image rows somebody reads)
            if ((rc = la_torgb_forward(x, T.weight, h->s_all + T.s_off, h->S, T.bias, nullptr, T.rgb_pre, rgb_dst, B, T.cin, h->imgc,
                                       res, res, h->clamp, stream, nullptr, iw ? ilo[k] : 0, iw ? ihi[k] : 0, skip_lo, h->fir)))
                return rc;
        }
        if (k == h->nblocks - 1) h->final_img = rgb_dst;
    }
    return LA_OK;
}

extern "C" int la_synth_backward(la_synth* h, const float* g_img, float* dws, hipStream_t stream) {
    LA_CHECK_ARG(h && g_img && dws, "synth_backward: null pointer");
    LA_CHECK_ARG(h->lastB >= 1, "synth_backward: no forward pass to differentiate");
    const int B = h->lastB;
    int rc;
    const float* gi = g_img;   // gradient w.r.t. the image at the current resolution
    int ci = h->nconv - 1;
    const float* gx_next = nullptr;   // gradient w.r.t. this block's conv1 output coming from the block above
    // every layer leaves its style-gradient partials in buffers of its own; ONE finish (3 launches) at the end of the pass turns
    // them into ds_all (la_style_backward_all), instead of 3 small launches per layer
    LaStyleFinish fin; memset(&fin, 0, sizeof(fin));
    fin.imgc = h->imgc; fin.d_stride = h->Dt; fin.s_stride = h->S; fin.ds_stride = h->S;
    auto fin_conv = [&](const ConvLayer& L, int ntiles, int nslabs) {
        LaStyleFinish::Conv& c = fin.conv[fin.nconv++];
        c.ds_part = L.dsp; c.ddn_part = L.ddnp; c.d = h->d_all + L.d_off; c.s = h->s_all + L.s_off; c.wsq = L.wsq;
        c.ds_out = h->ds_all + L.s_off; c.ntiles = ntiles; c.nslabs = nslabs; c.cin = L.cin; c.cout = L.cout;
    };
    static const bool no_fuse = la_dev_env("LA_NO_SEAM_FUSE") != nullptr;      // dev knobs: A/B of the fused seams on one box
    static const bool no_fuse2 = la_dev_env("LA_NO_SEAM2_FUSE") != nullptr;
    const bool f16 = h->precision == LA_PREC_F32 ? false : h->precision == LA_PREC_F16X2;
    // fp16 operand scales of the backward contractions: the kernels whose epilogues produce a contraction's input (direct contraction
    // kernels, split-K finish pass, seam kernel) lower the consumer's slot row themselves (la_xs_lower; la_synth_forward reset the
    // rows): no plane maxima, no reduction launch between producer and consumer.  Dev knob LA_NO_XS_HANDOFF: plane maxima +
    // la_xscale_pmax at the consumer.
    static const bool no_xs = la_dev_env("LA_NO_XS_HANDOFF") != nullptr;
    const bool xs_hand = f16 && !no_xs;
    const float up_mult = la_modconv_up2_bwd_xs_mult(h->fir);
    auto xs_slot = [&](int conv_index) { return xs_hand ? h->xs_bwd + (long)conv_index * B * LA_XS_FAN : nullptr; };
    bool seam2_done = false;      // this block's conv1 seam was already applied by the epilogue of the up layer's backward above it
    bool pyramid_done = false;    // the image-gradient levels below the current block exist already (la_image_grad_pyramid)
    // Row windows (la_synth_set_row_window): the gradient of a conv output is non-zero only inside the rows its forward window covers (the
    // window IS the cone of the image window), so the backward pass of a windowed forward pass reads, computes and writes those rows only:
    // a producer writes its window, its consumer reads everything outside it as zeros (LaBwdRows) -- the ping-pong buffers G0 / G1 hold
    // older contents there.  Default path only (fused seams, slot rows); needs an image gradient that is zero outside the image window.
    const bool bw = xs_hand && !no_fuse && !no_fuse2 && h->precision != LA_PREC_F32;
    auto r4 = [](int lo, int hi, int res, int* o_lo, int* o_hi) { *o_lo = lo & ~3; *o_hi = (hi + 3) & ~3; if (*o_hi > res) *o_hi = res; };
    for (int k = h->nblocks - 1; k >= 0; --k) {
        const int res = 4 << k;
        const long HW = (long)res * res;
        // W1 = window of this block's conv1 output (= valid rows of the gradient G0 that reaches it), R1 = the 4-row tiles around the
        // window of conv0's output (what conv1's backward contraction writes into G1), Wb = window of the block below's conv1 output
        const bool win = bw && k > 0 && h->fw_hi[ci] > 0;
        LaBwdRows rw1{0, 0, 0, 0, 0, 0, 0, 0}, rw0{0, 0, 0, 0, 0, 0, 0, 0};
        if (win && k == h->nblocks - 1 && h->fw_c1 > 0) {      // top block: conv1's backward contraction writes the window's tile columns, the FIR adjoint reads the others as zeros
            rw1.out_c0 = h->fw_c0; rw1.out_c1 = h->fw_c1; rw0.in_c0 = h->fw_c0; rw0.in_c1 = h->fw_c1;
        }
        if (win) {
            rw1.in_lo = h->fw_lo[ci]; rw1.in_hi = h->fw_hi[ci];
            r4(h->fw_lo[ci - 1], h->fw_hi[ci - 1], res, &rw1.out_lo, &rw1.out_hi);
            rw0.in_lo = rw1.out_lo; rw0.in_hi = rw1.out_hi;
            if (ci - 2 >= 0 && h->fw_hi[ci - 2] > 0) { rw0.out_lo = h->fw_lo[ci - 2]; rw0.out_hi = h->fw_hi[ci - 2]; }
        }
        RgbLayer& T = h->rgb[k];
        ConvLayer& L1 = h->conv[ci];
        const int slabs = la_seam_slabs(HW);
        const int tiles1 = la_modconv_ds_tiles(res);
        const int nseg1 = seam2_done ? tiles1 : slabs;      // granularity of this block's ddn / dweff / plane-maxima partials
        // ---- seam at conv1 output: ToRGB backward + act backward (top block, or when it was not fused above)
        LaSeamArgs s;
        if (!seam2_done) {
            memset(&s, 0, sizeof(s));
            s.y = L1.y; s.gx_next = gx_next; s.gz = h->G0; s.HW = HW; s.C = L1.cout;
            s.demod = h->d_all + L1.d_off; s.demod_stride = h->Dt; s.bias = L1.bias;
            s.noise = L1.noise_used; s.noise_bstride = L1.noise_bstride; s.noise_strength = L1.noise_strength;
            s.act = LA_ACT_LRELU; s.alpha = 0.2f; s.gain = sqrtf(2.f); s.clamp = h->clamp;
            s.ddn_part = L1.ddnp;
            s.g_img = gi; s.rgb_pre = T.rgb_pre; s.rgb_clamp = h->clamp; s.wrgb = T.weight;
            s.s_rgb = h->s_all + T.s_off; s.s_stride = h->S; s.dweff_part = T.dwep;
            if (xs_hand) { s.xs_out = xs_slot(ci); s.xs_mult = 1.f; }      // the seam kernel lowers the slot row of the contraction that follows ...
            else if (f16) s.pmax_out = h->pmax;                              // ... or leaves the plane maxima of gz for it
            if (win) { s.p_lo = (long)rw1.in_lo * res; s.p_hi = (long)rw1.in_hi * res; }
            if ((rc = la_seam_backward(s, B, h->imgc, stream))) return rc;
        }
        {
            LaStyleFinish::Rgb& r = fin.rgb[fin.nrgb++];
            r.dweff_part = T.dwep; r.wrgb = T.weight; r.ds_out = h->ds_all + T.s_off; r.nslabs = nseg1; r.C = T.cin;
        }
        // ---- conv1 backward-data (+ style-gradient partials).  16-bit modes, blocks above the first: the seam of the up-sampling
        // layer L0 (whose saved output is this contraction's xin) is applied in the same epilogue -- no separate pass over y0 and
        // the gradient; its demod-gradient partials and plane maxima come out per pixel tile.
        const bool fuse_seam = k > 0 && h->precision != LA_PREC_F32 && !no_fuse;
        {
            const float* xin = (k == 0) ? h->cst : h->conv[ci - 1].y;
            const long xin_bs = (k == 0) ? 0 : (long)L1.cin * HW;
            LaSeamFuse sf; memset(&sf, 0, sizeof(sf));
            if (fuse_seam) {
                const ConvLayer& L0f = h->conv[ci - 1];
                sf.demod = h->d_all + L0f.d_off; sf.demod_stride = h->Dt; sf.bias = L0f.bias;
                sf.noise = L0f.noise_used; sf.noise_bstride = L0f.noise_bstride; sf.noise_strength = L0f.noise_strength;
                sf.act = LA_ACT_LRELU; sf.alpha = 0.2f; sf.gain = sqrtf(2.f); sf.clamp = h->clamp;
                sf.ddn_part = h->conv[ci - 1].ddnp;
                sf.pmax = f16 ? h->pmax2 : nullptr;
                if (xs_hand) { sf.xs_out = xs_slot(ci - 1); sf.xs_mult = up_mult; }
            }
            if ((rc = la_modconv3x3_bwd_ex(h->G0, f16 ? (seam2_done ? h->pmax3 : h->pmax) : nullptr, nseg1, L1.wb, L1.wqb, h->precision, h->s_all + L1.s_off,
                                           h->S, xin, xin_bs, h->G1, L1.dsp, h->cws, h->cws_bytes, B, L1.cin, L1.cout, res, stream,
                                           fuse_seam ? &sf : nullptr, xs_slot(ci), win ? &rw1 : nullptr)))
                return rc;
            fin_conv(L1, tiles1, nseg1);
        }
        --ci;
        if (k == 0) break;
        // ---- conv0 (up-sampling layer): act backward (unless fused above) -> FIR adjoint -> stride-2 backward-data
        ConvLayer& L0 = h->conv[ci];
        if (!fuse_seam) {
            memset(&s, 0, sizeof(s));
            s.y = L0.y; s.gx_next = h->G1; s.gz = h->G1; s.HW = HW; s.C = L0.cout;
            s.demod = h->d_all + L0.d_off; s.demod_stride = h->Dt; s.bias = L0.bias;
            s.noise = L0.noise_used; s.noise_bstride = L0.noise_bstride; s.noise_strength = L0.noise_strength;
            s.act = LA_ACT_LRELU; s.alpha = 0.2f; s.gain = sqrtf(2.f); s.clamp = h->clamp;
            s.ddn_part = L0.ddnp;
            if (xs_hand) { s.xs_out = xs_slot(ci); s.xs_mult = up_mult; }
            else if (f16) s.pmax_out = h->pmax;      // plane maxima of gz: bound for the operand scale of the fused FIR-adjoint + split pass
            if ((rc = la_seam_backward(s, B, 0, stream))) return rc;
        }
        // ---- image gradient one level down: adjoint of upsample2d = FIR (flipped) + decimate 2, pad (1,1,1,1), gain 4
        // (before the up layer's backward contraction: its epilogue may apply the block below's ToRGB backward)
        RgbLayer& P = h->rgb[k - 1];
        if (!pyramid_done) {
            if ((rc = la_upfirdn2d_ex(gi, P.g_img, B, h->imgc, res, res, h->fir, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1, 1, 4.f, nullptr,
                                      stream)))
                return rc;
            // every level below in ONE launch (la_image_grad_pyramid: the levels depend on the image gradient only), from <= 256^2 inputs
            if (k >= 2 && res / 2 <= 256) {
                float* outs[MAX_BLOCKS];
                for (int q = k - 2; q >= 0; --q) outs[k - 2 - q] = h->rgb[q].g_img;
                if ((rc = la_image_grad_pyramid(P.g_img, outs, k - 1, B * h->imgc, res / 2, h->fir, stream))) return rc;
                pyramid_done = true;
            }
        }
        const bool fuse_seam2 = h->precision != LA_PREC_F32 && !no_fuse2;
        {
            const int hin = res / 2;
            const int tiles = la_modconv_ds_tiles(hin);
            const int nseg0 = fuse_seam ? tiles1 : slabs;
            LaSeamFuse sf2; memset(&sf2, 0, sizeof(sf2));
            if (fuse_seam2) {
                // the conv1 seam of the block BELOW (its saved output is this contraction's xin), incl. its ToRGB backward
                const ConvLayer& Lb = h->conv[ci - 1];
                sf2.demod = h->d_all + Lb.d_off; sf2.demod_stride = h->Dt; sf2.bias = Lb.bias;
                sf2.noise = Lb.noise_used; sf2.noise_bstride = Lb.noise_bstride; sf2.noise_strength = Lb.noise_strength;
                sf2.act = LA_ACT_LRELU; sf2.alpha = 0.2f; sf2.gain = sqrtf(2.f); sf2.clamp = h->clamp;
                sf2.ddn_part = Lb.ddnp;
                sf2.pmax = f16 ? h->pmax3 : nullptr;
                if (xs_hand) { sf2.xs_out = xs_slot(ci - 1); sf2.xs_mult = 1.f; }
                sf2.imgc = h->imgc; sf2.g_img = P.g_img; sf2.rgb_pre = P.rgb_pre; sf2.rgb_clamp = h->clamp;
                sf2.wrgb = P.weight; sf2.s_rgb = h->s_all + P.s_off; sf2.s_rgb_stride = h->S; sf2.dweff_part = P.dwep;
            }
            if ((rc = la_modconv3x3_up2_bwd_ex(h->G1, f16 ? (fuse_seam ? h->pmax2 : h->pmax) : nullptr, nseg0, L0.wb, L0.wqb, h->precision, h->s_all + L0.s_off, h->S,
                                               h->conv[ci - 1].y, (long)L0.cin * hin * hin, h->fir, h->zT, h->G0, L0.dsp, h->cws, h->cws_bytes, B,
                                               L0.cin, L0.cout, res, stream, fuse_seam2 ? &sf2 : nullptr, xs_slot(ci), win ? &rw0 : nullptr)))
                return rc;
            fin_conv(L0, tiles, nseg0);
        }
        --ci;
        gx_next = h->G0;
        gi = P.g_img;
        seam2_done = fuse_seam2;
    }
    if ((rc = la_style_backward_all(fin, B, stream))) return rc;
    return la_affine_backward(h->st, h->ds_all, B, h->wdim, dws, h->num_ws, h->aff_part, stream);
}
