/* latentaug_hip.h -- C ABI of liblatentaug_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the latent-optimisation hot path of ltronchin/LatentAugment.  Every entry point names the
 * reference interface it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is DEVICE memory (fp32, contiguous NCHW) unless its name ends
 *     in `_host`; the caller has selected the device and owns every buffer; nothing here allocates device memory.
 *   - all work is enqueued on `stream` (a hipStream_t, passed as void* from foreign code); no host<->device sync.
 *   - return 0 on success, negative on failure (LA_ERR_*); la_last_error() gives the thread's last message.
 *     This replaces the TORCH_CHECKs of the reference bindings (bias_act.cpp:35-51, upfirdn2d.cpp:19-40).
 *   - re-entrant: the only global mutable state is the thread-local error string (cf. bias_act.cpp:54,88), a once-per-device
 *     "kernel attribute set" flag (atomic) and the opt-in launch profiler la_prof_* (off by default; ONE process-wide instance
 *     that is not thread-safe: measurement runs only).  No kernel-variant switches, no environment variables (see la_prof_*).
 */
#ifndef LATENTAUG_HIP_H
#define LATENTAUG_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef LA_STREAM_T
#define LA_STREAM_T
typedef struct ihipStream_t* la_stream_t; /* == hipStream_t */
#endif

#define LA_OK 0
#define LA_ERR_ARG (-1)
#define LA_ERR_HIP (-2)
#define LA_ERR_WORKSPACE (-3)

/* activation ids = the reference's cuda_idx (torch_utils/ops/bias_act.py:20-30) */
#define LA_ACT_LINEAR 1
#define LA_ACT_RELU 2
#define LA_ACT_LRELU 3

const char* la_last_error(void);
int la_abi_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * L0 ops -- replace the pybind plugins of models/stylegan3/torch_utils/ops/
 * ------------------------------------------------------------------------------------------------------------- */

/* bias_act forward.  Replaces bias_act_plugin.bias_act(x,b,xref,yref,dy,grad=0,dim,act,alpha,gain,clamp)
 * (bias_act.cpp:32-90, kernel bias_act.cu:23-147).  Element i takes b[(i / stepb) % nb]; for NCHW and dim=1:
 * stepb = H*W, nb = C.  b may be NULL.  clamp < 0 disables clamping. */
int la_bias_act_f32(const float* x, const float* b, float* y, long n, long stepb, int nb, int act, float alpha,
                    float gain, float clamp, la_stream_t stream);

/* bias_act first-order backward (grad=1 of the same plugin; python side bias_act.py:155-177):
 * dx = dy * act'(.) taken from the saved OUTPUT yref (sign for lrelu, zero where |yref| >= clamp), db[c] = sum dx. */
int la_bias_act_grad_f32(const float* dy, const float* yref, float* dx, float* db, long n, long stepb, int nb, int act,
                         float alpha, float gain, float clamp, la_stream_t stream);

/* The general form of the same plugin entry point (bias_act.cpp:32 `bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)`):
 * every activation of bias_act.py:20-30 (act = its cuda_idx 1..9: linear relu lrelu tanh sigmoid elu selu softplus swish) and grad = 0
 * (forward), 1 (x = incoming gradient: x * f' * gain * dy) or 2 (x = gradient of the gradient: x * f'' * gain * dy); f', f'' are formed
 * from yref / gain (swish: from xref + b), results are zero where |yref| >= clamp.  b / xref / yref / dy may be NULL ("absent", the
 * reference's empty tensors); dim enters as (stepb, nb): element i belongs to bias entry (i / stepb) % nb. */
int la_bias_act_ex_f32(const float* x, const float* b, const float* xref, const float* yref, const float* dy, float* out, long n,
                       long stepb, int nb, int grad, int act, float alpha, float gain, float clamp, la_stream_t stream);
/* db [nb] = sum of dx over every axis but the bias axis (what bias_act.py:187,206 forms with Tensor.sum): element i -> (i / stepb) % nb. */
int la_bias_sum_f32(const float* dx, float* db, long n, long stepb, int nb, la_stream_t stream);

/* upfirdn2d.  Replaces upfirdn2d_plugin.upfirdn2d(x,f,upx,upy,downx,downy,padx0,padx1,pady0,pady1,flip,gain)
 * (upfirdn2d.cpp:16-98, kernels upfirdn2d.cu:29-200).  f_host: fh*fw taps in HOST memory (<= 8x8), as produced by
 * setup_filter (upfirdn2d.py:70-114).  Output size per axis: la_upfirdn2d_out_size (upfirdn2d.cpp:35-36).
 * The backward of the op is the same op with up<->down swapped, flip negated and the pads of upfirdn2d.py:255-266. */
int la_upfirdn2d_out_size(int in_size, int up, int down, int pad0, int pad1, int taps);
int la_upfirdn2d_f32(const float* x, const float* f_host, float* y, int N, int C, int H, int W, int fh, int fw, int upx,
                     int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                     la_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Modulated 3x3 convolution of a SynthesisLayer (the SG2 `modulated_conv2d` + `bias_act` pair that the reference
 * reaches through G.synthesis, util_latent_aug.py:227; resampling algebra conv2d_resample.py:82-134).
 * Non-fused formulation: y = act((W * (x.s)) . d + noise + bias); never materialises per-sample weights.
 * ------------------------------------------------------------------------------------------------------------- */

/* W[cout][cin][ktaps] -> wf[t][cin][cout] (forward A-operand), wb[t][cout][cin] (backward), wsq[cout][cin] = sum_t W^2.
 * Any of wf/wb/wsq may be NULL. */
int la_pack_conv_weights_f32(const float* w, float* wf, float* wb, float* wsq, int cout, int cin, int ktaps,
                             la_stream_t stream);

/* same-resolution layer (conv1): x [B][cin][res][res] (x_bstride = 0 broadcasts one sample), s [B][s_stride] styles,
 * d [B][d_stride] demodulation coefficients (NULL = no demod), noise [res][res] (noise_bstride 0) or per sample. */
int la_modconv3x3_fwd_f32(const float* x, long x_bstride, const float* wf, const void* wq, int precision, const float* s, int s_stride, const float* d,
                          int d_stride, const float* noise, long noise_bstride, float noise_strength, const float* bias,
                          int act, float alpha, float gain, float clamp, float* y, void* ws, size_t ws_bytes, int B, int cin, int cout, int res,
                          la_stream_t stream);

/* up-sampling layer (conv0): x [B][cin][res/2][res/2] -> y [B][cout][res][res];
 * scratch: B*cout*(res+1)^2 floats (the transposed-conv intermediate of conv2d_resample.py:125). */
int la_modconv3x3_up2_fwd_f32(const float* x, long x_bstride, const float* wf, const void* wq, int precision, const float* s, int s_stride,
                              const float* d, int d_stride, const float* noise, long noise_bstride, float noise_strength,
                              const float* bias, int act, float alpha, float gain, float clamp, const float* fir_host,
                              float* scratch, float* y, void* ws, size_t ws_bytes, int B, int cin, int cout, int res, la_stream_t stream);

/* backward-data + style-gradient partials.  gz [B][cout][res][res] = gradient w.r.t. the raw contraction (already
 * multiplied by d and by act').  gx = (W^T * gz) . s ;  ds_part[b][i][tile] = partial sums of sum_p (W^T*gz) . xin. */
int la_modconv3x3_bwd_f32(const float* gz, const float* wb, const void* wq, int precision, const float* s, int s_stride, const float* xin,
                          long xin_bstride, float* gx, float* ds_part, void* ws, size_t ws_bytes, int B, int cin, int cout, int res,
                          la_stream_t stream);
int la_modconv3x3_up2_bwd_f32(const float* gz, const float* wb, const void* wq, int precision, const float* s, int s_stride, const float* xin,
                              long xin_bstride, const float* fir_host, float* scratch, float* gx, float* ds_part, void* ws, size_t ws_bytes, int B,
                              int cin, int cout, int res, la_stream_t stream);
int la_modconv_ds_tiles(int grid_res); /* leading dimension of ds_part for a backward over a grid_res^2 grid */
/* ws (may be NULL for precision 0): scratch of la_modconv_workspace_bytes() bytes.  With it, layers of <= 34x34 split
 * their K loop over workgroups (deterministic slice sum) instead of serialising it on a few CUs; the split-bf16
 * precisions also park the pre-split copy of the launch input there. */
size_t la_modconv_workspace_bytes(int B, int cin, int cout, int res, int up);
/* Contraction precision (the `precision` argument of the la_modconv3x3_* calls; `wq` = weights packed for it or NULL):
 *   0 LA_PREC_F32     exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *   1 LA_PREC_BF16X3  fp32 operands split into 3 bf16 terms, 6 bf16 MFMAs per product, fp32 accumulate (fp32-class error)
 *   2 LA_PREC_BF16X2  2 bf16 terms, 3 bf16 MFMAs (~4e-6 relative error per layer)
 *   3 LA_PREC_F16X2   operands scaled by per-sample / per-layer powers of two and split into 2 fp16 terms, 3 fp16 MFMAs per
 *                     product, fp32 accumulate (fp32-class error: 2 x 11 mantissa bits)
 * wq is produced by la_pack_conv_weights_bf16_f32 (transpose = 0 for the forward calls, 1 for the backward calls). */
#define LA_PREC_F32 0
#define LA_PREC_BF16X3 1
#define LA_PREC_BF16X2 2
#define LA_PREC_F16X2 3
size_t la_modconv_bf16_pack_bytes(int cin, int cout, int transpose, int nterm);
int la_pack_conv_weights_bf16_f32(const float* w, void* out, int cout, int cin, int ktaps, int transpose, int nterm,
                                  la_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Criteria and optimiser
 * ------------------------------------------------------------------------------------------------------------- */

/* l2_loss_vectorized (augments/utils/util_latent_aug.py:315-361) on flattened rows: X [n][K], Y [m][K] ->
 * D [m][n] = |Y_m|^2 + |X_n|^2 - 2<Y_m,X_n> (compute_mean=False); mean_out (may be NULL) = sum(D)/(m*n)/K.
 * workspace: la_pairwise_l2_workspace_floats(n, m) floats (K-slice partials of the bank scan). */
long la_pairwise_l2_workspace_floats(int n, long m);
int la_pairwise_l2_f32(const float* X, int n, const float* Y, long m, long K, float* D, float* mean_out,
                       float* workspace, la_stream_t stream);

/* get_center_crop (augments/utils/util_dataset.py:317-323) on [planes][R][R] -> [planes][cc][cc]. */
int la_center_crop_f32(const float* src, float* dst, long planes, int R, int cc, int off, la_stream_t stream);

/* torch.optim.Adam step as used at util_latent_aug.py:213,274-276 (step is 1-based). */
int la_adam_step_f32(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1, float beta2,
                     float eps, la_stream_t stream);

/* Unit normals for the explicit per-layer noise tensors of noise_mode='random' (the reference's SynthesisLayer draws torch.randn inside
 * G.synthesis, call site augments/utils/util_latent_aug.py:308): out [rows][row_elems]; element e of GLOBAL sample row row0 + r of
 * `layer` is a pure function of (seed, layer, row0 + r, e) -- Philox4x32-10 + Box-Muller -- so a rank generates only the rows of its
 * shard and the gathered batch does not depend on the sharding. */
int la_noise_normal_f32(float* out, long rows, long row_elems, unsigned long long seed, unsigned layer, long row0, la_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Synthesis network engine: replaces  G.synthesis(ws, noise_mode=...)  (call sites util_latent_aug.py:227,488) and the
 * autograd backward to ws that loss.backward() (:275) runs through it.  Architecture 'skip', fp32
 * (models/stylegan3/legacy.py:122-144).
 *
 * params: flat list of device tensors in execution order (names as in legacy.py:171-203):
 *   b4:            const | conv1.{affine.weight, affine.bias, weight, bias, noise_const} | torgb.{affine.weight, affine.bias, weight, bias}
 *   b8 .. bR each: conv0.{5 tensors as above} | conv1.{5} | torgb.{4}
 * noise_strength_host: one float per SynthesisLayer in the same order.  channels[k] = channels at resolution 4<<k.
 * fir_host: the 4x4 resample filter (setup_filter([1,3,3,1])).  workspace: la_synth_workspace_bytes() of device memory,
 * owned by the caller and alive as long as the handle.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct la_synth la_synth;
int la_synth_num_ws(int img_resolution);
int la_synth_num_params(int img_resolution);
size_t la_synth_workspace_bytes(int img_resolution, int img_channels, int w_dim, const int* channels, int max_batch);
int la_synth_create(int img_resolution, int img_channels, int w_dim, const int* channels, float conv_clamp,
                    const float* const* params, int nparams, const float* noise_strength_host, int nlayers,
                    const float* fir_host, int fir_h, int fir_w, int max_batch, void* workspace, size_t workspace_bytes,
                    la_stream_t stream, la_synth** out);
void la_synth_destroy(la_synth* h);
/* contraction precision of every modulated conv of the engine (LA_PREC_*, default LA_PREC_F32) */
int la_synth_set_precision(la_synth* h, int precision);
/* Kept for ABI compatibility, no effect since round 4 (rounds 2-3: 0 = fp16 operand scale of the forward contractions from the a-priori
 * bound conv_clamp * max|style|, 1 = from data maxima).  Every fp16 operand scale is now derived from the data of each pass by the
 * kernel that produces the tensor (slot rows lowered with atomicMin, csrc/la_common.h): no bound, no calibration, nothing to select. */
int la_synth_set_operand_scale(la_synth* h, int from_data);
/* Row window of the image for the forward passes that follow (rows [row_lo, row_hi) of img_resolution; 0, 0 = whole frames): the 16-bit
 * forward kernels of the blocks at >= 64^2 compute the rows that window depends on -- 3x3 / FIR taps and the up-sampling geometry of
 * SynthesisBlock / SynthesisLayer geometry (conv2d_resample.py:112-134, upfirdn2d.py:342-348) followed down the blocks -- and leave the other rows of every buffer as they were.  la_synth_backward is
 * unchanged and expects an image gradient that is zero outside the window. */
int la_synth_set_row_window(la_synth* h, int row_lo, int row_hi);
/* Column window on top of the row window (0, 0 = all): the top block's conv1, the FIR in front of it and the FIR adjoint behind it follow it in
 * whole 32-column tiles; every other kernel computes whole rows. */
int la_synth_set_col_window(la_synth* h, int col_lo, int col_hi);
int la_synth_get_precision(const la_synth* h);
/* ws element (b,l,j) = ws[b*ws_bstride + l*ws_lstride + j] (ws_lstride = 0: W space, one w per sample).
 * noise_mode 0 'none', 1 'const', 2 explicit unit-variance tensors noises[layer] [B][res][res] ('random' drawn by the caller).
 * img_out NULL: the image stays in the engine (la_synth_image). */
int la_synth_forward(la_synth* h, const float* ws, long ws_bstride, long ws_lstride, int B, int noise_mode,
                     const float* const* noises, float* img_out, la_stream_t stream);
/* d(loss)/d(ws) [B][num_ws][w_dim] from d(loss)/d(img) [B][C][R][R]; differentiates the last la_synth_forward. */
int la_synth_backward(la_synth* h, const float* g_img, float* dws, la_stream_t stream);
const float* la_synth_image(const la_synth* h);
const float* la_synth_block_image(const la_synth* h, int block);
const float* la_synth_layer_output(const la_synth* h, int layer);
const float* la_synth_styles(const la_synth* h);
const float* la_synth_style_grads(const la_synth* h);
int la_synth_style_rows(const la_synth* h);

/* ---------------------------------------------------------------------------------------------------------------
 * Mapping network (rand_aug mode): replaces G.mapping(z, c=None, truncation_psi=...) at util_latent_aug.py:203,460.
 * weights[i] = mapping.fc{i}.weight [w_dim][in], biases[i] = mapping.fc{i}.bias (legacy.py:175-176), lr_mul 0.01
 * (legacy.py:141), w_avg = mapping.w_avg (legacy.py:172).  tmp: 2*B*max(z_dim,w_dim) floats.
 * la_fc_f32: FullyConnectedLayer forward y = act(x @ (W*lr_mul/sqrt(in))^T + b*lr_mul) * gain.
 * ------------------------------------------------------------------------------------------------------------- */
int la_fc_f32(const float* x, const float* W, const float* bias, float* y, int B, int in, int out, float lr_mul, int act,
              float alpha, float gain, la_stream_t stream);
int la_mapping_forward_f32(const float* z, int B, int z_dim, int w_dim, int num_layers, const float* const* weights,
                           const float* const* biases, float lr_mul, const float* w_avg, float truncation_psi,
                           int num_ws, float* tmp, float* ws_out, la_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Discriminator engine: replaces  D(x, c=None)  and its backward to x inside calc_loss_disc (util_latent_aug.py:363-371).
 * Architecture 'resnet' + MinibatchStd epilogue (legacy.py:220-247).  params: device tensors in this order (names of
 * legacy.py:271-288), resolution R first:
 *   bR: fromrgb.weight, fromrgb.bias, conv0.weight, conv0.bias, conv1.weight, conv1.bias, skip.weight
 *   b(R/2) .. b8: conv0.weight, conv0.bias, conv1.weight, conv1.bias, skip.weight
 *   b4: conv.weight, conv.bias, fc.weight, fc.bias, out.weight, out.bias
 * channels[k] = channels at resolution 4 << k (same table as the generator).  mbstd_group_size: 4 in every SG2 config;
 * the batch of a forward must be divisible by min(group, batch) exactly as in the reference.
 * la_disc_loss: loss_out[0] = softplus(-logits).mean() * w_disc and keeps d(loss)/d(logits) for la_disc_backward.
 * la_disc_backward: g_img [B][C][R][R] (accumulate != 0: added to what g_img holds).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct la_disc la_disc;
int la_disc_num_params(int img_resolution);
size_t la_disc_workspace_bytes(int img_resolution, int img_channels, const int* channels, int max_batch);
int la_disc_create(int img_resolution, int img_channels, const int* channels, float conv_clamp, const float* const* params,
                   int nparams, const float* fir_host, int mbstd_group_size, int max_batch, void* workspace,
                   size_t workspace_bytes, la_stream_t stream, la_disc** out);
void la_disc_destroy(la_disc* h);
int la_disc_set_precision(la_disc* h, int precision);
int la_disc_forward(la_disc* h, const float* img, int B, la_stream_t stream);
int la_disc_loss(la_disc* h, float w_disc, int norm_batch, float* loss_out, la_stream_t stream);
int la_disc_backward(la_disc* h, const float* dlogits, float* g_img, int accumulate, la_stream_t stream);
const float* la_disc_logits(const la_disc* h);

/* ---------------------------------------------------------------------------------------------------------------
 * Perceptual feature engine: replaces  self.vgg16(x, resize_images=False, return_lpips=True)  and its backward inside
 * calc_loss_lpips_torchscript (util_latent_aug.py:387-409).  The network is described by the caller as a list of ops
 * (VGG16 = 13 x conv3x3+ReLU, 4 x max-pool, 5 taps); a tap emits f * rsqrt(sum_c f^2 + 1e-10) * sqrt(lin[c]) / sqrt(H*W),
 * so squared L2 between two outputs is their LPIPS distance.  params: per op in order -- conv: weight [cout][cin][3][3],
 * bias [cout]; tap: lin [C]; pools: none.  la_crop_repeat_f32: the crop + `.repeat([1,3,1,1])` of :394 for every modality
 * (rows ordered modality-major: row = c*B + b) with an affine preprocess; la_crop_repeat_grad_f32: its adjoint, ADDED to
 * g_img.
 * ------------------------------------------------------------------------------------------------------------- */
#define LA_FEAT_CONV_RELU 0
#define LA_FEAT_TAP 1
#define LA_FEAT_MAXPOOL2 2
#define LA_FEAT_AVGPOOL2 3
typedef struct la_feat_op { int kind, cin, cout; } la_feat_op;
typedef struct la_feat la_feat;
size_t la_feat_workspace_bytes(int nops, const la_feat_op* ops, int in_ch, int in_res, int max_batch);
int la_feat_create(int nops, const la_feat_op* ops, const float* const* params, int nparams, int in_ch, int in_res,
                   int max_batch, void* workspace, size_t workspace_bytes, la_stream_t stream, la_feat** out);
void la_feat_destroy(la_feat* h);
int la_feat_num_features(const la_feat* h);
int la_feat_set_precision(la_feat* h, int precision);
int la_feat_forward(la_feat* h, const float* x, int N, float* feat_out, la_stream_t stream);
int la_feat_backward(la_feat* h, const float* gfeat, float* gx, la_stream_t stream);
int la_crop_repeat_f32(const float* img, float* xc, int B, int imgc, int R, int S, int y0, int x0, int rep, float scale,
                       float shift, la_stream_t stream);
int la_crop_repeat_grad_f32(const float* gxc, float* g_img, int B, int imgc, int R, int S, int y0, int x0, int rep,
                            float scale, la_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The loop: replaces LatentAug.forward(w, fname) (augments/utils/util_latent_aug.py:207-310) for 3-D w input.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct la_opt_config {
    int steps;              /* opt_num_epochs (latent_aug.py:81) */
    float lr;               /* opt_lr (latent_aug.py:82) */
    float beta1, beta2, eps;
    float w_latent, w_pix, w_disc, w_lpips; /* latent_aug.py:88-91 */
    int criterion_mode;     /* 0 | 1, kept for ABI stability: both use bank column sums reduced once per handle for the gradient;
                               loss scalars (only when losses_out is given) always use the reference's GEMM form over the banks */
    int soft_aug;           /* latent_aug.py:94 */
    float alpha;            /* latent_aug.py:95 */
    int loop_noise_mode;    /* 1 = 'const' (util_latent_aug.py:227) */
    int final_noise_mode;   /* util_latent_aug.py:488 uses the generator default ('random'): pass 2 + tensors */
    int norm_batch;         /* n of the criteria's 1/(m*n); 0 = local batch (what a DataParallel replica sees) */
    int crop, crop_off;     /* util_dataset.py:317-323: int(sqrt(R*R/2)), round((R-crop)/2) */
} la_opt_config;
typedef struct la_latent_opt la_latent_opt;
size_t la_latent_opt_workspace_bytes(int img_resolution, int img_channels, int w_dim, const la_opt_config* cfg, long Mw,
                                     long Mx, int max_batch);
/* bankW [Mw][num_ws][w_dim] (register_buffer 'W', :148); bankXc [C][Mx][crop*crop] = centre-cropped 'X' (:158, :253),
 * modality-major. */
int la_latent_opt_create(la_synth* g, int img_resolution, int img_channels, int w_dim, const la_opt_config* cfg,
                         const float* bankW, long Mw, const float* bankXc, long Mx, int max_batch, void* workspace,
                         size_t workspace_bytes, la_latent_opt** out);
void la_latent_opt_destroy(la_latent_opt* h);
/* attach the discriminator used when cfg.w_disc != 0 (must outlive the loop handle) */
int la_latent_opt_set_disc(la_latent_opt* h, la_disc* d);
/* attach the LPIPS criterion used when cfg.w_lpips != 0: feature engine, real-feature banks [C][Mf][F] (register_buffer
 * 'fea_<mode>', util_latent_aug.py:171; modality-major), crop size S (crop_size_aug), input preprocess x*scale+shift.
 * la_latent_opt_set_crop_pos: absolute (x, y) of the S x S window, drawn by the host once per forward
 * (util_dataset.py:284-296 + the centre-crop offset). */
size_t la_latent_opt_lpips_workspace_bytes(int img_channels, int F, int S, long Mf, int max_batch);
int la_latent_opt_set_lpips(la_latent_opt* h, la_feat* f, const float* bankF, long Mf, int S, float pre_scale,
                            float pre_shift, void* ws, size_t ws_bytes);
/* Per-channel input affine of the feature net, x_k * scale[k] + shift[k] for the n (<= 3) repeated channels of :394 -- the
 * (x - mean_k) / std_k input layer inside NVIDIA's TorchScript vgg16.pt (util_latent_aug.py:35-43).  Overrides the scalar pair. */
int la_latent_opt_set_lpips_preproc(la_latent_opt* h, const float* scale, const float* shift, int n);
int la_latent_opt_set_crop_pos(la_latent_opt* h, int x, int y);
/* Launch mode of the step loop (util_latent_aug.py:219-276).  1 (default): one optimisation step is captured as a hipGraph
 * after its first eager execution and replayed for every further step and batch of the same size -- the loop is ~230 short
 * launches per step and otherwise host-launch-bound at small batches.  0: every launch eager.  Results are identical. */
int la_latent_opt_set_graph(la_latent_opt* h, int enable);
/* 1: a captured step is being replayed; 0: eager launches (as asked / nothing run yet); -1: eager because the runtime refused the
   capture of the step.  (No reference counterpart: the reference loop is eager PyTorch, util_latent_aug.py:240-300.) */
int la_latent_opt_graph_state(const la_latent_opt* h);
/* Per-step snapshots for the reference's verbose_log (util_latent_aug.py:292-295 snap_w / snap_img): device buffers (or NULL)
 * w_trace [steps][B][w_dim] = the optimised latent after every step, img_trace [steps][B][C][R][R] = the image synthesised in
 * every step.  While either is set the loop launches eagerly. */
int la_latent_opt_set_trace(la_latent_opt* h, float* w_trace, float* img_trace);
/* dw_trace [steps][B][w_dim] (device, or NULL) = dL/dw of every step, L = -latent - pix - lpips + disc (util_latent_aug.py:270):
 * the tensor `loss.backward()` leaves in w_opt.grad (:275) before Adam consumes it.  While set the loop launches eagerly. */
int la_latent_opt_set_grad_trace(la_latent_opt* h, float* dw_trace);
/* 1 (default): with both the discriminator and the perceptual criterion active, the two run side by side inside a step -- the
 * discriminator branch on the launch stream, crop + feature net forward / backward on a stream of the handle's own, forked after the
 * synthesis forward and joined before the crop gradient is added to the image gradient (two parallel branches of the captured step).
 * 0: one after the other.  Bit-identical results either way (same launches, same accumulation order).  Drops a captured step. */
int la_latent_opt_set_overlap(la_latent_opt* h, int enable);
/* Image rows [row_lo, row_hi) that the loop's image criteria read (0, 0 = not known, the default).  The reference synthesises a whole
 * frame in every epoch and hands the pixel criterion its centre crop and the perceptual criterion a window inside it
 * (util_latent_aug.py:216, :246-262; util_dataset.py:284-323): no output of the loop depends on the other rows.  With a window given, the
 * synthesis passes of the loop steps compute only what those rows depend on (la_synth_set_row_window); the final synthesis of the
 * augmented latent (:303) is a whole frame.  Ignored while the discriminator (whole frame, :233-242) is active or per-step images are
 * traced.  Drops a captured step when the window changes. */
int la_latent_opt_set_row_window(la_latent_opt* h, int row_lo, int row_hi);
/* ... and the image columns [col_lo, col_hi) they read (the centre crop is a square: util_dataset.py:317-323); 0, 0 = all.  Used with the row window. */
int la_latent_opt_set_col_window(la_latent_opt* h, int col_lo, int col_hi);
/* verbose_log timers of the reference's first batch (time_latent / time_disc / time_pix / time_lpips / time_epoch,
 * util_latent_aug.py:221-272): with the time trace on, a run that asks for the loss scalars brackets the criteria of every step with
 * HIP events on the launch stream; la_latent_opt_get_times (after the stream has drained, or blocking) fills ms [steps][5] =
 * {latent, disc, pix, lpips, epoch} in milliseconds.  A criterion's bracket holds its loss scalar and its gradient launches. */
int la_latent_opt_set_time_trace(la_latent_opt* h, int enable);
int la_latent_opt_get_times(la_latent_opt* h, float* ms);
/* The banks handed to la_latent_opt_create / _set_lpips (register_buffer('W'/'X'/'fea_*'), util_latent_aug.py:137-171) must stay
 * IMMUTABLE for the life of the handle: their column sums are reduced once (both criterion modes) and every later gradient uses
 * them.  A caller that does rewrite bank contents in place calls this before the next la_latent_opt_run. */
int la_latent_opt_invalidate_banks(la_latent_opt* h);
/* w0 [B][w_dim] -> img_out [B][C][R][R], w_aug_out [B][num_ws][w_dim]; losses_out (may be NULL) [steps][4] =
 * weighted {latent, pix, disc, lpips} per step. */
int la_latent_opt_run(la_latent_opt* h, const float* w0, int B, const float* const* final_noises, float* img_out,
                      float* w_aug_out, float* losses_out, la_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Quality metrics on the augmented outputs: the numeric core of the reference's FID and Improved
 * Precision/Recall downstream of the detector features (the detectors are NVIDIA-hosted pickles, metric_utils.py:46-60).
 *   la_feature_moments_f64  FeatureStats.append, metrics/metric_utils.py:104-118: raw_mean[D] += sum_k x[k];
 *                           raw_cov[D][D] += x^T x with float64 accumulators; x float32 [n][D].
 *   la_cdist_f16            compute_distances, metrics/precision_recall.py:19-32 (torch.cdist of float16 features):
 *                           dist float32 [nr][nc].  rows/cols: float16 [n][D] row-major, D % 16 == 0, 16-byte aligned.
 *   la_pr_kth_f16           precision_recall.py:75-79: kth[i] = (nhood_size+1)-th smallest distance of row i.
 *   la_pr_member_f16        precision_recall.py:80-84: member[i] = any_j dist(i, j) <= radius[j].
 * ws: la_pr_workspace_floats(nr, nc) floats.  The [nr][nc] matrix is not materialised by the last two.
 * ------------------------------------------------------------------------------------------------------------- */
int la_feature_moments_f64(const float* x, long n, int D, double* raw_mean, double* raw_cov, la_stream_t stream);
size_t la_pr_workspace_floats(long nr, long nc);
int la_cdist_f16(const void* rows, long nr, const void* cols, long nc, int D, float* dist, float* ws, la_stream_t stream);
int la_pr_kth_f16(const void* rows, long nr, const void* cols, long nc, int D, int nhood_size, float* kth, float* ws,
                  la_stream_t stream);
int la_pr_member_f16(const void* rows, long nr, const void* cols, long nc, int D, const float* radius, unsigned char* member,
                     float* ws, la_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Opt-in profiler for the contraction launches (HIP events on the launch stream).  No reference counterpart: the
 * reference's only timing hook is wall-clock stats_time (augments/latent_aug.py:276).
 * la_prof_end: summed device ms, launch count, algorithmic FLOPs (2*MACs) and algorithmic bytes (input + output +
 * weights, each once) of every la_conv launch since la_prof_begin.
 * ------------------------------------------------------------------------------------------------------------- */
int la_prof_begin(void);
int la_prof_end(double* total_ms, long* launches, double* flops, double* bytes);
/* la_prof_set_stride(k): bracket a hashed 1-in-k sample of the launches instead of all of them (an event pair costs ~3 us on
 * the stream); la_prof_end then reports the sampled launches' ms / count / FLOPs / bytes, la_prof_total_launches() all of them. */
int la_prof_set_stride(int stride);
/* (The development build, `make dev` -> liblatentaug_hip_dev.so, additionally exports `la_dev_knob_set` (int id, int value -> int): it
 *  selects kernel variants for in-process A/B timing by scripts/bench_layer.py --ab.  The product library has no such symbol, no
 *  kernel-variant state and reads no LA_* environment variable.) */
long la_prof_total_launches(void);
/* Per kernel class (la_prof_num_classes() entries per array): 0 contraction / halo, 1 contraction / flat, 2 contraction /
 * split-K incl. its finish pass, 3 contraction / exact-fp32 MFMA, 4 operand preparation (plane maxima, pre-split copy),
 * 5 FIR (upfirdn2d family), 6 backward seam (bias_act backward + ToRGB backward), 7 ToRGB forward, 8 bank scans.
 * Bytes are the algorithmic ones (each operand once).  While the profiler is on, the step loop launches eagerly. */
int la_prof_num_classes(void);
int la_prof_end_classes(double* ms, long* launches, double* flops, double* bytes, int nclass);

#ifdef __cplusplus
}
#endif
#endif /* LATENTAUG_HIP_H */
