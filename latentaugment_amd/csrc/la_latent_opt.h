// Latent-optimisation loop engine.  Public declarations with reference citations: include/latentaug_hip.h.
#pragma once
#include "la_synth.h"

extern "C" {
typedef struct la_opt_config {
    int steps;              // opt_num_epochs                       (latent_aug.py:81)
    float lr;               // opt_lr                               (latent_aug.py:82)
    float beta1, beta2, eps;  // Adam (0.9, 0.999, 1e-8)            (util_latent_aug.py:213)
    float w_latent, w_pix, w_disc, w_lpips;   //                    (latent_aug.py:88-91)
    int criterion_mode;     // 0 | 1, kept for ABI stability: both use bank column sums reduced once per handle for the gradient;
                            // loss scalars (only when losses_out is given) always use the reference's GEMM form over the banks
    int soft_aug;           //                                      (latent_aug.py:94)
    float alpha;            //                                      (latent_aug.py:95)
    int loop_noise_mode;    // 1 = 'const'                          (util_latent_aug.py:227)
    int final_noise_mode;   // 0 none / 1 const / 2 explicit tensors (util_latent_aug.py:488 uses the G default, 'random')
    int norm_batch;         // n in the criteria's 1/(m*n); 0 = local batch (what DataParallel replicas see)
    int crop, crop_off;     // centre-crop size int(sqrt(R*R/2)) and offset round((R-crop)/2)  (util_dataset.py:317-323)
} la_opt_config;

struct la_latent_opt;
size_t la_latent_opt_workspace_bytes(int img_resolution, int img_channels, int w_dim, const la_opt_config* cfg, long Mw,
                                     long Mx, int max_batch);
int la_latent_opt_create(la_synth* g, int img_resolution, int img_channels, int w_dim, const la_opt_config* cfg,
                         const float* bankW, long Mw, const float* bankXc, long Mx, int max_batch, void* workspace,
                         size_t workspace_bytes, la_latent_opt** out);
void la_latent_opt_destroy(la_latent_opt* h);
struct la_disc;
int la_latent_opt_set_disc(la_latent_opt* h, la_disc* d);
struct la_feat;
size_t la_latent_opt_lpips_workspace_bytes(int img_channels, int F, int S, long Mf, int max_batch);
int la_latent_opt_set_lpips(la_latent_opt* h, la_feat* f, const float* bankF, long Mf, int S, float pre_scale, float pre_shift,
                            void* ws, size_t ws_bytes);
int la_latent_opt_set_crop_pos(la_latent_opt* h, int x, int y);
int la_latent_opt_set_graph(la_latent_opt* h, int enable);
int la_latent_opt_set_overlap(la_latent_opt* h, int enable);
int la_latent_opt_set_row_window(la_latent_opt* h, int row_lo, int row_hi);
int la_latent_opt_set_col_window(la_latent_opt* h, int col_lo, int col_hi);
int la_latent_opt_set_trace(la_latent_opt* h, float* w_trace, float* img_trace);
int la_latent_opt_set_time_trace(la_latent_opt* h, int enable);
int la_latent_opt_get_times(la_latent_opt* h, float* ms);
int la_latent_opt_run(la_latent_opt* h, const float* w0, int B, const float* const* final_noises, float* img_out,
                      float* w_aug_out, float* losses_out, hipStream_t stream);
}
