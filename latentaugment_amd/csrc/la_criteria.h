// Internal (C++) interface of la_criteria.hip.
#pragma once
#include "la_common.h"

int la_bank_dot(const float* Y, long m, long K, const float* X, int n, long ldx, long xmod, float* yx, float* yy,
                hipStream_t stream);
int la_bank_colsum(const float* Y, long m, long K, float* colsum, hipStream_t stream);
// scratch sizes of the bank kernels (floats): yx 5*m*n, yy 5*m, xx 33*n
#define LA_YX_FLOATS(m, n) (17L * (m) * (n))      // result + up to 16 K-slice partials (la_criteria.hip KSPLIT)
#define LA_YY_FLOATS(m) (17L * (m))
#define LA_XX_FLOATS(n) (33L * (n))
// out[0] (+)= scale * sum_{m,n} (|Y_m|^2 + |X_n|^2 - 2<Y_m,X_n>); workspaces: yx_ws LA_YX_FLOATS, yy_ws LA_YY_FLOATS, xx_ws LA_XX_FLOATS
int la_l2_mean_from_bank(const float* Y, long m, long K, const float* X, int n, long ldx, long xmod, float* yx_ws,
                         float* yy_ws, float* xx_ws, float scale, float* out, int accumulate, hipStream_t stream);
int la_pix_grad(const float* img, const float* colsum, float* g, int B, int imgc, int R, int cc, int off, float coef2,
                float mrows, hipStream_t stream);
int la_latent_combine(const float* dws, const float* w, const float* colsumW, float* dw, int B, int num_ws, int wdim,
                      float lat2, float mrows, hipStream_t stream);
int la_broadcast_mix(const float* w_opt, const float* w0, float* w_aug, int B, int num_ws, int wdim, float alpha,
                     int soft, hipStream_t stream);
// loop-engine Adam: bias corrections from a device table indexed by a device-side step counter (la_misc.hip)
int la_adam_step_tab(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                     const float* tab, const int* ctr, hipStream_t stream);
int la_step_advance(int* ctr, hipStream_t stream);
// la_latent_combine + la_adam_step_tab + la_step_advance in one launch (bit-identical); ticket: a zeroed device int of the handle
int la_step_tail(const float* dws, const float* colsumW, float* dw, float* p, float* m, float* v, int B, int num_ws, int wdim, float lat2,
                 float mrows, float lr, float beta1, float beta2, float eps, const float* tab, int* ctr, int* ticket, hipStream_t stream);
void la_adam_fill_table(float* tab_host, int steps, float beta1, float beta2);
extern "C" {
long la_pairwise_l2_workspace_floats(int n, long m);
int la_pairwise_l2_f32(const float* X, int n, const float* Y, long m, long K, float* D, float* mean_out,
                       float* workspace, hipStream_t stream);
int la_center_crop_f32(const float* src, float* dst, long planes, int R, int cc, int off, hipStream_t stream);
int la_adam_step_f32(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1, float beta2,
                     float eps, hipStream_t stream);
}
