// Perceptual feature extractor engine (opaque handle).  Public declarations: include/latentaug_hip.h.
#pragma once
#include <stddef.h>
#include "la_common.h"

#define LA_FEAT_CONV_RELU 0
#define LA_FEAT_TAP 1
#define LA_FEAT_MAXPOOL2 2
#define LA_FEAT_AVGPOOL2 3

extern "C" {
typedef struct la_feat_op { int kind, cin, cout; } la_feat_op;
struct la_feat;
size_t la_feat_workspace_bytes(int nops, const la_feat_op* ops, int in_ch, int in_res, int max_batch);
int la_feat_create(int nops, const la_feat_op* ops, const float* const* params, int nparams, int in_ch, int in_res, int max_batch,
                   void* workspace, size_t workspace_bytes, hipStream_t stream, la_feat** out);
void la_feat_destroy(la_feat* h);
int la_feat_num_features(const la_feat* h);
int la_feat_set_precision(la_feat* h, int precision);
int la_feat_forward(la_feat* h, const float* x, int N, float* feat_out, hipStream_t stream);
int la_feat_backward(la_feat* h, const float* gfeat, float* gx, hipStream_t stream);
int la_crop_repeat_f32(const float* img, float* xc, int B, int imgc, int R, int S, int y0, int x0, int rep, float scale, float shift,
                       hipStream_t stream);
int la_crop_repeat_grad_f32(const float* gxc, float* g_img, int B, int imgc, int R, int S, int y0, int x0, int rep, float scale,
                            hipStream_t stream);
}
// internal: window position read from device memory {y0, x0} when pos_dev != null (captured launches)
int la_crop_repeat_ex(const float* img, float* xc, int B, int imgc, int R, int S, int y0, int x0, const int* pos_dev, int rep, float scale,
                      float shift, hipStream_t stream);
int la_crop_repeat_grad_ex(const float* gxc, float* g_img, int B, int imgc, int R, int S, int y0, int x0, const int* pos_dev, int rep,
                           float scale, hipStream_t stream);
// per-repeated-channel affine ([rep] values each, rep <= 4): the input layer of an ImageNet-style net, (x - mean_k) / std_k
int la_crop_repeat_ex3(const float* img, float* xc, int B, int imgc, int R, int S, int y0, int x0, const int* pos_dev, int rep,
                       const float* scale, const float* shift, hipStream_t stream);
int la_crop_repeat_grad_ex3(const float* gxc, float* g_img, int B, int imgc, int R, int S, int y0, int x0, const int* pos_dev, int rep,
                            const float* scale, hipStream_t stream);
