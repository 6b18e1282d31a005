// Discriminator engine (opaque handle).  Public declarations with reference citations: include/latentaug_hip.h.
#pragma once
#include <stddef.h>
#include "la_common.h"

struct la_disc;
extern "C" {
int la_disc_num_params(int img_resolution);
size_t la_disc_workspace_bytes(int img_resolution, int img_channels, const int* channels, int max_batch);
int la_disc_create(int img_resolution, int img_channels, const int* channels, float conv_clamp, const float* const* params,
                   int nparams, const float* fir_host, int mbstd_group_size, int max_batch, void* workspace,
                   size_t workspace_bytes, hipStream_t stream, la_disc** out);
void la_disc_destroy(la_disc* h);
int la_disc_set_precision(la_disc* h, int precision);
int la_disc_forward(la_disc* h, const float* img, int B, hipStream_t stream);
int la_disc_loss(la_disc* h, float w_disc, int norm_batch, float* loss_out, hipStream_t stream);
int la_disc_backward(la_disc* h, const float* dlogits, float* g_img, int accumulate, hipStream_t stream);
const float* la_disc_logits(const la_disc* h);
}
