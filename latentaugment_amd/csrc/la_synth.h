// Synthesis engine (opaque handle).  Public declarations with reference citations: include/latentaug_hip.h.
#pragma once
#include <stddef.h>
#include "la_common.h"

struct la_synth;
extern "C" {
int la_synth_num_ws(int img_resolution);
int la_synth_num_params(int img_resolution);
size_t la_synth_workspace_bytes(int img_resolution, int img_channels, int w_dim, const int* channels, int max_batch);
int la_synth_create(int img_resolution, int img_channels, int w_dim, const int* channels, float conv_clamp,
                    const float* const* params, int nparams, const float* noise_strength, int nlayers,
                    const float* fir_host, int fir_h, int fir_w, int max_batch, void* workspace, size_t workspace_bytes,
                    hipStream_t stream, la_synth** out);
void la_synth_destroy(la_synth* h);
int la_synth_set_precision(la_synth* h, int precision);
int la_synth_get_precision(const la_synth* h);
int la_synth_set_row_window(la_synth* h, int row_lo, int row_hi);
int la_synth_set_col_window(la_synth* h, int col_lo, int col_hi);
int la_synth_forward(la_synth* h, const float* ws, long ws_bstride, long ws_lstride, int B, int noise_mode,
                     const float* const* noises, float* img_out, hipStream_t stream);
int la_synth_backward(la_synth* h, const float* g_img, float* dws, hipStream_t stream);
const float* la_synth_image(const la_synth* h);
const float* la_synth_block_image(const la_synth* h, int k);
const float* la_synth_layer_output(const la_synth* h, int k);
const float* la_synth_styles(const la_synth* h);
const float* la_synth_style_grads(const la_synth* h);
int la_synth_style_rows(const la_synth* h);
}
