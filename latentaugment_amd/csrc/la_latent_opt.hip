// The latent-optimisation loop: LatentAug.forward (augments/utils/util_latent_aug.py:207-310) as one host-driven
// launch sequence with no host<->device synchronisation inside the loop (the reference syncs 5x per step through
// `.item()`, :234-271).  W-space optimisation: ws = w repeated num_ws times (:226, :493-494).
#include "la_latent_opt.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "la_conv.h"
#include "la_criteria.h"
#include "la_disc.h"
#include "la_feat.h"

struct la_latent_opt {
    la_synth* g;
    la_disc* d;
    la_opt_config cfg;
    int R, imgc, wdim, num_ws, maxB;
    const float* bankW; long Mw;
    const float* bankX; long Mx;     // [imgc][Mx][cc*cc]
    float *w_opt, *m, *v, *dw, *dws, *g_img, *colsumW, *colsumX, *yx, *yy, *xx, *xc, *losses;
    int colsums_valid;
    // LPIPS criterion (la_latent_opt_set_lpips)
    la_feat* f;
    const float* bankF; long Mf; int F, S, crop_x, crop_y;
    float pre_scale[4], pre_shift[4];     // input affine of the feature net, per repeated channel
    float *l_xc, *l_feat, *l_gfeat, *l_gxc, *l_colsum, *l_yx, *l_yy, *l_xx;
    int l_colsum_valid;
    // step-invariant launch sequence: device-side step counter + Adam bias-correction table + crop position, so that ONE
    // captured step (hipGraph) is replayed for every step of every batch of the same size
    float* adam_tab;        // device [steps][2]
    float* adam_tab_host;   // host copy (malloc), uploaded ONCE (first run; constant afterwards)
    int adam_tab_valid;
    int* step_ctr;          // device (16 ints: [0] counter, [4..5] crop_dev, [8] ticket of la_step_tail)
    int* crop_dev;          // device {y0, x0}
    float* trace_dw;        // optional [steps][B][w_dim]: dL/dw of every step (la_latent_opt_set_grad_trace); forces eager launches
    int graph_mode;         // 0 eager, 1 replay a captured step (default)
    int graph_B;            // batch the captured step was built for (0: none)
    int graph_win;          // ... and whether its synthesis passes were windowed
    int graph_refused;      // 1: stream capture / instantiation of the step failed once; the handle launches eagerly since
    hipGraph_t graph;
    hipGraphExec_t graph_exec;
    // split replay (overlap mode 1, round 5): the step as FOUR graphs -- A: synthesis forward + pixel gradient, P: perceptual branch, D:
    // discriminator branch, Z: crop gradient + synthesis backward + step tail -- with P replayed on the side stream beside D (events)
    hipGraph_t seg_graph[4];
    hipGraphExec_t seg_exec[4];
    int seg_valid;
    hipStream_t cap_stream; // capture needs a non-default stream; the replay goes to the caller's stream
    float* trace_w;         // optional [steps][B][w_dim]: the latent after every step (verbose_log snapshots); forces eager launches
    float* trace_img;       // optional [steps][B][C][R][R]: the image synthesised in every step
    // Independent image criteria side by side (round 4): the discriminator branch on the launch stream, the perceptual branch (crop,
    // feature net forward + backward) on a side stream, forked after the synthesis forward and joined before the crop gradient is
    // added into g_img -- same arithmetic, same summation order, captured into the step's graph as two parallel branches.  Possible
    // since no kernel of the library contains packed-FP32 arithmetic ('Two streams').
    int wcol_lo, wcol_hi;   // ... and image columns (la_latent_opt_set_col_window)
    int win_lo, win_hi;     // image rows the criteria read (la_latent_opt_set_row_window; 0 / 0 = unknown: whole frames in every step)
    int overlap;            // 1 (default): fork / join when both criteria are active and no loss scalars / traces are asked for
    hipStream_t side_stream;
    hipEvent_t ev_fork, ev_join;
    // optional per-criterion times of the verbose_log batch (la_latent_opt_set_time_trace): LA_TEV events per step on the launch stream
    int time_trace, tev_steps;
    hipEvent_t* tev;
};
#define LA_TEV 7      // step start | after synthesis | after latent | after pix | after disc | after lpips | step end

static size_t al(size_t n) { return ((n * sizeof(float)) + 63) & ~(size_t)63; }

static size_t carve(la_latent_opt* h, char* base) {
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? (float*)(base + off) : nullptr; off += al(n); return p; };
    const size_t B = h->maxB, wd = h->wdim, cc2 = (size_t)h->cfg.crop * h->cfg.crop;
    h->w_opt = take(B * wd); h->m = take(B * wd); h->v = take(B * wd); h->dw = take(B * wd);
    h->dws = take(B * h->num_ws * wd);
    h->g_img = take(B * h->imgc * (size_t)h->R * h->R);
    h->colsumW = take((size_t)h->num_ws * wd);
    h->colsumX = take((size_t)h->imgc * cc2);
    const size_t mm = (size_t)(h->Mw > h->Mx ? h->Mw : h->Mx);
    h->yx = take(LA_YX_FLOATS(mm ? mm : 1, B)); h->yy = take(LA_YY_FLOATS(mm ? mm : 1)); h->xx = take(LA_XX_FLOATS(B));
    h->xc = take(B * h->imgc * cc2);
    h->losses = take((size_t)(h->cfg.steps > 0 ? h->cfg.steps : 1) * 4);
    h->adam_tab = take((size_t)(h->cfg.steps > 0 ? h->cfg.steps : 1) * 2);
    h->step_ctr = reinterpret_cast<int*>(take(16));
    h->crop_dev = h->step_ctr + 4;
    return off;
}

extern "C" size_t la_latent_opt_workspace_bytes(int img_resolution, int img_channels, int w_dim, const la_opt_config* cfg,
                                                long Mw, long Mx, int max_batch) {
    if (!cfg) return 0;
    la_latent_opt h; memset(&h, 0, sizeof(h));
    h.cfg = *cfg; h.R = img_resolution; h.imgc = img_channels; h.wdim = w_dim; h.num_ws = la_synth_num_ws(img_resolution);
    h.maxB = max_batch; h.Mw = Mw; h.Mx = Mx;
    return carve(&h, nullptr);
}

extern "C" int la_latent_opt_create(la_synth* g, int img_resolution, int img_channels, int w_dim, const la_opt_config* cfg,
                                    const float* bankW, long Mw, const float* bankXc, long Mx, int max_batch,
                                    void* workspace, size_t workspace_bytes, la_latent_opt** out) {
    LA_CHECK_ARG(g && cfg && workspace && out, "latent_opt_create: null pointer");
    LA_CHECK_ARG(cfg->steps >= 0, "latent_opt_create: negative step count");
    LA_CHECK_ARG(cfg->w_latent == 0.f || (bankW && Mw >= 1), "latent_opt_create: w_latent > 0 needs the latent bank W");
    LA_CHECK_ARG(cfg->w_pix == 0.f || (bankXc && Mx >= 1), "latent_opt_create: w_pix > 0 needs the cropped image bank X");
    LA_CHECK_ARG(cfg->crop >= 1 && cfg->crop_off >= 0 && cfg->crop + cfg->crop_off <= img_resolution,
                 "latent_opt_create: centre crop does not fit the image");
    LA_CHECK_ARG(cfg->criterion_mode == 0 || cfg->criterion_mode == 1, "latent_opt_create: criterion_mode must be 0 or 1");
    la_latent_opt* h = (la_latent_opt*)malloc(sizeof(la_latent_opt));
    LA_CHECK_ARG(h, "latent_opt_create: out of host memory");
    memset(h, 0, sizeof(*h));
    h->g = g; h->cfg = *cfg; h->R = img_resolution; h->imgc = img_channels; h->wdim = w_dim;
    h->num_ws = la_synth_num_ws(img_resolution); h->maxB = max_batch;
    h->bankW = bankW; h->Mw = cfg->w_latent != 0.f ? Mw : 0; h->bankX = bankXc; h->Mx = cfg->w_pix != 0.f ? Mx : 0;
    const size_t need = carve(h, (char*)workspace);
    if (need > workspace_bytes) { free(h); la_set_error("latent_opt_create: workspace too small"); return LA_ERR_WORKSPACE; }
    h->adam_tab_host = (float*)malloc(sizeof(float) * 2 * (size_t)(cfg->steps > 0 ? cfg->steps : 1));
    if (!h->adam_tab_host) { free(h); la_set_error("latent_opt_create: out of host memory"); return LA_ERR_ARG; }
    la_adam_fill_table(h->adam_tab_host, cfg->steps, cfg->beta1, cfg->beta2);
    h->graph_mode = 1;
    h->overlap = 2;
    h->win_lo = h->win_hi = 0; h->wcol_lo = h->wcol_hi = 0;
    *out = h;
    return LA_OK;
}

static void drop_graph(la_latent_opt* h) {
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    for (int k = 0; k < 4; ++k) {
        if (h->seg_exec[k]) { (void)hipGraphExecDestroy(h->seg_exec[k]); h->seg_exec[k] = nullptr; }
        if (h->seg_graph[k]) { (void)hipGraphDestroy(h->seg_graph[k]); h->seg_graph[k] = nullptr; }
    }
    h->seg_valid = 0;
    h->graph_B = 0; h->graph_win = 0;
}

extern "C" void la_latent_opt_destroy(la_latent_opt* h) {
    if (!h) return;
    drop_graph(h);
    if (h->tev) { for (int i = 0; i < h->tev_steps * LA_TEV; ++i) (void)hipEventDestroy(h->tev[i]); free(h->tev); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    free(h->adam_tab_host);
    free(h);
}

// 1 (default): the discriminator and the perceptual criterion of a step run side by side (two branches of the step); 0: one after the other.
// Results are bit-identical either way (same launches, same accumulation order into the image gradient).
extern "C" int la_latent_opt_set_overlap(la_latent_opt* h, int enable) {
    LA_CHECK_ARG(h, "latent_opt_set_overlap: null handle");
    // 0: criteria one after the other; 2 (the handle's default): fork / join captured as parallel branches of ONE step graph; 1: split
    // replay -- the perceptual branch as a graph of its own on the side stream beside the discriminator branch (round 5: measured equal)
    const int mode = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    if (h->overlap != mode) drop_graph(h);
    h->overlap = mode;
    drop_graph(h);
    return LA_OK;
}

// Image rows [row_lo, row_hi) that the IMAGE criteria of the loop read (0, 0 = not known): with the pixel criterion on its centre crop and
// the perceptual criterion on a window inside it, nothing in a loop step depends on the other rows of the synthesised image, and the
// synthesis passes of the steps deliver that window only (la_synth_set_row_window: the top blocks compute the rows it depends on).  The
// final synthesis of the augmented latent is always a whole frame.  Ignored while the discriminator (whole frame) is active or
// per-step images are traced.
extern "C" int la_latent_opt_set_row_window(la_latent_opt* h, int row_lo, int row_hi) {
    LA_CHECK_ARG(h && row_lo >= 0 && (row_hi == 0 ? row_lo == 0 : (row_hi > row_lo && row_hi <= h->R)), "latent_opt_set_row_window: bad window");
    if (row_lo != h->win_lo || row_hi != h->win_hi) drop_graph(h);
    h->win_lo = row_lo; h->win_hi = row_hi;
    return LA_OK;
}

// ... and the image columns [col_lo, col_hi) they read (0, 0 = all): used together with the row window (la_synth_set_col_window)
extern "C" int la_latent_opt_set_col_window(la_latent_opt* h, int col_lo, int col_hi) {
    LA_CHECK_ARG(h && col_lo >= 0 && (col_hi == 0 ? col_lo == 0 : (col_hi > col_lo && col_hi <= h->R)), "latent_opt_set_col_window: bad window");
    if (col_lo != h->wcol_lo || col_hi != h->wcol_hi) drop_graph(h);
    h->wcol_lo = col_lo; h->wcol_hi = col_hi;
    return LA_OK;
}

// 1 (default): steps 2..N of the first batch and every step of later batches replay ONE captured step; 0: every launch eager
extern "C" int la_latent_opt_set_trace(la_latent_opt* h, float* w_trace, float* img_trace) {
    LA_CHECK_ARG(h, "latent_opt_set_trace: null handle");
    h->trace_w = w_trace; h->trace_img = img_trace;
    return LA_OK;
}

// Per-criterion times of a run that asks for the loss scalars (the reference's verbose_log timers time_latent / time_disc / time_pix /
// time_lpips / time_epoch, util_latent_aug.py:221-272): HIP events on the launch stream around the criteria of every step.  Here a
// criterion's bracket holds its loss scalar AND its gradient launches (the reference's holds the forward only; its backward is inside
// the untimed loss.backward()), time_epoch the whole step.  Read back with la_latent_opt_get_times after the stream has drained.
extern "C" int la_latent_opt_set_time_trace(la_latent_opt* h, int enable) {
    LA_CHECK_ARG(h, "latent_opt_set_time_trace: null handle");
    if (enable && !h->tev && h->cfg.steps > 0) {
        h->tev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * (size_t)h->cfg.steps * LA_TEV);
        LA_CHECK_ARG(h->tev, "latent_opt_set_time_trace: out of host memory");
        for (int i = 0; i < h->cfg.steps * LA_TEV; ++i) LA_HIP(hipEventCreate(&h->tev[i]));
        h->tev_steps = h->cfg.steps;
    }
    h->time_trace = enable ? 1 : 0;
    return LA_OK;
}

// ms [steps][5] = {time_latent, time_disc, time_pix, time_lpips, time_epoch} in milliseconds of the last run that recorded them
// (waits for the events: call after the run's stream work is complete or let it block)
extern "C" int la_latent_opt_get_times(la_latent_opt* h, float* ms) {
    LA_CHECK_ARG(h && ms && h->tev, "latent_opt_get_times: no time trace recorded");
    for (int s = 0; s < h->tev_steps; ++s) {
        hipEvent_t* e = h->tev + (size_t)s * LA_TEV;
        LA_HIP(hipEventSynchronize(e[LA_TEV - 1]));
        float t[LA_TEV - 1];
        for (int k = 0; k + 1 < LA_TEV; ++k) LA_HIP(hipEventElapsedTime(&t[k], e[k], e[k + 1]));
        float* o = ms + (size_t)s * 5;
        o[0] = t[1]; o[2] = t[2]; o[1] = t[3]; o[3] = t[4];
        LA_HIP(hipEventElapsedTime(&o[4], e[0], e[LA_TEV - 1]));
    }
    return LA_OK;
}

// dL/dw [steps][B][w_dim] of every step (L = -latent - pix - lpips + disc, util_latent_aug.py:270): what Adam consumes, before
// its sign-like normalisation hides magnitudes.  Parity tests compare step 1 with a float64 gradient at full size.
extern "C" int la_latent_opt_set_grad_trace(la_latent_opt* h, float* dw_trace) {
    LA_CHECK_ARG(h, "latent_opt_set_grad_trace: null handle");
    h->trace_dw = dw_trace;
    return LA_OK;
}

// The banks (W, X crops, LPIPS features) are CONSTANTS of the handle: their column sums are reduced once and reused by every later
// step and batch.  A caller that rewrites bank contents in place must call this before the next run (the loss scalars are always
// computed from the live banks; without this call the gradient would keep using the old sums).
extern "C" int la_latent_opt_invalidate_banks(la_latent_opt* h) {
    LA_CHECK_ARG(h, "latent_opt_invalidate_banks: null handle");
    h->colsums_valid = 0;
    h->l_colsum_valid = 0;
    return LA_OK;
}

// 1: a captured step is instantiated and being replayed; 0: eager launches (as asked, or no batch has run yet); -1: eager because the
// capture of the step was refused by the runtime (la_latent_opt_set_graph(h, 1) re-arms it)
extern "C" int la_latent_opt_graph_state(const la_latent_opt* h) {
    if (!h) return 0;
    return (h->graph_exec || h->seg_valid) ? 1 : (h->graph_refused ? -1 : 0);
}

extern "C" int la_latent_opt_set_graph(la_latent_opt* h, int enable) {
    LA_CHECK_ARG(h, "latent_opt_set_graph: null handle");
    h->graph_mode = enable ? 1 : 0;
    h->graph_refused = 0;
    if (!enable) drop_graph(h);
    return LA_OK;
}

// attach the discriminator engine used by the w_disc criterion (required before la_latent_opt_run when w_disc != 0)
extern "C" int la_latent_opt_set_disc(la_latent_opt* h, la_disc* d) {
    LA_CHECK_ARG(h, "latent_opt_set_disc: null handle");
    h->d = d;
    drop_graph(h);
    return LA_OK;
}

// LPIPS criterion: feature engine, real-feature banks [imgc][Mf][F] (modality-major), crop size S (crop_size_aug) and the
// affine input preprocess x*scale + shift.  ws: la_latent_opt_lpips_workspace_bytes() of device memory.
extern "C" size_t la_latent_opt_lpips_workspace_bytes(int img_channels, int F, int S, long Mf, int max_batch) {
    const size_t n = (size_t)img_channels * max_batch;
    size_t off = 0;
    off += al(n * 3 * S * S) * 2;          // xc, gxc
    off += al(n * F) * 2;                  // feat, gfeat
    off += al((size_t)img_channels * F);   // colsum
    off += al(LA_YX_FLOATS(Mf, max_batch)) + al(LA_YY_FLOATS(Mf)) + al(LA_XX_FLOATS(max_batch));
    return off;
}

extern "C" int la_latent_opt_set_lpips(la_latent_opt* h, la_feat* f, const float* bankF, long Mf, int S, float pre_scale,
                                       float pre_shift, void* ws, size_t ws_bytes) {
    LA_CHECK_ARG(h && f && bankF && Mf >= 1 && S >= 1 && ws, "latent_opt_set_lpips: bad arguments");
    const int F = la_feat_num_features(f);
    LA_CHECK_ARG(ws_bytes >= la_latent_opt_lpips_workspace_bytes(h->imgc, F, S, Mf, h->maxB), "latent_opt_set_lpips: workspace too small");
    h->f = f; h->bankF = bankF; h->Mf = Mf; h->F = F; h->S = S; for (int k = 0; k < 4; ++k) { h->pre_scale[k] = pre_scale; h->pre_shift[k] = pre_shift; }
    char* base = (char*)ws; size_t off = 0;
    auto take = [&](size_t n) { float* p = (float*)(base + off); off += al(n); return p; };
    const size_t n = (size_t)h->imgc * h->maxB;
    h->l_xc = take(n * 3 * S * S); h->l_gxc = take(n * 3 * S * S);
    h->l_feat = take(n * F); h->l_gfeat = take(n * F);
    h->l_colsum = take((size_t)h->imgc * F);
    h->l_yx = take(LA_YX_FLOATS(Mf, h->maxB)); h->l_yy = take(LA_YY_FLOATS(Mf)); h->l_xx = take(LA_XX_FLOATS(h->maxB));
    h->l_colsum_valid = 0;
    drop_graph(h);
    return LA_OK;
}

// per-channel input affine of the feature net (the three repeated channels of :394): x_k * scale[k] + shift[k], e.g. the
// (x - mean_k) / std_k input layer of NVIDIA's vgg16.pt.  Overrides the scalar pair of la_latent_opt_set_lpips.
extern "C" int la_latent_opt_set_lpips_preproc(la_latent_opt* h, const float* scale, const float* shift, int n) {
    LA_CHECK_ARG(h && scale && shift && n >= 1 && n <= 4, "latent_opt_set_lpips_preproc: bad arguments");
    for (int k = 0; k < 4; ++k) { h->pre_scale[k] = scale[k < n ? k : n - 1]; h->pre_shift[k] = shift[k < n ? k : n - 1]; }
    drop_graph(h);
    return LA_OK;
}

// position of the crop_size_aug window inside the image, drawn by the host once per forward (util_dataset.py:284-296);
// (x, y) are absolute pixel coordinates (centre-crop offset already added)
extern "C" int la_latent_opt_set_crop_pos(la_latent_opt* h, int x, int y) {
    LA_CHECK_ARG(h && x >= 0 && y >= 0, "latent_opt_set_crop_pos: bad arguments");
    h->crop_x = x; h->crop_y = y;
    return LA_OK;
}

__global__ void la_lpips_gfeat_kernel(const float* __restrict__ feat, const float* __restrict__ colsum, float* __restrict__ g, int B,
                                      int F, float coef2, float mrows, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long n = i / F; const int k = (int)(i - n * F);
    const int c = (int)(n / B);
    g[i] = coef2 * (mrows * feat[i] - colsum[(long)c * F + k]);
}

// crop window position by value (no host buffer that a later call could overwrite while a copy is still in flight)
__global__ void la_set_int2_kernel(int* __restrict__ dst, int a, int b) { dst[0] = a; dst[1] = b; }

static int refresh_colsums(la_latent_opt* h, hipStream_t stream) {
    int rc;
    const long cc2 = (long)h->cfg.crop * h->cfg.crop;
    if (h->Mw && (rc = la_bank_colsum(h->bankW, h->Mw, (long)h->num_ws * h->wdim, h->colsumW, stream))) return rc;
    if (h->Mx)
        for (int c = 0; c < h->imgc; ++c)
            if ((rc = la_bank_colsum(h->bankX + (long)c * h->Mx * cc2, h->Mx, cc2, h->colsumX + (long)c * cc2, stream)))
                return rc;
    return LA_OK;
}

extern "C" int la_latent_opt_run(la_latent_opt* h, const float* w0, int B, const float* const* final_noises,
                                 float* img_out, float* w_aug_out, float* losses_out, hipStream_t stream) {
    LA_CHECK_ARG(h && w0 && img_out && w_aug_out, "latent_opt_run: null pointer");
    LA_CHECK_ARG(B >= 1 && B <= h->maxB, "latent_opt_run: batch exceeds max_batch");
    const la_opt_config& c = h->cfg;
    LA_CHECK_ARG(c.final_noise_mode != 2 || final_noises, "latent_opt_run: explicit final noise requested but not given");
    const int wd = h->wdim, cc = c.crop, off = c.crop_off;
    const long cc2 = (long)cc * cc;
    const long nw = (long)B * wd;
    const float nb = (float)(c.norm_batch > 0 ? c.norm_batch : B);
    int rc;
    LA_HIP(hipMemcpyAsync(h->w_opt, w0, nw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    LA_HIP(hipMemsetAsync(h->m, 0, nw * sizeof(float), stream));
    LA_HIP(hipMemsetAsync(h->v, 0, nw * sizeof(float), stream));
    LA_HIP(hipMemsetAsync(h->step_ctr, 0, 16 * sizeof(int), stream));      // step counter [0], crop position [4..5] (set below), ticket of la_step_tail [8]
    if (c.steps > 0 && !h->adam_tab_valid) {      // constant after create: one (host-blocking, pageable) upload per handle, not per batch
        LA_HIP(hipMemcpyAsync(h->adam_tab, h->adam_tab_host, sizeof(float) * 2 * (size_t)c.steps, hipMemcpyHostToDevice, stream));
        LA_HIP(hipStreamSynchronize(stream));
        h->adam_tab_valid = 1;
    }
    hipLaunchKernelGGL(la_set_int2_kernel, dim3(1), dim3(1), 0, stream, h->crop_dev, h->crop_y, h->crop_x);
    const bool use_disc = c.w_disc != 0.f;
    LA_CHECK_ARG(!use_disc || h->d, "latent_opt_run: w_disc != 0 but no discriminator attached (la_latent_opt_set_disc)");
    const bool use_lpips = c.w_lpips != 0.f;
    LA_CHECK_ARG(!use_lpips || h->f, "latent_opt_run: w_lpips != 0 but no feature engine attached (la_latent_opt_set_lpips)");
    if (use_disc && use_lpips && h->overlap && !h->side_stream) {      // once per handle: the side stream and the fork / join events
        LA_HIP(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        LA_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        LA_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    LA_CHECK_ARG(!use_lpips || (h->crop_x + h->S <= h->R && h->crop_y + h->S <= h->R), "latent_opt_run: LPIPS crop outside the image");
    const bool img_crit = c.w_pix != 0.f || use_disc || use_lpips;
    const float lp_coef = use_lpips ? c.w_lpips / ((float)h->imgc * (float)h->Mf * nb) : 0.f;
    // The loss SCALARS only feed the reference's log lines (:234-271); they are computed when the caller asks for them
    // (losses_out, i.e. verbose_log).  The gradient needs the bank column sums only, and the banks are constants of the handle:
    // reduced once (deterministic summation order, so this is bit-identical to re-reducing them every step).
    const bool want_losses = losses_out != nullptr;
    if (!h->colsums_valid) {
        if ((rc = refresh_colsums(h, stream))) return rc;
        h->colsums_valid = 1;
    }
    if (use_lpips && !h->l_colsum_valid) {
        for (int ch = 0; ch < h->imgc; ++ch)
            if ((rc = la_bank_colsum(h->bankF + (long)ch * h->Mf * h->F, h->Mf, h->F, h->l_colsum + (long)ch * h->F, stream))) return rc;
        h->l_colsum_valid = 1;
    }
    // loop steps: the synthesis delivers the rows the criteria read (la_latent_opt_set_row_window); whole frames with the discriminator,
    // with per-step image snapshots, or with no window given
    // The window is a caller-supplied hint (la_latent_opt_set_row_window / _col_window): it is honoured only while every row and column the
    // criteria of THIS run read lies inside it -- the pixel criterion's centre crop, the perceptual criterion's crop_size window at the
    // position of la_latent_opt_set_crop_pos.  A window that does not contain them would optimise against stale rows of an earlier pass;
    // such a run synthesises whole frames instead.
    auto inside = [&](int y0, int y1, int x0, int x1) {
        return y0 >= h->win_lo && y1 <= h->win_hi && (h->wcol_hi <= 0 || (x0 >= h->wcol_lo && x1 <= h->wcol_hi));
    };
    const bool win_covers = (c.w_pix == 0.f || inside(off, off + cc, off, off + cc)) &&
                            (!use_lpips || inside(h->crop_y, h->crop_y + h->S, h->crop_x, h->crop_x + h->S));
    const bool windowed = h->win_hi > 0 && !use_disc && !h->trace_img && win_covers;
    struct WinGuard {      // (whole frames again on every way out, and for the final synthesis below)
        la_synth* g; ~WinGuard() { (void)la_synth_set_row_window(g, 0, 0); (void)la_synth_set_col_window(g, 0, 0); }
    } win_guard{h->g};
    if ((rc = la_synth_set_row_window(h->g, windowed ? h->win_lo : 0, windowed ? h->win_hi : 0))) return rc;
    if ((rc = la_synth_set_col_window(h->g, windowed ? h->wcol_lo : 0, windowed ? h->wcol_hi : 0))) return rc;
    if (want_losses && c.steps > 0) LA_HIP(hipMemsetAsync(h->losses, 0, (size_t)c.steps * 4 * sizeof(float), stream));
    // loss = -loss_latent - loss_pix - loss_lpips + loss_disc  (:270): the diversity terms enter with a minus sign
    const float lat_coef = h->Mw ? c.w_latent / ((float)h->Mw * nb * (float)h->num_ws * (float)wd) : 0.f;
    const float pix_coef = h->Mx ? c.w_pix / ((float)h->imgc * (float)h->Mx * nb * (float)cc2) : 0.f;

    // one optimisation step; L = this step's row of loss scalars or null.  Everything it launches is independent of the step
    // number (the Adam bias corrections and the LPIPS window position are read from device memory), so the same launch
    // sequence can be captured once and replayed.
    // seg: which parts of the step to launch (1 = A: synthesis forward, loss scalars, pixel gradient; 2 = P: perceptual branch; 4 = D:
    // discriminator branch; 8 = Z: crop gradient, synthesis backward, step tail).  15 = the whole step; the split replay captures the four
    // parts as graphs of their own.  sl_over: the stream of part P when it is launched alone.
    auto run_step = [&](float* L, hipStream_t st, int step_index = -1, int seg = 15) -> int {
        int rc;
        const bool segA = seg & 1, segP = seg & 2, segD = seg & 4, segZ = seg & 8;
        hipEvent_t* tev = (h->time_trace && h->tev && L && step_index >= 0 && step_index < h->tev_steps) ? h->tev + (size_t)step_index * LA_TEV : nullptr;
        auto mark = [&](int k) { if (tev) (void)hipEventRecord(tev[k], st); };
        const float* img = la_synth_image(h->g);      // (the loop's image buffer: fixed per handle, also before the first pass)
        if (segA) {
            mark(0);
            if ((rc = la_synth_forward(h->g, h->w_opt, wd, 0, B, c.loop_noise_mode, nullptr, nullptr, st))) return rc;
            img = la_synth_image(h->g);
            mark(1);
            if (L) {
                if (h->Mw && (rc = la_l2_mean_from_bank(h->bankW, h->Mw, (long)h->num_ws * wd, h->w_opt, B, wd, wd, h->yx, h->yy,
                                                        h->xx, lat_coef, L + 0, 0, st)))
                    return rc;
            }
            // brackets of the per-criterion timers (la_latent_opt_get_times): [1,2) latent loss scalar, [2,3) pixel loss scalar + gradient,
            // [3,4) discriminator, [4,5) perceptual; the latent criterion's GRADIENT is one fused launch with the total (la_step_tail,
            // after the synthesis backward) and sits in the epoch bracket only
            mark(2);
            if (L) {
                if (h->Mx) {
                    if ((rc = la_center_crop_f32(img, h->xc, (long)B * h->imgc, h->R, cc, off, st))) return rc;
                    for (int ch = 0; ch < h->imgc; ++ch)
                        if ((rc = la_l2_mean_from_bank(h->bankX + (long)ch * h->Mx * cc2, h->Mx, cc2, h->xc + (long)ch * cc2, B,
                                                       (long)h->imgc * cc2, 0, h->yx, h->yy, h->xx, pix_coef, L + 1, ch > 0, st)))
                            return rc;
                }
            }
        }
        const float* dws = nullptr;
        if (img_crit) {
            if (segA) {
                if (c.w_pix != 0.f &&
                    (rc = la_pix_grad(img, h->colsumX, h->g_img, B, h->imgc, h->R, cc, off, -2.f * pix_coef, (float)h->Mx, st)))
                    return rc;
                mark(3);
            }
            // fork (whole-step launches, overlap mode 2): the perceptual branch needs the image only -- it runs on the side stream beside the
            // discriminator branch (the launch profiler, the loss scalars and the traces keep everything on one stream).  Overlap mode 1
            // (split replay) launches the parts one by one instead: the caller places part P on the side stream.
            const bool fork = seg == 15 && h->overlap == 2 && use_disc && use_lpips && !L && !tev && !la_prof_enabled() && h->side_stream && h->ev_fork && h->ev_join;
            hipStream_t sl = fork ? h->side_stream : st;      // stream of the perceptual branch
            if (fork) {
                LA_HIP(hipEventRecord(h->ev_fork, st));
                LA_HIP(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
            }
            if (use_disc && segD) {
                // loss_disc = softplus(-D(x)).mean() * w_disc enters the total with a plus sign (:270)
                if ((rc = la_disc_forward(h->d, img, B, st))) return rc;
                if ((rc = la_disc_loss(h->d, c.w_disc, c.norm_batch, L ? L + 2 : nullptr, st))) return rc;
                if ((rc = la_disc_backward(h->d, nullptr, h->g_img, c.w_pix != 0.f, st))) return rc;
            }
            if (segD) mark(4);
            if (use_lpips) {
                // loss_lpips = mean_modes( sum_{m,n} |f_n - F_m|^2 / (n*m) ) * w_lpips, entering the total with a minus sign (:270)
                const int N = h->imgc * B;
                const long FF = h->F;
                if (segP) {
                    if (c.w_pix == 0.f && !use_disc) LA_HIP(hipMemsetAsync(h->g_img, 0, sizeof(float) * (size_t)B * h->imgc * h->R * h->R, st));
                    if ((rc = la_crop_repeat_ex3(img, h->l_xc, B, h->imgc, h->R, h->S, h->crop_y, h->crop_x, h->crop_dev, 3, h->pre_scale, h->pre_shift, sl))) return rc;
                    if ((rc = la_feat_forward(h->f, h->l_xc, N, h->l_feat, sl))) return rc;
                    if (L) {
                        for (int ch = 0; ch < h->imgc; ++ch)
                            if ((rc = la_l2_mean_from_bank(h->bankF + (long)ch * h->Mf * FF, h->Mf, FF, h->l_feat + (long)ch * B * FF, B, FF, 0,
                                                           h->l_yx, h->l_yy, h->l_xx, lp_coef, L + 3, ch > 0, st)))
                                return rc;
                    }
                    const long total = (long)N * FF;
                    hipLaunchKernelGGL(la_lpips_gfeat_kernel, dim3(la_cdiv(total, 256)), dim3(256), 0, sl, h->l_feat, h->l_colsum, h->l_gfeat,
                                       B, (int)FF, -2.f * lp_coef, (float)h->Mf, total);
                    if ((rc = la_feat_backward(h->f, h->l_gfeat, h->l_gxc, sl))) return rc;
                }
                if (fork) {      // join: the crop gradient is added into g_img on the launch stream, after the discriminator's
                    LA_HIP(hipEventRecord(h->ev_join, h->side_stream));
                    LA_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
                }
                if (segZ && (rc = la_crop_repeat_grad_ex3(h->l_gxc, h->g_img, B, h->imgc, h->R, h->S, h->crop_y, h->crop_x, h->crop_dev, 3, h->pre_scale, st)))
                    return rc;
            }
            if (segZ) {
                mark(5);
                if ((rc = la_synth_backward(h->g, h->g_img, h->dws, st))) return rc;
            }
            dws = h->dws;
        } else if (segZ) { mark(3); mark(4); mark(5); }
        if (!segZ) return LA_OK;
        // dw = sum_ws dws + latent gradient, Adam, step counter: one launch (la_step_tail)
        rc = la_step_tail(dws, h->Mw ? h->colsumW : nullptr, h->dw, h->w_opt, h->m, h->v, B, h->num_ws, wd, -2.f * lat_coef, (float)h->Mw,
                          c.lr, c.beta1, c.beta2, c.eps, h->adam_tab, h->step_ctr, h->step_ctr + 8, st);
        mark(6);
        return rc;
    };

    // Replay of a captured step.  The step is captured AFTER one eager execution with the same batch size (module loading,
    // per-device attribute opt-ins and every other first-use effect happen outside the capture).  While the launch profiler
    // (la_prof_begin .. la_prof_end) is on, launches stay eager so that its event brackets see them.  If capture is refused the handle falls back to eager launches of the same kernels.
    const bool tracing = h->trace_w || h->trace_img || h->trace_dw;
    const bool replay = h->graph_mode == 1 && !want_losses && !tracing && c.steps > 0 && !la_prof_enabled();
    int first_graph_step = 1;
    // split replay (overlap mode 1, both image criteria on): four graphs per step, the perceptual one on the side stream
    const bool split = replay && h->overlap == 1 && use_disc && use_lpips && h->side_stream && h->ev_fork && h->ev_join;
    auto capture = [&](int seg, hipGraph_t* g, hipGraphExec_t* e) -> bool {
        bool ok = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
            const int rcs = run_step(nullptr, h->cap_stream, -1, seg);
            const hipError_t er = hipStreamEndCapture(h->cap_stream, g);
            ok = rcs == LA_OK && er == hipSuccess && *g;
        }
        if (ok) ok = hipGraphInstantiate(e, *g, nullptr, nullptr, 0) == hipSuccess;
        return ok;
    };
    const bool have = split ? (h->seg_valid != 0) : (h->graph_exec != nullptr);
    if (replay && (!have || h->graph_B != B || h->graph_win != (int)windowed)) {
        drop_graph(h);
        if ((rc = run_step(nullptr, stream))) return rc;          // step 1, eager
        first_graph_step = 2;
        bool ok = c.steps >= 2;
        if (ok && !h->cap_stream) ok = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) == hipSuccess;
        if (ok && split) {
            for (int k = 0; k < 4 && ok; ++k) ok = capture(1 << k, &h->seg_graph[k], &h->seg_exec[k]);
            if (ok) h->seg_valid = 1;
        } else if (ok) ok = capture(15, &h->graph, &h->graph_exec);
        if (ok) { h->graph_B = B; h->graph_win = (int)windowed; }
        else { drop_graph(h); (void)hipGetLastError(); if (c.steps >= 2) { h->graph_mode = 0; h->graph_refused = 1; } }
    }
    for (int step = first_graph_step; step <= c.steps; ++step) {
        if (replay && split && h->seg_valid && h->graph_B == B) {
            // A on the launch stream | fork | P on the side stream beside D on the launch stream | join | Z
            LA_HIP(hipGraphLaunch(h->seg_exec[0], stream));
            LA_HIP(hipEventRecord(h->ev_fork, stream));
            LA_HIP(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
            LA_HIP(hipGraphLaunch(h->seg_exec[1], h->side_stream));
            LA_HIP(hipEventRecord(h->ev_join, h->side_stream));
            LA_HIP(hipGraphLaunch(h->seg_exec[2], stream));
            LA_HIP(hipStreamWaitEvent(stream, h->ev_join, 0));
            LA_HIP(hipGraphLaunch(h->seg_exec[3], stream));
        } else if (replay && !split && h->graph_exec && h->graph_B == B) LA_HIP(hipGraphLaunch(h->graph_exec, stream));
        else {
            if ((rc = run_step(want_losses ? h->losses + (size_t)(step - 1) * 4 : nullptr, stream, step - 1))) return rc;
            // verbose_log snapshots (util_latent_aug.py:292-295): the image synthesised in this step, the latent after its update
            if (h->trace_img)
                LA_HIP(hipMemcpyAsync(h->trace_img + (size_t)(step - 1) * B * h->imgc * h->R * h->R, la_synth_image(h->g),
                                      sizeof(float) * (size_t)B * h->imgc * h->R * h->R, hipMemcpyDeviceToDevice, stream));
            if (h->trace_w) LA_HIP(hipMemcpyAsync(h->trace_w + (size_t)(step - 1) * nw, h->w_opt, sizeof(float) * nw, hipMemcpyDeviceToDevice, stream));
            if (h->trace_dw) LA_HIP(hipMemcpyAsync(h->trace_dw + (size_t)(step - 1) * nw, h->dw, sizeof(float) * nw, hipMemcpyDeviceToDevice, stream));
        }
    }
    if ((rc = la_broadcast_mix(h->w_opt, w0, w_aug_out, B, h->num_ws, wd, c.alpha, c.soft_aug, stream))) return rc;
    if ((rc = la_synth_set_row_window(h->g, 0, 0)) || (rc = la_synth_set_col_window(h->g, 0, 0))) return rc;      // the augmented image: a whole frame
    if ((rc = la_synth_forward(h->g, w_aug_out, (long)h->num_ws * wd, wd, B, c.final_noise_mode, final_noises, img_out, stream)))
        return rc;
    if (losses_out && c.steps > 0)
        LA_HIP(hipMemcpyAsync(losses_out, h->losses, (size_t)c.steps * 4 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return LA_OK;
}
