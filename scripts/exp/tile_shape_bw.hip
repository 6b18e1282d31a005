// Experiment: does the SHAPE of a 128-pixel tile matter for HBM efficiency when a workgroup streams C channel planes of an NCHW fp32
// tensor (what every contraction kernel here does on its way in and out)?  One workgroup = one tile of TH x TW pixels (TH * TW = 128)
// x all C channels of one sample: it reads them and writes them to a second tensor at the same positions, 4 bytes per lane, a
// half-wave on 32 consecutive pixels (the epilogue's store pattern) -- for 4 x 32 (the halo kernel's tile), 2 x 64 and 1 x 128.
//   hipcc --offload-arch=gfx950 -O3 -o tile_shape_bw tile_shape_bw.hip && ./tile_shape_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int TH, int TW, bool READ, bool WRITE>
__global__ __launch_bounds__(256) void tile_copy(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W) {
    const int tiles_x = W / TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.y;
    const int lane32 = threadIdx.x & 31, grp = threadIdx.x >> 5;      // 8 groups of 32 lanes
    // the tile's 128 pixels as 4 segments of 32 consecutive pixels: segment s -> (row, column block)
    const long plane = (long)H * W;
    float acc = 0.f;
#pragma unroll 4
    for (int it = grp; it < C * 4; it += 8) {
        const int c = it >> 2, s = it & 3;
        const int row = ty * TH + (s * 32) / TW, col = tx * TW + (s * 32) % TW + lane32;
        const long off = ((long)b * C + c) * plane + (long)row * W + col;
        float v = 1.f;
        if (READ) v = in[off];
        if (WRITE) out[off] = v * 2.f;
        else acc += v;
    }
    if (!WRITE && acc == 12345.678f) out[0] = acc;
}

template <int TH, int TW, bool READ, bool WRITE>
static double run(const float* in, float* out, int B, int C, int H, int W) {
    dim3 grid((H / TH) * (W / TW), B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_copy<TH, TW, READ, WRITE>), grid, dim3(256), 0, 0, in, out, C, H, W);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tile_copy<TH, TW, READ, WRITE>), grid, dim3(256), 0, 0, in, out, C, H, W);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}

// The halo kernel's memory pattern without its arithmetic: thread hp < 204 owns one pixel of the 6 x 34 halo of a 4 x 32 tile and loads
// it from all C planes (C loads in flight per thread, out-of-image pixels skipped), the values go through LDS (one float per pixel and
// channel group), then the workgroup writes MOUT output channels of its 4 x 32 pixels the way the epilogue does.
template <int CCH>
__global__ __launch_bounds__(256) void halo_pattern(const float* __restrict__ in, float* __restrict__ out, int C, int MOUT, int H, int W) {
    __shared__ float tile[204][CCH + 1];
    const int tiles_x = W / 32;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.y;
    const long plane = (long)H * W;
    const int hp = threadIdx.x;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += CCH) {
        float v[CCH];
#pragma unroll
        for (int k = 0; k < CCH; ++k) v[k] = 0.f;
        if (hp < 204) {
            const int hy = hp / 34, hx = hp - hy * 34;
            const int y = ty * 4 - 1 + hy, x = tx * 32 - 1 + hx;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const float* p = in + ((long)b * C + c0) * plane + (long)y * W + x;
#pragma unroll
                for (int k = 0; k < CCH; ++k) v[k] = p[(long)k * plane];
            }
#pragma unroll
            for (int k = 0; k < CCH; ++k) tile[hp][k] = v[k];
        }
        __syncthreads();
        // stand-in for the tap loop: every thread sums a few LDS values
        const int px = threadIdx.x & 127;
        const int r = px >> 5, cx = px & 31;
#pragma unroll
        for (int k = 0; k < CCH; k += 8) acc += tile[(r + 1) * 34 + cx + 1][k];
        __syncthreads();
    }
    const int lane32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (int it = grp; it < MOUT * 4; it += 8) {
        const int m = it >> 2, r = it & 3;
        out[((long)b * MOUT + m) * plane + (long)(ty * 4 + r) * W + tx * 32 + lane32] = acc + (float)m;
    }
}

static double run_halo(const float* in, float* out, int B, int C, int R) {
    dim3 grid((R / 4) * (R / 32), B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((halo_pattern<32>), grid, dim3(256), 0, 0, in, out, C, C, R, R);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((halo_pattern<32>), grid, dim3(256), 0, 0, in, out, C, C, R, R);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}

int main() {
    struct Cfg { int B, C, R; } cfgs[] = {{2, 32, 1024}, {4, 64, 512}, {8, 128, 256}, {8, 256, 128}, {8, 512, 64}};
    for (const Cfg& c : cfgs) {
        const size_t n = (size_t)c.B * c.C * c.R * c.R;
        float *in, *out;
        CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
        CK(hipMemset(in, 0, n * 4));
        const double mb = n * 4e-6;
        printf("B=%d C=%d %dx%d (%.0f MB per direction)\n", c.B, c.C, c.R, c.R, mb);
#define ROW(TH, TW) if (c.R % TW == 0) { \
            const double tc = run<TH, TW, true, true>(in, out, c.B, c.C, c.R, c.R), tr = run<TH, TW, true, false>(in, out, c.B, c.C, c.R, c.R), \
                         tw = run<TH, TW, false, true>(in, out, c.B, c.C, c.R, c.R); \
            printf("   tile %d x %-3d  copy %.1f us (%.2f TB/s of read + write)   read only %.1f us (%.2f TB/s)   write only %.1f us (%.2f TB/s)\n", TH, TW, \
                   tc, 2 * mb / tc, tr, mb / tr, tw, mb / tw); }
        ROW(4, 32) ROW(2, 64) ROW(1, 128)
        { const double th = run_halo(in, out, c.B, c.C, c.R); printf("   halo pattern (6 x 34 halo per 4 x 32 tile, %d loads in flight per thread, LDS, epilogue-style stores): %.1f us (%.2f TB/s of algorithmic read + write)\n", 32, th, 2 * mb / th); }
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
