// Experiment: an UPPER BOUND for a Winograd F(2x2, 3x3) form of the 128 -> 128 @ 256^2, B = 8 forward layer
// (154.6 GFLOP direct; the shipped halo kernel: 436-443 us).  Only the part Winograd cannot avoid is built -- the 16 per-frequency
// contractions  D_xi[cout][tile] += U_xi[cout][cin] * V_xi[cin][tile]  in the product's arithmetic (fp32 = two fp16 terms, 3 MFMAs
// v_mfma_f32_16x16x32_f16 per product), with the transformed weights U streamed from global memory exactly as the halo kernel streams
// its weights (fragment-order pack, one 16-byte load per lane and fragment, straight into registers) and the transformed input V read
// from LDS.  NOT built (all of it would only add time): loading the input halo, the input transform B^T d B (32 adds + 16 two-term
// fp16 splits per tile and channel), refilling LDS per K chunk (V is written once and re-read), the inverse transform A^T m A (the 16
// accumulators of a tile are just summed), bias / noise / activation.  If this skeleton is not clearly faster than the shipped kernel,
// the full Winograd kernel cannot be.
//   build: hipcc --offload-arch=gfx950 -O3 -o winograd_skeleton winograd_skeleton.hip ;  run: ./winograd_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int C = 128, M = 128, NCH = C / 32, NXI = 16;

// U pack: [xi][chunk][wave][mtile(2)][term(2)][lane(64)] x 16 bytes     V (LDS): [xi][ntile][term(2)][lane(64)] x 16 bytes
template <int NT>      // 16-tile column groups per workgroup: 2 -> 32 Winograd tiles = 128 output pixels, 1 -> 16 tiles = 64 pixels
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT == 1 ? 2 : 1))) void wino_skeleton(const uint4* __restrict__ U, float* __restrict__ out) {
    extern __shared__ uint4 V[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NXI * NT * 2 * 64; i += 256) {      // transformed input: written ONCE (a real kernel rewrites it per chunk)
        const _Float16 v = (_Float16)(0.001f * (float)((i * 37 + blockIdx.x) & 63));
        f16x8 x = {v, v, v, v, v, v, v, v};
        V[i] = __builtin_bit_cast(uint4, x);
    }
    __syncthreads();
    f32x4 acc[NXI][2][NT];
#pragma unroll
    for (int xi = 0; xi < NXI; ++xi)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[xi][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto ua = [&](int ch, int xi, int m, int t) { return U[((((long)xi * NCH + ch) * 4 + wave) * 2 + m) * 2 * 64 + t * 64 + lane]; };
    uint4 a[2][2][2];                                           // [buffer][mtile][term]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) a[0][m][t] = ua(0, 0, m, t);
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int xi = 0; xi < NXI; ++xi) {
            const int cur = xi & 1, nxt = cur ^ 1;
            const int nxi = xi + 1 < NXI ? xi + 1 : 0, nch = xi + 1 < NXI ? ch : (ch + 1 < NCH ? ch + 1 : ch);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) a[nxt][m][t] = ua(nch, nxi, m, t);      // next frequency's weight fragments: in flight under the MFMAs
            uint4 b[NT][2];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int t = 0; t < 2; ++t) b[n][t] = V[((xi * NT + n) * 2 + t) * 64 + lane];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, a[cur][m][0]), al = __builtin_bit_cast(f16x8, a[cur][m][1]);
                    const f16x8 bh = __builtin_bit_cast(f16x8, b[n][0]), bl = __builtin_bit_cast(f16x8, b[n][1]);
                    acc[xi][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[xi][m][n], 0, 0, 0);
                    acc[xi][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[xi][m][n], 0, 0, 0);
                    acc[xi][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[xi][m][n], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);      // one frequency at a time: the next one's fragments in flight, nothing hoisted further
        }
    }
    // stand-in for the inverse transform: 16 accumulators -> 4 outputs per tile by plain sums (4 frequencies each), stored as the
    // real kernel stores (fp32, 4 output pixels per tile and output channel)
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 32 + m * 16 + rq * 4 + r;
                float4 o;
                o.x = (acc[0][m][n][r] + acc[1][m][n][r]) + (acc[2][m][n][r] + acc[3][m][n][r]);
                o.y = (acc[4][m][n][r] + acc[5][m][n][r]) + (acc[6][m][n][r] + acc[7][m][n][r]);
                o.z = (acc[8][m][n][r] + acc[9][m][n][r]) + (acc[10][m][n][r] + acc[11][m][n][r]);
                o.w = (acc[12][m][n][r] + acc[13][m][n][r]) + (acc[14][m][n][r] + acc[15][m][n][r]);
                reinterpret_cast<float4*>(out)[((long)blockIdx.x * M + row) * (NT * 16) + n * 16 + col] = o;
            }
}

template <int NT>
static void run(const uint4* U, float* out, long tiles) {
    const int grid = (int)(tiles / (NT * 16));
    const size_t lds = (size_t)NXI * NT * 2 * 64 * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_skeleton<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wino_skeleton<NT>, dim3(grid), dim3(256), lds, 0, U, out);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wino_skeleton<NT>, dim3(grid), dim3(256), lds, 0, U, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    const double mfma_flops = (double)grid * 4 /*waves*/ * NCH * NXI * 2 * NT * 3 * (2.0 * 16 * 16 * 32);
    int nregs = 0;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(wino_skeleton<NT>)) == hipSuccess) nregs = fa.numRegs;
    printf("%2d tiles per workgroup (%d workgroups, %zu KB LDS, %d registers): %.1f us per launch; executed MFMA rate %.0f TFLOP/s = %.2f of 2.5 PF; "
           "weights streamed L2 -> registers: %.2f GB per launch\n",
           NT * 16, grid, lds >> 10, nregs, us, mfma_flops / us * 1e-6, mfma_flops / us * 1e-6 / 2500.0, (double)grid * NCH * NXI * 4 * 4 * 1024 * 1e-9);
}

int main() {
    const long tiles = 8L * (256 / 2) * (256 / 2);                  // B x (H/2) x (W/2) Winograd tiles
    const size_t ubytes = (size_t)NXI * NCH * 4 * 2 * 2 * 64 * 16;   // 1 MB
    std::vector<_Float16> hu(ubytes / 2);
    for (size_t i = 0; i < hu.size(); ++i) hu[i] = (_Float16)(0.01f * (float)((i * 13) & 31) - 0.15f);
    uint4* U; float* out;
    CK(hipMalloc(&U, ubytes));
    CK(hipMemcpy(U, hu.data(), ubytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)8 * M * 256 * 256 * 4));
    printf("Winograd F(2x2,3x3) skeleton, 128 -> 128 @ 256^2, B = 8 (direct: 154.6 GFLOP, shipped halo kernel 436-443 us, 48 MFMA-flop per direct flop/2.25 here)\n");
    run<2>(U, out, tiles);
    run<1>(U, out, tiles);
    CK(hipFree(U)); CK(hipFree(out));
    return 0;
}
