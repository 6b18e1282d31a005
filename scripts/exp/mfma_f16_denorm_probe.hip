// Probe: do v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 and v_cvt_pk_f16_f32 keep fp16 SUBNORMAL values (gfx950)?
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/exp/mfma_f16_denorm_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, float tiny) {
    // A[row][k] = tiny (an fp16 subnormal after conversion) for k = 0, else 0;  B[k][col] = 1 for k = 0
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    const _Float16 t16 = (_Float16)tiny;             // v_cvt: kept or flushed?
    if (lane < 32) { a[0] = t16; b[0] = (_Float16)1.f; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
    f16x8 a2, b2;
    for (int j = 0; j < 8; ++j) { a2[j] = (_Float16)0.f; b2[j] = (_Float16)0.f; }
    if (lane < 16) { a2[0] = t16; b2[0] = (_Float16)1.f; }
    c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, c4, 0, 0, 0);
    if (lane == 0) { out[0] = (float)t16; out[1] = c[0]; out[2] = c4[0]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float vals[4] = {1e-5f /* normal: 2^-16.6 > 2^-14? no: 6.1e-5 is min normal -> 1e-5 is subnormal */, 3e-6f, 6e-8f, 1e-3f};
    for (float v : vals) {
        probe<<<1, 64>>>(d, v);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("input %.3e: after cvt to f16 %.6e | mfma 32x32x16 -> %.6e | mfma 16x16x32 -> %.6e\n", v, h[0], h[1], h[2]);
    }
    return 0;
}
