// Stand-alone probe (hipcc --offload-arch=gfx950 -O3 vpk_overlap_probe.hip -o vpk_probe): does packed-FP32 arithmetic stay correct while
// another hardware queue keeps the chip busy?  Kernel A: a small 4-tap horizontal x vertical filter over planes, four outputs per thread,
// written (a) with v_pk_fma_f32 / v_pk_mul_f32 (inline asm: the instruction is what is probed) and (b) with scalar v_fma_f32 -- the shape
// of la_fir4x4_s1p_kernel, in which round 4 found the wrong values.  Kernel B: MFMA + memory traffic on a second stream.
// Every launch of A is compared bit for bit with its own solo result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { f2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
template <bool PK> __global__ __launch_bounds__(256) void fir_kernel(const float* __restrict__ in, float* __restrict__ out, int W, int H, int planes) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int w4 = W / 4, per = w4 * H;
    const int p = (int)(gid / per);
    if (p >= planes) return;
    const int r = (int)(gid - (long)p * per), y = r / w4, x0 = (r - y * w4) * 4;
    const float* ip = in + (long)p * (H + 3) * (W + 4);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const float fx[4] = {0.125f, 0.375f, 0.375f, 0.125f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float* rp = ip + (long)(y + a) * (W + 4) + x0;
        float c[8];
        *reinterpret_cast<f4*>(c) = *reinterpret_cast<const f4*>(rp); *reinterpret_cast<f4*>(c + 4) = *reinterpret_cast<const f4*>(rp + 4);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float wgt = fx[a] * fx[b];
            if (PK) {
                f2 lo = {acc[0], acc[1]}, hi = {acc[2], acc[3]};
                lo = pk_fma(f2{wgt, wgt}, f2{c[b], c[b + 1]}, lo); hi = pk_fma(f2{wgt, wgt}, f2{c[b + 2], c[b + 3]}, hi);
                acc = f4{lo[0], lo[1], hi[0], hi[1]};
            } else { acc[0] = __builtin_fmaf(wgt, c[b], acc[0]); acc[1] = __builtin_fmaf(wgt, c[b + 1], acc[1]); acc[2] = __builtin_fmaf(wgt, c[b + 2], acc[2]); acc[3] = __builtin_fmaf(wgt, c[b + 3], acc[3]); }
        }
    }
    *reinterpret_cast<f4*>(out + ((long)p * H + y) * W + x0) = acc;
}
__global__ __launch_bounds__(256) void busy_kernel(float* __restrict__ buf, long n, int iters) {      // MFMA chains + a stream over `buf`
    f16v acc = {0}; h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (k + 1)); }
    for (int i = 0; i < iters; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        const long o = ((long)blockIdx.x * 256 + threadIdx.x + (long)i * 65536 * 256) % n;
        buf[o] = buf[o] * 0.5f + acc[i & 15];
    }
}
int main() {
    const int W = 8, H = 8, planes = 4 * 512;      // the failing launches: 8x8 / 16x16 planes, 2048 of them
    const long nin = (long)planes * (H + 3) * (W + 4), nout = (long)planes * H * W, nbuf = 64l << 20;
    float *in, *out, *buf; hipMalloc(&in, nin * 4); hipMalloc(&out, nout * 4); hipMalloc(&buf, nbuf * 4);
    std::vector<float> h(nin); for (long i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice); hipMemset(buf, 0, nbuf * 4);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int grid = (int)((long)planes * (W / 4) * H + 255) / 256;
    for (int pk = 1; pk >= 0; --pk) {
        std::vector<float> ref(nout), got(nout);
        if (pk) fir_kernel<true><<<grid, 256, 0, s1>>>(in, out, W, H, planes); else fir_kernel<false><<<grid, 256, 0, s1>>>(in, out, W, H, planes);
        hipStreamSynchronize(s1); hipMemcpy(ref.data(), out, nout * 4, hipMemcpyDeviceToHost);
        long bad_launches = 0, bad_elems = 0; long hist[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 200; ++rep) {
            busy_kernel<<<2048, 256, 0, s2>>>(buf, nbuf, 400);
            for (int k = 0; k < 20; ++k) { if (pk) fir_kernel<true><<<grid, 256, 0, s1>>>(in, out, W, H, planes); else fir_kernel<false><<<grid, 256, 0, s1>>>(in, out, W, H, planes); }
            hipStreamSynchronize(s1); hipMemcpy(got.data(), out, nout * 4, hipMemcpyDeviceToHost); hipStreamSynchronize(s2);
            long nb = 0; for (long i = 0; i < nout; ++i) if (got[i] != ref[i]) { ++nb; ++hist[i & 3]; }
            if (nb) { ++bad_launches; bad_elems += nb; }
        }
        printf("%s: %ld of 200 overlapped rounds differ from the solo result (%ld elements; by output component x/y/z/w: %ld %ld %ld %ld)\n",
               pk ? "v_pk_fma_f32" : "v_fma_f32   ", bad_launches, bad_elems, hist[0], hist[1], hist[2], hist[3]);
    }
    return 0;
}
