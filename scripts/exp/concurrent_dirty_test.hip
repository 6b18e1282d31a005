// dev experiment: do kernel boundaries on one stream lose the not-yet-written-back L2 lines of a kernel running on another stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void slow_fill(float* p, long n, float v, int spin) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float a = v;
        for (int k = 0; k < spin; ++k) a = a * 1.0000001f + 1e-9f;
        p[i] = a;
    }
}
__global__ void tiny(float* q, float v) { q[threadIdx.x + blockIdx.x * blockDim.x] = v; }
int main() {
    const long n = 1 << 26;
    float *a, *q;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&q, 1 << 20));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<float> h(n);
    int bad = 0;
    for (int it = 0; it < 10; ++it) {
        const float v = 1.f + it;
        slow_fill<<<2048, 256, 0, s1>>>(a, n, v, 200);
        for (int k = 0; k < 300; ++k) tiny<<<64, 256, 0, s2>>>(q, (float)k);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(h.data(), a, n * 4, hipMemcpyDeviceToHost));
        long wrong = 0;
        for (long i = 0; i < n; ++i) if (h[i] < v) ++wrong;
        if (wrong) { ++bad; printf("iter %d: %ld stale elements of %ld\n", it, wrong, n); }
    }
    printf("%d bad iterations of 10\n", bad);
    return 0;
}
