// dev experiment: does a non-blocking side stream honour hipStreamWaitEvent on an event recorded in another stream (fork / join)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void slow_fill(float* p, long n, float v, int spin) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    float a = v;
    for (int k = 0; k < spin; ++k) a = a * 1.0000001f + 1e-9f;
    if (i < n) p[i] = a;
}
__global__ void copy_k(const float* a, float* b, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
int main(int argc, char** argv) {
    const long n = 1 << 24;
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipStream_t mainS = nullptr, side;
    if (argc > 1) hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    hipEvent_t ef, ej;
    hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    std::vector<float> h(n);
    int bad = 0;
    for (int it = 0; it < 20; ++it) {
        const float v = 1.f + it;
        slow_fill<<<n / 256, 256, 0, mainS>>>(a, n, v, 2000);
        hipEventRecord(ef, mainS);
        hipStreamWaitEvent(side, ef, 0);
        copy_k<<<n / 256, 256, 0, side>>>(a, b, n);
        hipEventRecord(ej, side);
        hipStreamWaitEvent(mainS, ej, 0);
        hipMemcpyAsync(h.data(), b, n * 4, hipMemcpyDeviceToHost, mainS);
        hipStreamSynchronize(mainS);
        long wrong = 0;
        for (long i = 0; i < n; i += 4097) if (h[i] < v) ++wrong;
        if (wrong) { ++bad; printf("iter %d: %ld stale samples\n", it, wrong); }
    }
    printf("%s main stream: %d bad iterations of 20\n", argc > 1 ? "user" : "null", bad);
    return 0;
}
