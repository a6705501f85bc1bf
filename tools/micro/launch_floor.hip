// Microbenchmark: per-kernel floor of a chain of dependent trivial kernels, eager vs hipGraph (MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void tiny1024(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 40, R = 200;
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
    CK(hipStreamSynchronize(s));
    auto t1 = std::chrono::steady_clock::now();
    printf("eager: %.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * R));
    for (int blocks : {1, 256, 2048}) for (int thr : {64, 1024}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(thr), 0, s, d);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t1 = std::chrono::steady_clock::now();
        printf("graph blocks=%d threads=%d: %.2f us per kernel\n", blocks, thr, std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * R));
    }
    return 0;
}
