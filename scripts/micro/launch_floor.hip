// Floor of a dependent kernel chain on one stream: N launches of (a) an empty kernel, (b) a 32-block x 256-thread kernel that writes
// 128 KB, timed with events; then the same chain captured into a hipGraph and launched as one graph.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/launch_floor.hip -o scripts/micro/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void empty_kernel() {}
__global__ void small_kernel(float* p, int it) { p[blockIdx.x * 1024 + threadIdx.x * 4 + (it & 3)] = (float)it; }
int main() {
    const int N = 2000;
    hipStream_t s; CK(hipStreamCreate(&s));
    float* buf; CK(hipMalloc(&buf, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        auto chain = [&]() {
            for (int i = 0; i < N; ++i) {
                if (mode == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
                else hipLaunchKernelGGL(small_kernel, dim3(32), dim3(256), 0, s, buf, i);
            }
        };
        chain(); CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventRecord(e0, s)); chain(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: stream launches %.2f us per kernel\n", mode ? "small" : "empty", ms * 1e3 / N);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: graph launch     %.2f us per kernel\n", mode ? "small" : "empty", ms * 1e3 / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // what a fork costs the stream it forks FROM: the same chain of small kernels with (a) an event record after every kernel, (b) the
    // record + a wait on a second stream + a kernel there (the side-stream pattern of the backward sweep), (c) additionally the main stream
    // waiting for the side stream's `done` event of two launches ago
    {
        hipStream_t side; CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        const int M = 500;
        hipEvent_t ready[M], done[M];
        for (int i = 0; i < M; ++i) { CK(hipEventCreateWithFlags(&ready[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); }
        for (int mode = 0; mode < 4; ++mode) {
            auto chain = [&]() {
                for (int i = 0; i < M; ++i) {
                    hipLaunchKernelGGL(small_kernel, dim3(32), dim3(256), 0, s, buf, i);
                    if (mode >= 1) (void)hipEventRecord(ready[i], s);
                    if (mode >= 2) {
                        (void)hipStreamWaitEvent(side, ready[i], 0);
                        hipLaunchKernelGGL(small_kernel, dim3(32), dim3(256), 0, side, buf + 65536, i);
                        (void)hipEventRecord(done[i], side);
                    }
                    if (mode >= 3 && i >= 2) (void)hipStreamWaitEvent(s, done[i - 2], 0);
                }
            };
            chain(); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(side));
            float ms = 0;
            CK(hipEventRecord(e0, s)); chain(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(side));
            CK(hipEventElapsedTime(&ms, e0, e1));
            const char* what[] = {"kernels only", "+ event record after each", "+ side stream waits, runs a kernel, records", "+ main waits for the side's event of two launches ago"};
            printf("fork cost, main stream: %-55s %.2f us per kernel\n", what[mode], ms * 1e3 / M);
        }
    }
    return 0;
}
