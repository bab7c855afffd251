// Launch rate of this runtime: T host threads, each launching N empty kernels on a stream of its own (hipcc --offload-arch=gfx950 -O2 tools/launch_rate.hip -o /tmp/launch_rate -lpthread).
// Round 6, one MI355X box: 1 thread 2.5 us per launch (0.40 M launches/s), 2: 3.2 (0.63 M/s), 4: 3.5 (1.13 M/s), 8: 4.5 (1.79 M/s) — launches of different threads do not
// queue behind one lock: the ~310 launches of a pipelined configs[2] batch are ~1.1 ms of its estimator thread, not what bounds the 4.1-4.5 ms step.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void emptyKernel(int * p) { if (p && threadIdx.x == 1000) *p = 1; }
int main() {
    hipSetDevice(0);
    const int N = 20000;
    for (int threads : {1, 2, 4, 8}) {
        std::vector<hipStream_t> streams(threads);
        for (auto & s : streams) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back([&, t]() {
            hipSetDevice(0);
            for (int i = 0; i < N; ++i) emptyKernel<<<dim3(1), dim3(64), 0, streams[t]>>>(nullptr);
            hipStreamSynchronize(streams[t]);
        });
        for (auto & th : pool) th.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%d threads x %d launches: %.1f ms, %.2f us per launch per thread, %.2f M launches/s in all\n", threads, N, s * 1e3, s * 1e6 / N, threads * N / s / 1e6);
        for (auto & s2 : streams) hipStreamDestroy(s2);
    }
    return 0;
}
