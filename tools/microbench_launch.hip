// What does one more dependent kernel cost on this box, and do independent chains overlap?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_launch.hip -o tools/microbench_launch
// Prints wall time per kernel for chains of small kernels: one stream (eager / hipGraph), K chains on
// K streams, K chains as parallel branches of ONE graph (fork / join captured through events).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// "touch": every workgroup reads `words` uint4 per thread from src and writes one word per thread.
__global__ void __launch_bounds__(256) touch_kernel(const uint4 *__restrict__ src, uint32_t *__restrict__ dst, int words) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int k = 0; k < words; ++k) {
        const uint4 v = src[t * words + k];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    dst[t] = acc + 1u;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Chain {
    uint4 *src;
    uint32_t *dst;
};

static void enqueue_chain(const Chain &c, int m, int grid, int words, hipStream_t s) {
    for (int i = 0; i < m; ++i) touch_kernel<<<grid, 256, 0, s>>>(c.src, c.dst, words);
}

int main() {
    const int M = 40, REP = 30;
    const int grids[] = {64, 256, 512, 2048};
    const int wordsv[] = {0, 2};
    const int KMAX = 4;
    Chain ch[KMAX];
    hipStream_t st[KMAX];
    for (int k = 0; k < KMAX; ++k) {
        CK(hipMalloc(&ch[k].src, (size_t)2048 * 256 * 2 * 16));
        CK(hipMalloc(&ch[k].dst, (size_t)2048 * 256 * 4));
        CK(hipMemset(ch[k].src, 1, (size_t)2048 * 256 * 2 * 16));
        CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    }
    hipEvent_t fork, join[KMAX];
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int k = 0; k < KMAX; ++k) CK(hipEventCreateWithFlags(&join[k], hipEventDisableTiming));

    for (int words : wordsv)
        for (int grid : grids) {
            // warm up
            enqueue_chain(ch[0], M, grid, words, st[0]);
            CK(hipStreamSynchronize(st[0]));
            // eager, one stream
            double t0 = now_us();
            for (int r = 0; r < REP; ++r) enqueue_chain(ch[0], M, grid, words, st[0]);
            CK(hipStreamSynchronize(st[0]));
            const double eager = (now_us() - t0) / (REP * M);
            // graph, one chain
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
            enqueue_chain(ch[0], M, grid, words, st[0]);
            CK(hipStreamEndCapture(st[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st[0]));
            CK(hipStreamSynchronize(st[0]));
            t0 = now_us();
            for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st[0]));
            CK(hipStreamSynchronize(st[0]));
            const double graph1 = (now_us() - t0) / (REP * M);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
            printf("grid %4d x256, %d x16B/thread: eager %.2f us/kernel, graph %.2f us/kernel", grid, words, eager, graph1);
            for (int K : {2, 4}) {
                // K chains on K streams, eager (fork / join through events on stream 0)
                auto forkjoin = [&]() {
                    CK(hipEventRecord(fork, st[0]));
                    for (int k = 1; k < K; ++k) CK(hipStreamWaitEvent(st[k], fork, 0));
                    for (int k = 0; k < K; ++k) enqueue_chain(ch[k], M, grid, words, st[k]);
                    for (int k = 1; k < K; ++k) {
                        CK(hipEventRecord(join[k], st[k]));
                        CK(hipStreamWaitEvent(st[0], join[k], 0));
                    }
                };
                forkjoin();
                CK(hipStreamSynchronize(st[0]));
                t0 = now_us();
                for (int r = 0; r < REP; ++r) forkjoin();
                CK(hipStreamSynchronize(st[0]));
                const double eagerK = (now_us() - t0) / (REP * M);   // per kernel of ONE chain (K chains in parallel)
                CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
                forkjoin();
                CK(hipStreamEndCapture(st[0], &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, st[0]));
                CK(hipStreamSynchronize(st[0]));
                t0 = now_us();
                for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st[0]));
                CK(hipStreamSynchronize(st[0]));
                const double graphK = (now_us() - t0) / (REP * M);
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
                printf(" | %d chains: eager %.2f, graph %.2f us per chain-step", K, eagerK, graphK);
            }
            printf("\n");
        }
    return 0;
}
