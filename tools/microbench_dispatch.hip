// How fast does the hardware start workgroups?  An (almost) empty kernel over the same number of WAVES, as single-wave
// workgroups and as four-wave workgroups; a body of ~2 us so that slots do not recycle during the ramp.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_dispatch.hip -o tools/microbench_dispatch && tools/microbench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void body(unsigned *out, int spin, unsigned long long *stamps) {
    const unsigned long long t0 = wall_clock64();
    unsigned x = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) out[0] = x;
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t0;
}

int main() {
    unsigned *out;
    unsigned long long *stamps;
    const int waves = 10240;
    hipMalloc(&out, 4);
    hipMalloc(&stamps, waves * 8);
    std::vector<unsigned long long> h(waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int spin : {0, 400, 4000}) {
        for (int threads : {64, 128, 256, 512}) {
            const int grid = waves * 64 / threads;
            float best = 1e9f;
            double ramp = 0;
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(e0);
                body<<<grid, threads>>>(out, spin, stamps);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
                hipMemcpy(h.data(), stamps, waves * 8, hipMemcpyDeviceToHost);
                const auto mm = std::minmax_element(h.begin(), h.end());
                ramp = (double)(*mm.second - *mm.first) * 0.01;
            }
            printf("spin %5d  %3d threads x %5d workgroups: kernel %.1f us, first-to-last wave start %.1f us\n", spin, threads, grid,
                   best * 1e3, ramp);
        }
    }
    return 0;
}
