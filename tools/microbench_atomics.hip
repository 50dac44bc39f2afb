// What do device-scope atomic adds on global memory cost on MI355X, by how many addresses they share?  The question
// behind every "fuse the next pass' histogram into this kernel" idea for the binning chain: N adds
// without a returned value from N threads onto K counters (counter = thread % K, 128 bytes apart) -- K = 1: one list
// counter; 256 / 1 024: a digit table shared by all workgroups; 131 072: a (digit, chunk) table; and the same with the
// adds of a workgroup first combined in LDS (one global add per counter and workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/microbench_atomics tools/microbench_atomics.hip && /tmp/microbench_atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) direct_kernel(uint32_t *table, uint32_t k, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) atomicAdd(table + (size_t)(i % k) * 32u, 1u);
}

// (the digit of thread i is i % kk, kk <= 1 024: a workgroup's 2 048 items -- 8 per thread -- meet in an LDS table first)
__global__ void __launch_bounds__(256) lds_kernel(uint32_t *table, uint32_t kk, uint32_t n) {
    __shared__ uint32_t h[1024];
    for (uint32_t d = threadIdx.x; d < kk; d += 256u) h[d] = 0u;
    __syncthreads();
    for (uint32_t r = 0; r < 8u; ++r) {
        const uint32_t i = (blockIdx.x * 8u + r) * 256u + threadIdx.x;
        if (i < n) atomicAdd(&h[(i * 2654435761u >> 12) % kk], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < kk; d += 256u)
        if (h[d]) atomicAdd(table + (size_t)d * 32u, h[d]);
}

int main() {
    const uint32_t n = 4u << 20;
    uint32_t *table;
    hipMalloc(&table, (size_t)131072 * 128);
    hipMemset(table, 0, (size_t)131072 * 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t ks[] = {1, 16, 256, 1024, 131072};
    for (uint32_t k : ks) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            direct_kernel<<<(n + 255) / 256, 256>>>(table, k, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("direct: %u adds onto %6u counters: %8.1f us  (%.2f ns per add, %.1f ns per add and counter)\n", n, k, best * 1e3f,
               best * 1e6f / n, best * 1e6f / n * k);
    }
    for (uint32_t kk : {128u, 256u, 1024u}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            lds_kernel<<<(n + 2047) / 2048, 256>>>(table, kk, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("through LDS: %u items in %u workgroups of 2 048, %4u digits: %8.1f us  (%u global adds, %.1f ns per add and counter)\n",
               n, (n + 2047) / 2048, kk, best * 1e3f, (n + 2047) / 2048 * kk, best * 1e6f / ((n + 2047) / 2048));
    }
    return 0;
}
