// What would the look-back that replaces row_scan_kernel cost on MI355X?  ("look-back confined to an XCD's
// region", published with agent-scope release / acquire.)  The chain ALONE, no sorting work: 8 regions (XCD x = blockIdx % 8
// takes the x-th eighth of the chunks, like scatter_kernel) of `per` chunks; a workgroup's thread d < ND owns digit d:
//   publish (AGGREGATE | count) for its chunk, walk back over its region's earlier chunks adding aggregates until a word flagged
//   INCLUSIVE turns up, publish (INCLUSIVE | sum + count).
// The tickets of a region are handed out by an atomic counter (a predecessor is then certainly running: no deadlock).
// Variants: agent-scope release / acquire (what the HIP memory model asks for between workgroups) and relaxed atomics without
// any fence (the floor: what a "same-XCD visibility" assumption would buy, unsafe).  Beside it: the row scan's job done the way
// the tree does it (one workgroup per digit row, 16 counts per thread) on the same table.  Sizes: C3's tile-sort pass (2 064 chunks,
// 128 digits), C2's (208 chunks) and C4's (10 400 chunks, 256 digits).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_lookback tools/microbench_lookback.hip && tools/microbench_lookback
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

constexpr uint32_t kAggregate = 1u << 30, kInclusive = 2u << 30, kValue = (1u << 30) - 1u;

template <bool FENCED>
__global__ void __launch_bounds__(256) lookback_kernel(uint32_t *status, uint32_t *tickets, uint32_t *out, uint32_t per, uint32_t nd,
                                                       uint32_t work_iters) {
    __shared__ uint32_t s_ticket;
    const uint32_t x = blockIdx.x & 7u;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&tickets[x * 32u], 1u);
    __syncthreads();
    const uint32_t i = s_ticket;
    if (i >= per) return;
    const uint32_t chunk = x * per + i, d = threadIdx.x;
    // stand-in for the local histogram + ranking of a scatter workgroup (keeps the workgroups as staggered as real ones)
    uint32_t cnt = (chunk * 2654435761u + d * 40503u) >> 28;
    for (uint32_t k = 0; k < work_iters; ++k) cnt = (cnt * 1664525u + 1013904223u) >> 28;
    if (d >= nd) return;
    uint32_t *mine = status + (size_t)chunk * nd + d;
    if (i == 0) {
        if (FENCED) __hip_atomic_store(mine, kInclusive | cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(mine, kInclusive | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out[(size_t)chunk * nd + d] = 0u;
        return;
    }
    if (FENCED) __hip_atomic_store(mine, kAggregate | cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(mine, kAggregate | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t sum = 0u;
    for (uint32_t j = i; j-- > 0u;) {
        const uint32_t *p = status + (size_t)(x * per + j) * nd + d;
        uint32_t s;
        do {
            s = FENCED ? __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                       : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((s >> 30) == 0u);
        sum += s & kValue;
        if (s & kInclusive) break;
    }
    if (FENCED) __hip_atomic_store(mine, kInclusive | ((sum + cnt) & kValue), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(mine, kInclusive | ((sum + cnt) & kValue), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[(size_t)chunk * nd + d] = sum;
}

// the tree's row scan, reduced to its memory pattern: table[d][chunk], one workgroup per row, 16 counts per thread
__global__ void __launch_bounds__(256) rowscan_kernel(uint32_t *table, int nbp, uint32_t *totals) {
    __shared__ uint32_t wave_sum[4];
    uint32_t *row = table + (size_t)blockIdx.x * nbp;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int seg = 0; seg < nbp; seg += 256 * 16) {
        const int i0 = seg + threadIdx.x * 16;
        uint32_t e[16], mine = 0;
        for (int k = 0; k < 16; ++k) e[k] = i0 + k < nbp ? row[i0 + k] : 0u;
        for (int k = 0; k < 16; ++k) { const uint32_t v = e[k]; e[k] = mine; mine += v; }
        uint32_t xs = mine;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up((int)xs, o); if (lane >= o) xs += y; }
        if (lane == 63) wave_sum[w] = xs;
        __syncthreads();
        uint32_t before = carry;
        for (int k = 0; k < w; ++k) before += wave_sum[k];
        before += xs - mine;
        for (int k = 0; k < 16; ++k) if (i0 + k < nbp) row[i0 + k] = before + e[k];
        carry += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

int main() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Case { const char *name; uint32_t chunks, nd; } cases[] = {{"C2 tile pass", 208, 128}, {"C3 tile pass", 2064, 128},
                                                                       {"C3 partition", 496, 256}, {"C4 tile pass", 10400, 256}};
    for (const Case &c : cases) {
        const uint32_t per = (c.chunks + 7u) / 8u, total = per * 8u;
        uint32_t *status, *tickets, *out, *table, *totals;
        hipMalloc(&status, (size_t)total * c.nd * 4);
        hipMalloc(&out, (size_t)total * c.nd * 4);
        hipMalloc(&tickets, 8 * 32 * 4);
        hipMalloc(&table, (size_t)total * c.nd * 4);
        hipMalloc(&totals, c.nd * 4);
        for (uint32_t work : {0u, 2000u}) {
            for (int fenced = 1; fenced >= 0; --fenced) {
                float best = 1e9f;
                for (int rep = 0; rep < 7; ++rep) {
                    hipMemsetAsync(status, 0, (size_t)total * c.nd * 4);
                    hipMemsetAsync(tickets, 0, 8 * 32 * 4);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    if (fenced) lookback_kernel<true><<<total, 256>>>(status, tickets, out, per, c.nd, work);
                    else lookback_kernel<false><<<total, 256>>>(status, tickets, out, per, c.nd, work);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                // check: out[chunk][d] = exclusive prefix of the counts inside the region
                std::vector<uint32_t> h((size_t)total * c.nd);
                hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
                size_t bad = 0;
                for (uint32_t x = 0; x < 8; ++x)
                    for (uint32_t d = 0; d < c.nd; d += 17) {
                        uint32_t run = 0;
                        for (uint32_t i = 0; i < per; ++i) {
                            const uint32_t chunk = x * per + i;
                            uint32_t cnt = (chunk * 2654435761u + d * 40503u) >> 28;
                            for (uint32_t k = 0; k < work; ++k) cnt = (cnt * 1664525u + 1013904223u) >> 28;
                            bad += h[(size_t)chunk * c.nd + d] != run;
                            run += cnt;
                        }
                    }
                printf("%-13s %5u chunks x %3u digits, look-back %-26s local work %4u iters: %8.1f us%s\n", c.name, c.chunks, c.nd,
                       fenced ? "agent release / acquire" : "relaxed, no fence (unsafe)", work, best * 1e3f, bad ? "  WRONG PREFIXES" : "");
            }
        }
        float best = 1e9f;
        for (int rep = 0; rep < 7; ++rep) {
            hipEventRecord(e0);
            rowscan_kernel<<<c.nd, 256>>>(table, (int)total, totals);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-13s %5u chunks x %3u digits, row scan (one workgroup per digit row, the tree's)               : %8.1f us\n", c.name, c.chunks,
               c.nd, best * 1e3f);
        hipFree(status); hipFree(out); hipFree(tickets); hipFree(table); hipFree(totals);
    }
    return 0;
}
