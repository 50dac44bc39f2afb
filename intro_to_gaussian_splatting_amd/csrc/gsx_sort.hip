// Stable LSD radix sort of (key, value) pairs for the two orderings of the frame pipeline:
// depth rank (N 32-bit keys) and tile binning (D tile ids, whose count D exists only in device
// memory when the sort is enqueued).
//
// Why not the library sort: rocPRIM's onesweep is built for large inputs; at the sizes of this
// path (1M depth keys, ~4M 16-bit tile keys) its passes are latency bound (26.7 us per pass at 1M,
// 325 workgroups in a look-back chain) and every call adds ~7 small fill kernels (94 us of
// __amd_rocclr_fillBufferAligned per frame, rocprofv3 profiles/r1c).  It also needs the element
// count on the HOST, which forced a device->host read-back in the middle of every frame.
//
// This implementation is the classic three-kernel LSD pass (two kernels for small inputs, see
// SELF_SCAN below), sized for these inputs:
//   count    each workgroup histograms its 2048 (4096 beyond 8M items) consecutive items by the pass's 8-bit digit in LDS
//            and stores table[digit][workgroup];
//   scan     one workgroup per digit row turns its row into an exclusive prefix and records the
//            row total;
//   scatter  each workgroup re-reads its items, ranks them STABLY inside the workgroup (wave-level
//            match by ballots, per-wave running digit counters in LDS, waves and rounds in item
//            order), parks them in LDS in digit-major order, and streams them out: item j of the
//            parked order goes to digit base (exclusive sum of the row totals, recomputed in LDS)
//            + row prefix + (j - first j of its digit), so neighbouring lanes write neighbouring
//            addresses.
// No inter-workgroup communication inside a kernel (the hazards of MI355X's non-coherent per-XCD
// L2s never arise), no fills, no host-side count: `n_dev` points at the element count in device
// memory and the grid is sized by a host-known upper bound; workgroups past the count exit.
#include "gsx_internal.h"

namespace gsx {
namespace {

constexpr int kThreads = 256;                      // 4 wavefronts
constexpr int kBins = 256;                         // 8-bit digits
// ROUNDS = items per lane; a workgroup owns ROUNDS * 256 consecutive items.  8 (2048 items) keeps
// enough workgroups in flight at 1M items; 16 halves the digit table for inputs beyond 8M.
constexpr int kSmallRounds = 8, kLargeRounds = 16;
constexpr int64_t kLargeInput = 8 << 20;
// Up to this many workgroups (131 072 items) a pass has no row-scan launch: measured one frame in flight,
// 2 000 Gaussians 96 -> 85 us, 100 000 Gaussians 166 -> 161 us; beyond ~100 workgroups the
// per-workgroup table walk costs more than the launch it saves (207 workgroups: +7 us per sort).
constexpr int kSelfScanBlocks = 64;

__device__ __forceinline__ uint32_t load_count(const uint32_t *n_dev, uint32_t bound) {
    if (!n_dev) return bound;
    uint32_t n = *n_dev;
    return n < bound ? n : bound;
}

// Each thread owns kRounds CONSECUTIVE items (one or two 16-byte loads) -- the histogram does not
// care about order, so the count kernel reads wide; the scatter kernel needs the wave-striped order.
// BLOCK_MAJOR: table[workgroup][digit] (what the self-scanning scatter reads, coalesced over the
// digits) instead of table[digit][workgroup] (contiguous rows for the row-scan kernel).
template <typename Key, int kRounds, bool BLOCK_MAJOR>
__global__ void __launch_bounds__(kThreads)
    count_kernel(const Key *__restrict__ keys, const uint32_t *__restrict__ n_dev, uint32_t bound, int shift,
                 uint32_t *__restrict__ table, int nblocks) {
    __shared__ uint32_t h[kBins];
    const uint32_t n = load_count(n_dev, bound);
    constexpr int kItems = kThreads * kRounds;
    constexpr int kPerVec = 16 / sizeof(Key), kVecs = kRounds / kPerVec;   // 8 x u16 or 4 x u32 per 16 B
    static_assert(kRounds % kPerVec == 0, "a thread's items must fill whole 16-byte vectors");
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * (uint32_t)kItems + threadIdx.x * (uint32_t)kRounds;
    if (first + kRounds <= n) {
        const uint4 *src = reinterpret_cast<const uint4 *>(keys + first);   // first is a multiple of kRounds
#pragma unroll
        for (int v = 0; v < kVecs; ++v) {
            const uint4 q = src[v];
            const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (sizeof(Key) == 4) {
                    atomicAdd(&h[(w4[c] >> shift) & (kBins - 1)], 1u);
                } else {
                    atomicAdd(&h[((w4[c] & 0xFFFFu) >> shift) & (kBins - 1)], 1u);
                    atomicAdd(&h[((w4[c] >> 16) >> shift) & (kBins - 1)], 1u);
                }
            }
        }
    } else {
        for (int r = 0; r < kRounds; ++r)
            if (first + r < n) atomicAdd(&h[((uint32_t)keys[first + r] >> shift) & (kBins - 1)], 1u);
    }
    __syncthreads();
    if (BLOCK_MAJOR)
        table[(size_t)blockIdx.x * kBins + threadIdx.x] = h[threadIdx.x];
    else
        table[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// One workgroup per digit row: in-place exclusive scan of table[d][0..nblocks), total -> totals[d].
__global__ void __launch_bounds__(kThreads) row_scan_kernel(uint32_t *__restrict__ table, int nblocks,
                                                            uint32_t *__restrict__ totals) {
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t carry_s;
    uint32_t *row = table + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += kThreads) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        uint32_t x = v;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up((int)x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_sum[w] = x;
        __syncthreads();
        uint32_t before = carry_s;
        for (int k = 0; k < w; ++k) before += wave_sum[k];
        if (i < nblocks) row[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == kThreads - 1) carry_s = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// SELF_SCAN: there is no row-scan launch; `table` holds the raw per-workgroup counts and every
// workgroup adds up, for each digit, the counts of the workgroups before it and the row total itself
// (nblocks loads per thread).  Pays for small inputs, where a pass is three launch latencies and
// the table is a few KiB.
template <typename Key, int kRounds, bool SELF_SCAN>
__global__ void __launch_bounds__(kThreads)
    scatter_kernel(const Key *__restrict__ kin, const uint32_t *__restrict__ vin, Key *__restrict__ kout,
                   uint32_t *__restrict__ vout, const uint32_t *__restrict__ n_dev, uint32_t bound, int shift,
                   const uint32_t *__restrict__ table, const uint32_t *__restrict__ totals, int nblocks) {
    constexpr int kItems = kThreads * kRounds, kWaveItems = kItems / 4;
    __shared__ uint32_t cnt[4][kBins];   // per-wave running digit counts, then per-wave LDS bases
    __shared__ uint32_t gbase[kBins];    // global address of parked item j of digit d = gbase[d] + j
    __shared__ uint32_t wsum[4], lsum[4];
    __shared__ Key skey[kItems];
    __shared__ uint32_t sval[kItems];
    const uint32_t n = load_count(n_dev, bound);
    const uint32_t block_base = blockIdx.x * (uint32_t)kItems;
    if (block_base >= n) return;
    const uint32_t live = min((uint32_t)kItems, n - block_base);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < 4 * kBins; k += kThreads) (&cnt[0][0])[k] = 0;
    __syncthreads();

    // ---- stable rank of every item among the same-digit items of its wave's consecutive slice
    const uint32_t wave_base = block_base + (uint32_t)w * kWaveItems;
    const unsigned long long lt = (1ull << lane) - 1ull;
    Key key[kRounds];
    uint32_t val[kRounds];
    uint16_t rank[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const uint32_t i = wave_base + (uint32_t)r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? kin[i] : (Key)0;
        val[r] = valid ? vin[i] : 0u;
        const uint32_t d = ((uint32_t)key[r] >> shift) & (kBins - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t old = 0;
        if (valid) {
            const int leader = __ffsll((long long)peers) - 1;
            if (lane == leader) {
                old = cnt[w][d];
                cnt[w][d] = old + (uint32_t)__popcll(peers);
            }
            old = (uint32_t)__shfl((int)old, leader);
        }
        rank[r] = (uint16_t)(old + (uint32_t)__popcll(peers & lt));
    }
    __syncthreads();

    // ---- per digit d (thread d): where its run starts in the parked (digit-major) order and in
    //      the global output: smaller digits (row totals) + this digit in earlier workgroups (table)
    {
        const int d = threadIdx.x;
        const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        uint32_t t, before;
        if (SELF_SCAN) {
            t = 0;
            before = 0;
            int b = 0;
            for (; b + 8 <= nblocks; b += 8) {      // 8 independent loads in flight (block-major: coalesced over d)
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = table[(size_t)(b + u) * kBins + d];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    before += b + u < (int)blockIdx.x ? v[u] : 0u;
                    t += v[u];
                }
            }
            for (; b < nblocks; ++b) {
                const uint32_t v = table[(size_t)b * kBins + d];
                before += b < (int)blockIdx.x ? v : 0u;
                t += v;
            }
        } else {
            t = totals[d];
            before = table[(size_t)d * nblocks + blockIdx.x];
        }
        const uint32_t l = c0 + c1 + c2 + c3;
        uint32_t x = t, y = l;  // inclusive wave scans of the global totals and of the local counts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t xu = __shfl_up((int)x, o), yu = __shfl_up((int)y, o);
            if (lane >= o) {
                x += xu;
                y += yu;
            }
        }
        if (lane == 63) {
            wsum[w] = x;
            lsum[w] = y;
        }
        __syncthreads();
        uint32_t gb = 0, lb = 0;
        for (int k = 0; k < w; ++k) {
            gb += wsum[k];
            lb += lsum[k];
        }
        const uint32_t lstart = lb + y - l;                       // first parked slot of digit d
        const uint32_t gstart = gb + x - t + before;
        gbase[d] = gstart - lstart;
        cnt[0][d] = lstart;
        cnt[1][d] = lstart + c0;
        cnt[2][d] = lstart + c0 + c1;
        cnt[3][d] = lstart + c0 + c1 + c2;
    }
    __syncthreads();

    // ---- park in LDS, digit-major, stable
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const uint32_t i = wave_base + (uint32_t)r * 64 + lane;
        if (i < n) {
            const uint32_t d = ((uint32_t)key[r] >> shift) & (kBins - 1);
            const uint32_t pos = cnt[w][d] + rank[r];
            skey[pos] = key[r];
            sval[pos] = val[r];
        }
    }
    __syncthreads();

    // ---- stream out: consecutive j of one digit -> consecutive addresses
    for (uint32_t j = threadIdx.x; j < live; j += kThreads) {
        const Key k = skey[j];
        const uint32_t dst = gbase[((uint32_t)k >> shift) & (kBins - 1)] + j;
        kout[dst] = k;
        vout[dst] = sval[j];
    }
}

template <typename Key, int kRounds>
hipError_t sort_rounds(void *temp, Key *&kc, Key *&ka, uint32_t *&vc, uint32_t *&va, const uint32_t *n_dev,
                       int64_t bound, int key_bits, hipStream_t s) {
    constexpr int kItems = kThreads * kRounds;
    const int nblocks = (int)((bound + kItems - 1) / kItems);
    uint32_t *table = (uint32_t *)temp;
    uint32_t *totals = table + (size_t)kBins * nblocks;
    const bool self_scan = nblocks <= kSelfScanBlocks;
    for (int shift = 0; shift < key_bits; shift += 8) {
        if (self_scan)
            count_kernel<Key, kRounds, true><<<nblocks, kThreads, 0, s>>>(kc, n_dev, (uint32_t)bound, shift, table,
                                                                          nblocks);
        else
            count_kernel<Key, kRounds, false><<<nblocks, kThreads, 0, s>>>(kc, n_dev, (uint32_t)bound, shift, table,
                                                                           nblocks);
        if (self_scan) {
            scatter_kernel<Key, kRounds, true><<<nblocks, kThreads, 0, s>>>(kc, vc, ka, va, n_dev, (uint32_t)bound,
                                                                            shift, table, totals, nblocks);
        } else {
            row_scan_kernel<<<kBins, kThreads, 0, s>>>(table, nblocks, totals);
            scatter_kernel<Key, kRounds, false><<<nblocks, kThreads, 0, s>>>(kc, vc, ka, va, n_dev, (uint32_t)bound,
                                                                             shift, table, totals, nblocks);
        }
        Key *tk = kc; kc = ka; ka = tk;
        uint32_t *tv = vc; vc = va; va = tv;
    }
    return hipGetLastError();
}

template <typename Key>
hipError_t sort_impl(void *temp, Key *&kc, Key *&ka, uint32_t *&vc, uint32_t *&va, const uint32_t *n_dev,
                     int64_t bound, int key_bits, hipStream_t s) {
    if (bound <= 0) return hipSuccess;
    if (bound > kLargeInput) return sort_rounds<Key, kLargeRounds>(temp, kc, ka, vc, va, n_dev, bound, key_bits, s);
    return sort_rounds<Key, kSmallRounds>(temp, kc, ka, vc, va, n_dev, bound, key_bits, s);
}

}  // namespace

size_t radix_temp_bytes(int64_t max_items) {
    constexpr int kItems = kThreads * kSmallRounds;
    const size_t nblocks = (size_t)((max_items + kItems - 1) / kItems) + 1;
    return (kBins * nblocks + kBins) * sizeof(uint32_t);
}

hipError_t radix_sort_pairs_u32(void *temp, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                                uint32_t *&vals_alt, const uint32_t *n_dev, int64_t bound, int key_bits,
                                hipStream_t s) {
    return sort_impl<uint32_t>(temp, keys_cur, keys_alt, vals_cur, vals_alt, n_dev, bound, key_bits, s);
}

hipError_t radix_sort_pairs_u16(void *temp, uint16_t *&keys_cur, uint16_t *&keys_alt, uint32_t *&vals_cur,
                                uint32_t *&vals_alt, const uint32_t *n_dev, int64_t bound, int key_bits,
                                hipStream_t s) {
    return sort_impl<uint16_t>(temp, keys_cur, keys_alt, vals_cur, vals_alt, n_dev, bound, key_bits, s);
}

}  // namespace gsx
