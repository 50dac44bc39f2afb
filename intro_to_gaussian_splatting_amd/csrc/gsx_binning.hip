// Depth ordering and tile binning on gfx950.
//
// The reference sorts all visible Gaussians once by view depth (splat/gaussian_scene.py:117) and
// then, for every tile, boolean-masks that sorted list (:209-218), so each tile's list is in
// global depth order.  Here the same lists are produced with two stable radix sorts:
//   1. N keys  (depth bits, value = original index)  -> depth rank of every Gaussian;
//   2. D keys  (window-local tile id, value = Gaussian index), emitted in rank order, so a
//      STABLE sort on the tile id alone (13 bits at 1080p, 2 radix passes instead of 6 for a
//      64-bit tile|depth key) leaves every tile's entries in depth order, ties broken by
//      original index exactly like a stable argsort.
// The radix sort (gsx_sort.hip) and the scan below are ours; no library primitive is left on the
// path.  D never leaves the device: every kernel after the scan reads it from offsets[n] and is
// launched on a grid sized by the workspace capacity.

#include "gsx_internal.h"

namespace gsx {
namespace {

constexpr int kBlock = 256;

// One thread per depth rank writes that Gaussian's (tile id, rank) pairs at its scan offset.
// A Gaussian covering more than kSerialMax tiles is spread over the whole wave instead, so one
// huge splat does not serialise 63 idle lanes behind it.
constexpr uint32_t kSerialMax = 16;

template <typename Key>
__device__ __forceinline__ void emit_one(const TileRect &r, uint32_t k, const TileGrid &g, uint32_t value,
                                         uint32_t base, uint32_t limit, Key *__restrict__ keys,
                                         uint32_t *__restrict__ vals) {
    if (base + k >= limit) return;  // speculative mode: the instance count exceeded the caller's hint
    uint32_t h = (uint32_t)(r.y1 - r.y0 + 1);
    uint32_t tx = r.x0 + k / h, ty = r.y0 + k % h;
    keys[base + k] = (Key)((tx - (uint32_t)g.wx0) * (uint32_t)g.nwy() + (ty - (uint32_t)g.wy0));
    vals[base + k] = value;
}

template <typename Key>
__global__ void __launch_bounds__(kBlock)
    emit_kernel(const TileRect *__restrict__ rect, const uint32_t *__restrict__ order,
                const uint32_t *__restrict__ offsets, int64_t n, TileGrid g, uint32_t limit,
                Key *__restrict__ keys, uint32_t *__restrict__ vals, uint2 *__restrict__ ranges) {
    int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r < g.count()) ranges[r] = make_uint2(0u, 0u);  // tiles without pairs keep an empty range
    TileRect tr;
    tr.x0 = 1; tr.x1 = 0; tr.y0 = 1; tr.y1 = 0;
    uint32_t base = 0, cnt = 0, gi = 0;
    if (r < n) {
        base = offsets[r];
        cnt = offsets[r + 1] - base;
        gi = order ? order[r] : (uint32_t)r;
        if (cnt) tr = rect[gi];
    }
    if (cnt <= kSerialMax)
        for (uint32_t k = 0; k < cnt; ++k) emit_one(tr, k, g, gi, base, limit, keys, vals);
    unsigned long long big = __ballot(cnt > kSerialMax);
    const int lane = threadIdx.x & 63;
    while (big) {
        int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        TileRect br;
        br.x0 = (uint16_t)__shfl((int)tr.x0, src);
        br.x1 = (uint16_t)__shfl((int)tr.x1, src);
        br.y0 = (uint16_t)__shfl((int)tr.y0, src);
        br.y1 = (uint16_t)__shfl((int)tr.y1, src);
        uint32_t bbase = (uint32_t)__shfl((int)base, src), bcnt = (uint32_t)__shfl((int)cnt, src);
        uint32_t bgi = (uint32_t)__shfl((int)gi, src);
        for (uint32_t k = lane; k < bcnt; k += 64) emit_one(br, k, g, bgi, bbase, limit, keys, vals);
    }
}

// ranges[t] = [first, last+1) of tile t inside the tile-sorted pair list; untouched (zeroed by
// the caller) for tiles with no entries.
// The grid covers the workspace capacity; the true pair count is read from device memory.
template <typename Key>
__global__ void __launch_bounds__(kBlock)
    tile_ranges_kernel(const Key *__restrict__ keys, const uint32_t *__restrict__ d_dev, uint32_t cap,
                       uint2 *__restrict__ ranges) {
    const uint32_t d = min(*d_dev, cap);
    const uint32_t j = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (j >= d) return;
    const Key t = keys[j];
    if (j == 0 || keys[j - 1] != t) ranges[t].x = j;
    if (j == d - 1 || keys[j + 1] != t) ranges[t].y = j + 1;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

// Exclusive scan of the tile counts in depth-rank order, reduce-then-scan in two launches:
//   1. every workgroup gathers counts[order[r]] for its 2048 ranks (the only random access),
//      parks them in offsets[r] and stores their sum in block_sums[b];
//   2. every workgroup adds up the block sums before it (at most a few thousand values) and
//      scans its own 2048 parked counts in place; the last one also stores the grand total D at
//      offsets[n].
constexpr int kScanItems = 8;
constexpr int kScanChunk = kBlock * kScanItems;

__device__ __forceinline__ uint32_t count_at(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ order,
                                             int64_t r, int64_t n) {
    return r < n ? counts[order ? order[r] : (uint32_t)r] : 0u;
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *wsum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_down((int)v, o);
    if (lane == 0) wsum[w] = v;
    __syncthreads();
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return total;
}

// sorted_keys (may be null): the depth keys in rank order; the number of visible Gaussians is
// where the culled keys start -- one writer, no atomics (a shared counter costs ~12 ns per
// wave-level atomic, 180 us at N = 1M).
__global__ void __launch_bounds__(kBlock)
    scan_block_sums_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ order,
                           const uint32_t *__restrict__ sorted_keys, int64_t n, uint32_t *__restrict__ block_sums,
                           uint32_t *__restrict__ offsets, uint32_t *__restrict__ n_visible) {
    __shared__ uint32_t wsum[4];
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanItems;
    if (sorted_keys && n == 0 && base == 0) *n_visible = 0u;   // no key to find the boundary at
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t r = base + k;
        const uint32_t c = count_at(counts, order, r, n);
        if (r <= n) offsets[r] = c;
        v += c;
        if (sorted_keys && r < n) {
            const bool vis = sorted_keys[r] != kCulledKey;
            if (r == 0 && !vis) *n_visible = 0u;
            if (vis && (r == n - 1 || sorted_keys[r + 1] == kCulledKey)) *n_visible = (uint32_t)(r + 1);
        }
    }
    const uint32_t total = block_sum(v, wsum);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// counts2[0] = n_visible, counts2[1] = D as int64: the first two fields of a GsxFrameStats.
__global__ void __launch_bounds__(kBlock)
    scan_apply_kernel(int64_t n, const uint32_t *__restrict__ block_sums, uint32_t *__restrict__ offsets,
                      const uint32_t *__restrict__ n_visible, int64_t n_visible_known,
                      int64_t *__restrict__ counts2) {
    __shared__ uint32_t wsum[4];
    __shared__ uint64_t wide[kBlock];
    uint32_t before = 0;
    uint64_t before64 = 0;  // the 32-bit offsets wrap beyond 2^32 pairs; the reported total must not
    for (uint32_t k = threadIdx.x; k < blockIdx.x; k += kBlock) {
        before += block_sums[k];
        before64 += block_sums[k];
    }
    const bool last = blockIdx.x == gridDim.x - 1;  // holds r == n
    if (last) {
        wide[threadIdx.x] = before64;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t t = 0;
            for (int k = 0; k < kBlock; ++k) t += wide[k];
            wide[0] = t;
        }
        __syncthreads();
        before64 = wide[0];
    }
    before = block_sum(before, wsum);
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanItems;
    uint32_t c[kScanItems], mine = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        c[k] = base + k <= n ? offsets[base + k] : 0u;
        mine += c[k];
    }
    // exclusive scan of `mine` over the workgroup
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint32_t run = before + x - mine;
    for (int k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k <= n) offsets[base + k] = run;   // r == n receives the grand total
        if (base + k == n) {
            counts2[0] = n_visible_known >= 0 ? n_visible_known : (int64_t)*n_visible;
            counts2[1] = (int64_t)(before64 + (uint64_t)(run - before));   // = run while D < 2^32
        }
        run += c[k];
    }
}

size_t binning_temp_bytes(int64_t n, int64_t cap) {
    const size_t scan = ((size_t)(n + 1 + kScanChunk - 1) / kScanChunk + 1) * sizeof(uint32_t);
    const size_t r = radix_temp_bytes(n > cap ? n : cap);
    const size_t m = scan > r ? scan : r;
    return (m + 255) & ~(size_t)255;
}

hipError_t sort_by_depth(void *temp, size_t temp_bytes, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                         uint32_t *&vals_alt, int64_t n, hipStream_t s) {
    (void)temp_bytes;
    if (n == 0) return hipSuccess;
    return radix_sort_pairs_u32(temp, keys_cur, keys_alt, vals_cur, vals_alt, nullptr, n, 32, s);
}

hipError_t scan_counts(void *temp, size_t temp_bytes, const uint32_t *counts, const uint32_t *order,
                       const uint32_t *sorted_keys, uint32_t *offsets, int64_t n, uint32_t *n_visible,
                       int64_t n_visible_known, int64_t *counts2, hipStream_t s) {
    (void)temp_bytes;
    const unsigned nb = (unsigned)((n + 1 + kScanChunk - 1) / kScanChunk);
    uint32_t *block_sums = (uint32_t *)temp;
    scan_block_sums_kernel<<<nb, kBlock, 0, s>>>(counts, order, sorted_keys, n, block_sums, offsets, n_visible);
    scan_apply_kernel<<<nb, kBlock, 0, s>>>(n, block_sums, offsets, n_visible, n_visible_known, counts2);
    return hipGetLastError();
}

// Emit (tile id, Gaussian index) pairs in rank order, sort them stably by tile id and derive
// every tile's [first, last) range.  Tile ids fit 16 bits for any frame up to 65536 tiles (4K has
// 32 026), which halves the key traffic of the sort; larger frames use 32-bit ids.
// keys0 / keys1 / vals0 / vals1 each hold `cap` 32-bit words; the pair count D = offsets[n] stays
// on the device, pairs beyond `cap` are dropped (the caller compares D with cap afterwards).
template <typename Key>
hipError_t bin_impl(void *temp, const TileRect *rect, const uint32_t *order, const uint32_t *offsets, int64_t n,
                    int64_t cap, const TileGrid &grid, void *keys0, void *keys1, uint32_t *vals0, uint32_t *vals1,
                    uint2 *ranges, int key_bits, const uint32_t **sorted_vals, hipStream_t s) {
    Key *kc = (Key *)keys0, *ka = (Key *)keys1;
    uint32_t *vc = vals0, *va = vals1;
    const uint32_t *d_dev = offsets + n;
    const int64_t nt = grid.count();
    emit_kernel<Key><<<blocks_for(n > nt ? n : nt), kBlock, 0, s>>>(rect, order, offsets, n, grid, (uint32_t)cap, kc,
                                                                     vc, ranges);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (sizeof(Key) == 2)
        e = radix_sort_pairs_u16(temp, (uint16_t *&)kc, (uint16_t *&)ka, vc, va, d_dev, cap, key_bits, s);
    else
        e = radix_sort_pairs_u32(temp, (uint32_t *&)kc, (uint32_t *&)ka, vc, va, d_dev, cap, key_bits, s);
    if (e != hipSuccess) return e;
    tile_ranges_kernel<Key><<<blocks_for(cap), kBlock, 0, s>>>(kc, d_dev, (uint32_t)cap, ranges);
    *sorted_vals = vc;
    return hipGetLastError();
}

hipError_t bin_instances(void *temp, size_t temp_bytes, const TileRect *rect, const uint32_t *order,
                         const uint32_t *offsets, int64_t n, int64_t cap, const TileGrid &grid, void *keys0,
                         void *keys1, uint32_t *vals0, uint32_t *vals1, uint2 *ranges, const uint32_t **sorted_vals,
                         hipStream_t s) {
    (void)temp_bytes;
    const int64_t nt = grid.count();
    *sorted_vals = vals0;
    if (cap == 0 || n == 0) return hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)nt, s);
    int bits = 1;
    while (((int64_t)1 << bits) < nt) ++bits;
    if (nt <= 65536)
        return bin_impl<uint16_t>(temp, rect, order, offsets, n, cap, grid, keys0, keys1, vals0, vals1, ranges, bits,
                                  sorted_vals, s);
    return bin_impl<uint32_t>(temp, rect, order, offsets, n, cap, grid, keys0, keys1, vals0, vals1, ranges, bits,
                              sorted_vals, s);
}

}  // namespace gsx
