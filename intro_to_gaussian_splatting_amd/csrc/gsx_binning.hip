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
// rocPRIM provides the device-wide radix sort and scan primitives; the emit / range kernels
// are ours.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/types/double_buffer.hpp>

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
                Key *__restrict__ keys, uint32_t *__restrict__ vals) {
    int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
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
// PADDED: the pair list was sized by the caller's hint and padded with all-ones keys.
template <typename Key, bool PADDED>
__global__ void __launch_bounds__(kBlock)
    tile_ranges_kernel(const Key *__restrict__ keys, int64_t d, uint2 *__restrict__ ranges) {
    int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= d) return;
    const Key t = keys[j];
    if (PADDED && t == (Key)~(Key)0) return;
    if (j == 0 || keys[j - 1] != t) ranges[t].x = (uint32_t)j;
    if (j == d - 1 || keys[j + 1] != t) ranges[t].y = (uint32_t)(j + 1);
}

// Device-side frame counts in the layout of the first two GsxFrameStats fields.
__global__ void publish_counts_kernel(const uint32_t *__restrict__ n_visible, const uint32_t *__restrict__ total,
                                      int64_t n_visible_known, int64_t *__restrict__ out2) {
    out2[0] = n_visible_known >= 0 ? n_visible_known : (int64_t)*n_visible;
    out2[1] = (int64_t)*total;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

// counts[order[r]] for r < n, 0 for r == n: the scan input in rank order without a gather pass.
struct PermutedCount {
    const uint32_t *counts, *order;
    uint32_t n;
    __host__ __device__ uint32_t operator()(uint32_t r) const {
        return r < n ? counts[order ? order[r] : r] : 0u;
    }
};
using CountIter = rocprim::transform_iterator<rocprim::counting_iterator<uint32_t>, PermutedCount, uint32_t>;

// Depth keys: always the onesweep radix sort (rocPRIM's default switches to a merge sort up to
// 1M items, which measured 160 us at N = 1M against ~70 us for four onesweep passes).
using DepthSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                   rocprim::default_config, 65536>;

template <typename Config, typename Key>
hipError_t sort_impl(void *temp, size_t &temp_bytes, Key *&keys_cur, Key *&keys_alt, uint32_t *&vals_cur,
                     uint32_t *&vals_alt, int64_t n, int end_bit, hipStream_t s) {
    rocprim::double_buffer<Key> kb(keys_cur, keys_alt);
    rocprim::double_buffer<uint32_t> vb(vals_cur, vals_alt);
    hipError_t e = rocprim::radix_sort_pairs<Config>(temp, temp_bytes, kb, vb, (size_t)n, 0u, (unsigned)end_bit, s);
    keys_cur = kb.current();
    keys_alt = kb.alternate();
    vals_cur = vb.current();
    vals_alt = vb.alternate();
    return e;
}

size_t binning_temp_bytes(int64_t n, int64_t cap) {
    size_t a = 0, b = 0, b16 = 0, c = 0;
    uint32_t *k = nullptr, *k2 = nullptr, *v = nullptr, *v2 = nullptr;
    uint16_t *h = nullptr, *h2 = nullptr;
    hipError_t e;
    e = sort_impl<DepthSortConfig>(nullptr, a, k, k2, v, v2, n > 0 ? n : 1, 32, (hipStream_t)0);
    if (e != hipSuccess) return 0;
    e = sort_impl<rocprim::default_config>(nullptr, b, k, k2, v, v2, cap > 0 ? cap : 1, 32, (hipStream_t)0);
    if (e != hipSuccess) return 0;
    e = sort_impl<rocprim::default_config>(nullptr, b16, h, h2, v, v2, cap > 0 ? cap : 1, 16, (hipStream_t)0);
    if (e != hipSuccess) return 0;
    CountIter it(rocprim::counting_iterator<uint32_t>(0u), PermutedCount{k, k, 0u});
    e = rocprim::exclusive_scan(nullptr, c, it, k, 0u, (size_t)(n + 1), rocprim::plus<uint32_t>(), (hipStream_t)0);
    if (e != hipSuccess) return 0;
    size_t m = a > b ? a : b;
    m = m > b16 ? m : b16;
    m = m > c ? m : c;
    return (m + 255) & ~(size_t)255;
}

hipError_t sort_by_depth(void *temp, size_t temp_bytes, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                         uint32_t *&vals_alt, int64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return sort_impl<DepthSortConfig>(temp, temp_bytes, keys_cur, keys_alt, vals_cur, vals_alt, n, 32, s);
}

hipError_t scan_counts(void *temp, size_t temp_bytes, const uint32_t *counts, const uint32_t *order,
                       uint32_t *offsets, int64_t n, hipStream_t s) {
    CountIter it(rocprim::counting_iterator<uint32_t>(0u), PermutedCount{counts, order, (uint32_t)n});
    return rocprim::exclusive_scan(temp, temp_bytes, it, offsets, 0u, (size_t)(n + 1), rocprim::plus<uint32_t>(), s);
}

// Emit (tile id, Gaussian index) pairs in rank order, sort them stably by tile id and derive
// every tile's [first, last) range.  Tile ids fit 16 bits for any frame up to 65535 tiles (4K has
// 32 026), which halves the key traffic of the sort; larger frames use 32-bit ids.
// keys0 / keys1 / vals0 / vals1 each hold `d` 32-bit words.  *sorted_vals = the sorted values.
// padded: `d` is the caller's hint, not the true count (which only the device knows): the key
// buffer is pre-filled with all-ones keys that sort behind every tile, and pairs past `d` are dropped.
template <typename Key>
hipError_t bin_impl(void *temp, size_t temp_bytes, const TileRect *rect, const uint32_t *order,
                    const uint32_t *offsets, int64_t n, int64_t d, bool padded, const TileGrid &grid, void *keys0,
                    void *keys1, uint32_t *vals0, uint32_t *vals1, uint2 *ranges, int key_bits,
                    const uint32_t **sorted_vals, hipStream_t s) {
    Key *kc = (Key *)keys0, *ka = (Key *)keys1;
    uint32_t *vc = vals0, *va = vals1;
    hipError_t e;
    if (padded) {
        e = hipMemsetAsync(kc, 0xFF, sizeof(Key) * (size_t)d, s);
        if (e != hipSuccess) return e;
        key_bits = (int)sizeof(Key) * 8;
    }
    emit_kernel<Key><<<blocks_for(n), kBlock, 0, s>>>(rect, order, offsets, n, grid, (uint32_t)d, kc, vc);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = sort_impl<rocprim::default_config>(temp, temp_bytes, kc, ka, vc, va, d, key_bits, s);
    if (e != hipSuccess) return e;
    if (padded)
        tile_ranges_kernel<Key, true><<<blocks_for(d), kBlock, 0, s>>>(kc, d, ranges);
    else
        tile_ranges_kernel<Key, false><<<blocks_for(d), kBlock, 0, s>>>(kc, d, ranges);
    *sorted_vals = vc;
    return hipGetLastError();
}

hipError_t bin_instances(void *temp, size_t temp_bytes, const TileRect *rect, const uint32_t *order,
                         const uint32_t *offsets, int64_t n, int64_t d, bool padded, const TileGrid &grid,
                         void *keys0, void *keys1, uint32_t *vals0, uint32_t *vals1, uint2 *ranges,
                         const uint32_t **sorted_vals, hipStream_t s) {
    const int64_t nt = grid.count();
    hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)nt, s);
    *sorted_vals = vals0;
    if (e != hipSuccess || d == 0 || n == 0) return e;
    int bits = 1;
    while (((int64_t)1 << bits) < nt) ++bits;
    if (nt <= 65535)
        return bin_impl<uint16_t>(temp, temp_bytes, rect, order, offsets, n, d, padded, grid, keys0, keys1, vals0,
                                  vals1, ranges, bits, sorted_vals, s);
    return bin_impl<uint32_t>(temp, temp_bytes, rect, order, offsets, n, d, padded, grid, keys0, keys1, vals0, vals1,
                              ranges, bits, sorted_vals, s);
}

hipError_t publish_counts(const uint32_t *n_visible, const uint32_t *total, int64_t n_visible_known, int64_t *out2,
                          hipStream_t s) {
    publish_counts_kernel<<<1, 1, 0, s>>>(n_visible, total, n_visible_known, out2);
    return hipGetLastError();
}

}  // namespace gsx
