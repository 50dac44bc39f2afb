// Depth ordering and tile binning on gfx950.
//
// The reference sorts all visible Gaussians once by view depth (splat/gaussian_scene.py:117) and
// then, for every tile, boolean-masks that sorted list (:209-218), so each tile's list is in
// global depth order.  Here the same lists are produced with two stable radix sorts:
//   1. N keys  (depth bits, value = original index)  -> depth rank of every Gaussian;
//   2. D keys  (window-local tile id, value = depth rank), emitted in rank order, so a STABLE
//      sort on the tile id alone (13 bits at 1080p, 2 radix passes instead of 6 for a 64-bit
//      tile|depth key) leaves every tile's entries in depth order, ties broken by original
//      index exactly like a stable argsort.
// rocPRIM provides the device-wide radix sort and scan primitives; the emit / range kernels
// are ours.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/types/double_buffer.hpp>

#include "gsx_internal.h"

namespace gsx {
namespace {

constexpr int kBlock = 256;

// One thread per depth rank writes that Gaussian's (tile id, rank) pairs at its scan offset.
// A Gaussian covering more than kSerialMax tiles is spread over the whole wave instead, so one
// huge splat does not serialise 63 idle lanes behind it.
constexpr uint32_t kSerialMax = 16;

__device__ __forceinline__ void emit_one(const TileRect &r, uint32_t k, const TileGrid &g, uint32_t rank,
                                         uint32_t base, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t h = (uint32_t)(r.y1 - r.y0 + 1);
    uint32_t tx = r.x0 + k / h, ty = r.y0 + k % h;
    keys[base + k] = (tx - (uint32_t)g.wx0) * (uint32_t)g.nwy() + (ty - (uint32_t)g.wy0);
    vals[base + k] = rank;
}

__global__ void __launch_bounds__(kBlock)
    emit_kernel(const TileRect *__restrict__ rect, const uint32_t *__restrict__ offsets, int64_t n, TileGrid g,
                uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    TileRect tr;
    tr.x0 = 1; tr.x1 = 0; tr.y0 = 1; tr.y1 = 0;
    uint32_t base = 0, cnt = 0;
    if (r < n) {
        tr = rect[r];
        base = offsets[r];
        cnt = offsets[r + 1] - base;
    }
    if (cnt <= kSerialMax)
        for (uint32_t k = 0; k < cnt; ++k) emit_one(tr, k, g, (uint32_t)r, base, keys, vals);
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
        uint32_t brank = (uint32_t)(r - lane + src);
        for (uint32_t k = lane; k < bcnt; k += 64) emit_one(br, k, g, brank, bbase, keys, vals);
    }
}

// ranges[t] = [first, last+1) of tile t inside the tile-sorted pair list; untouched (zeroed by
// the caller) for tiles with no entries.
__global__ void __launch_bounds__(kBlock)
    tile_ranges_kernel(const uint32_t *__restrict__ keys, int64_t d, uint2 *__restrict__ ranges) {
    int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= d) return;
    uint32_t t = keys[j];
    if (j == 0 || keys[j - 1] != t) ranges[t].x = (uint32_t)j;
    if (j == d - 1 || keys[j + 1] != t) ranges[t].y = (uint32_t)(j + 1);
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

size_t binning_temp_bytes(int64_t n, int64_t cap) {
    size_t a = 0, b = 0, c = 0;
    uint32_t *k = nullptr;
    rocprim::double_buffer<uint32_t> kb(k, k), vb(k, k);
    hipError_t e;
    e = rocprim::radix_sort_pairs(nullptr, a, kb, vb, (size_t)(n > 0 ? n : 1), 0u, 32u, (hipStream_t)0);
    if (e != hipSuccess) return 0;
    e = rocprim::radix_sort_pairs(nullptr, b, kb, vb, (size_t)(cap > 0 ? cap : 1), 0u, 32u, (hipStream_t)0);
    if (e != hipSuccess) return 0;
    e = rocprim::exclusive_scan(nullptr, c, k, k, 0u, (size_t)(n + 1), rocprim::plus<uint32_t>(), (hipStream_t)0);
    if (e != hipSuccess) return 0;
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    return (m + 255) & ~(size_t)255;
}

hipError_t sort_pairs(void *temp, size_t temp_bytes, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                      uint32_t *&vals_alt, int64_t n, int end_bit, hipStream_t s) {
    if (n == 0) return hipSuccess;
    rocprim::double_buffer<uint32_t> kb(keys_cur, keys_alt), vb(vals_cur, vals_alt);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, kb, vb, (size_t)n, 0u, (unsigned)end_bit, s);
    keys_cur = kb.current();
    keys_alt = kb.alternate();
    vals_cur = vb.current();
    vals_alt = vb.alternate();
    return e;
}

hipError_t scan_counts(void *temp, size_t temp_bytes, const uint32_t *counts, uint32_t *offsets, int64_t n_plus_1,
                       hipStream_t s) {
    return rocprim::exclusive_scan(temp, temp_bytes, counts, offsets, 0u, (size_t)n_plus_1,
                                   rocprim::plus<uint32_t>(), s);
}

hipError_t launch_emit(const TileRect *rect, const uint32_t *offsets, int64_t n, const TileGrid &grid,
                       uint32_t *tile_keys, uint32_t *tile_vals, hipStream_t s) {
    if (n == 0) return hipSuccess;
    emit_kernel<<<blocks_for(n), kBlock, 0, s>>>(rect, offsets, n, grid, tile_keys, tile_vals);
    return hipGetLastError();
}

hipError_t launch_tile_ranges(const uint32_t *sorted_tile_keys, int64_t d, uint2 *ranges, int64_t n_tiles,
                              hipStream_t s) {
    hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)n_tiles, s);
    if (e != hipSuccess || d == 0) return e;
    tile_ranges_kernel<<<blocks_for(d), kBlock, 0, s>>>(sorted_tile_keys, d, ranges);
    return hipGetLastError();
}

}  // namespace gsx
