// Test hooks of libgsx_test.so (declared in gsx_debug.h): the pipeline's own sorts on caller-provided data, so that
// tests can hold them against numpy / torch sorts at sizes and key layouts no rendered frame would produce.
// This file is NOT linked into libgsx.so.
#include "gsx_debug.h"
#include "gsx_internal.h"

using namespace gsx;
using namespace gsx::plan;

#define GSX_DBG_HIP(expr)                          \
    do {                                           \
        if ((expr) != hipSuccess) return GSX_ERR_HIP; \
    } while (0)

namespace gsx { hipError_t set_blend_probe(void *device_buffer); }

extern "C" {

int gsx_debug_set_blend_probe(void *device_buffer) {
    return gsx::set_blend_probe(device_buffer) == hipSuccess ? GSX_OK : GSX_ERR_HIP;
}

int gsx_debug_sort_pairs(void *keys, uint32_t *vals, int64_t n, int32_t key_bits, int32_t key16,
                         const uint32_t *count_dev, void *scratch, size_t scratch_bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0 || !keys || !vals || !scratch) return GSX_ERR_INVALID_ARGUMENT;
    const size_t need = align_up((size_t)n * 4) * 2 + radix_temp_bytes(n);
    if (scratch_bytes < need) return GSX_ERR_WORKSPACE_TOO_SMALL;
    char *sc = (char *)scratch;
    uint32_t *valt = (uint32_t *)(sc + align_up((size_t)n * 4));
    void *temp = sc + 2 * align_up((size_t)n * 4);
    uint32_t *vc = vals, *va = valt;
    if (key16) {
        uint16_t *kc = (uint16_t *)keys, *ka = (uint16_t *)sc;
        GSX_DBG_HIP(gsx::radix_sort_pairs_u16(temp, kc, ka, vc, va, count_dev, n, key_bits, s));
        if (kc != (uint16_t *)keys) GSX_DBG_HIP(hipMemcpyAsync(keys, kc, (size_t)n * 2, hipMemcpyDeviceToDevice, s));
    } else {
        uint32_t *kc = (uint32_t *)keys, *ka = (uint32_t *)sc;
        GSX_DBG_HIP(gsx::radix_sort_pairs_u32(temp, kc, ka, vc, va, count_dev, n, key_bits, s));
        if (kc != (uint32_t *)keys) GSX_DBG_HIP(hipMemcpyAsync(keys, kc, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    }
    if (vc != vals) GSX_DBG_HIP(hipMemcpyAsync(vals, vc, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return GSX_OK;
}

int gsx_debug_depth_sort(uint32_t *keys, int64_t n, const void *rect, void *rrect, uint32_t *order_out, int32_t mode,
                         uint32_t lds_cap, int64_t kept_hint, int64_t *counts_host, void *scratch, size_t scratch_bytes,
                         void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0 || !keys || !rect || !rrect || !order_out || !scratch) return GSX_ERR_INVALID_ARGUMENT;
    if (mode != -1 && mode != 0 && mode != 1 && mode != 2 && mode != 4) return GSX_ERR_INVALID_ARGUMENT;
    const size_t words = align_up((size_t)n * 4);
    const size_t need = 3 * words + 256 + binning_temp_bytes(n, 1) + (mode == 4 ? 2 * words : 0);
    if (scratch_bytes < need) return GSX_ERR_WORKSPACE_TOO_SMALL;
    char *sc = (char *)scratch;
    uint32_t *k1 = (uint32_t *)sc, *v0 = (uint32_t *)(sc + words), *v1 = (uint32_t *)(sc + 2 * words);
    uint32_t *counters = (uint32_t *)(sc + 3 * words);
    void *temp = sc + 3 * words + 256;
    GSX_DBG_HIP(hipMemsetAsync(counters, 0, 64, s));
    gsx::DepthRoute route = mode < 0 ? gsx::depth_sort_route(n, kept_hint)
                                     : (mode == 0 || mode == 4 ? gsx::kDepthLsd : (mode == 2 ? gsx::kDepth1024 : gsx::kDepth256));
    // mode 4: the LSD passes with the rectangles carried along (coordinates below 256), in two more arrays at the end
    uint32_t *carry0 = mode == 4 ? (uint32_t *)(sc + need - 2 * words) : nullptr;
    uint32_t *carry1 = mode == 4 ? (uint32_t *)(sc + need - words) : nullptr;
    if (mode > 0 && n <= 16384 && lds_cap == 0) route = gsx::kDepthOneWorkgroup;
    if (route != gsx::kDepthLsd)
        GSX_DBG_HIP(gsx::sort_depth_sampled(route, temp, keys, k1, v0, v1, n, kept_hint, counters + kCtrKept,
                                            counters + kCtrCulled, (const gsx::TileRect *)rect, (gsx::TileRect *)rrect, lds_cap,
                                            nullptr, gsx::SortHints{nullptr, nullptr, nullptr, false}, s));
    else
        GSX_DBG_HIP(gsx::sort_depth_compact(temp, keys, k1, v0, v1, n, counters + kCtrKept, counters + kCtrCulled,
                                            (const gsx::TileRect *)rect, (gsx::TileRect *)rrect, s, nullptr, carry0, carry1));
    uint32_t host[4] = {0, 0, 0, 0};
    GSX_DBG_HIP(hipMemcpyAsync(host, counters, 16, hipMemcpyDeviceToHost, s));
    GSX_DBG_HIP(hipMemcpyAsync(order_out, v0, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    GSX_DBG_HIP(hipStreamSynchronize(s));
    if (counts_host) {
        counts_host[0] = host[kCtrKept];
        counts_host[1] = host[kCtrCulled];
        counts_host[2] = (int64_t)route;
    }
    return GSX_OK;
}

}  // extern "C"
