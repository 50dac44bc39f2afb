/*
 * gsx_debug.h -- test hooks of libgsx_test.so (the library sources built with -DGSX_TEST_HOOKS plus gsx_debug.hip).
 * NOT part of the C ABI: the shipping libgsx.so neither declares nor exports these, and reads no environment
 * variable.  tests/ and tools/ bind them with ctypes (intro_to_gaussian_splatting_amd/_ffi.py: load_test_hooks()).
 */
#ifndef GSX_DEBUG_H_
#define GSX_DEBUG_H_

#include "gsx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The pipeline's radix sort on caller-provided pairs.  keys / vals: n 32-bit words each, sorted in place;
 * key16 != 0 sorts uint16 keys.  scratch must hold 2 * 4n bytes + the radix table (use
 * gsx_workspace_bytes(n, 16, 16, 16, n)).  count_dev (may be NULL) = device pointer to the element count, as
 * the tile sort uses it. */
GSX_API int gsx_debug_sort_pairs(void *keys, uint32_t *vals, int64_t n, int32_t key_bits, int32_t key16,
                                 const uint32_t *count_dev, void *scratch, size_t scratch_bytes, void *stream);

/* The depth sort of the whole-path entry on caller-provided keys.  keys (n, device; >= 0xFFFFFFFE = dropped;
 * overwritten), rect / rrect (n x 4 uint16), order_out (n): on return order_out[0 .. counts_host[0]) = index of
 * each rank, rrect[rank] = rect[index]; counts_host (3 entries) = {kept, culled, route taken}.
 * mode 0: four LSD passes (the last one gathers rect[index]); 1: sample-partitioned, 256 buckets; 2: sample-partitioned,
 * 1024 buckets (a route nothing in gsx_render_forward selects: measured slower, kept as a tested alternative);
 * 4: the LSD passes with the rectangles CARRIED along, packed into 4 bytes -- every coordinate of `rect` must then be
 * below 256 (GSX_ERR_INVALID_ARGUMENT is NOT detected on the device: the caller guarantees it) and the scratch holds two
 * more arrays; -1: the route gsx_render_forward would take for (n, kept_hint).  Any other mode:
 * GSX_ERR_INVALID_ARGUMENT.  lds_cap: bucket size above which the bucket kernel sorts through global memory (0 = its
 * LDS capacity).  scratch: 3 x align256(4 n) + 256 + gsx_workspace_bytes(n, 16, 16, 16, 1) bytes, + 2 x align256(4 n) for
 * mode 4 (24 n + 4096 + gsx_workspace_bytes(..) covers every mode; too little: GSX_ERR_WORKSPACE_TOO_SMALL).
 * Synchronises. */
GSX_API int gsx_debug_depth_sort(uint32_t *keys, int64_t n, const void *rect, void *rrect, uint32_t *order_out,
                                 int32_t mode, uint32_t lds_cap, int64_t kept_hint, int64_t *counts_host,
                                 void *scratch, size_t scratch_bytes, void *stream);

/* Who is the compositing launch waiting for, and where does it run?  device_buffer: 4 x 2^17 x 16 bytes, or NULL to switch
 * the probe off.  While set, every workgroup of the tile-16 REF_CPU compositing kernel stores (cycles it ran, window-
 * local tile id | 1 << 30 for a long tile's helper, length of the tile's list, records staged | saturated << 31) at
 * index blockIdx.x and (batches staged | entries walked under the exact rule << 12, wall clock at its end [10 ns], HW_ID & 0xFFFF | XCC_ID << 16,
 * wall clock at its start) at index 2^17 + blockIdx.x (tools/attic/blend_probe.py, tools/attic/simd_balance.py); the
 * instance that evaluates reference-order records (blend_tile16_ref_kernel) stores (cycles, batches | first batch with a reference-order record << 12 | saturated
 * << 31, such batches | such records << 12, list length) per tile at 2^18 + tile (tools/attic/ref_probe.py). */
GSX_API int gsx_debug_set_blend_probe(void *device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* GSX_DEBUG_H_ */
