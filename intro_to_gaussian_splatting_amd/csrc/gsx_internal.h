// Internal declarations shared by the translation units of libgsx.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsx.h"

namespace gsx {

constexpr uint32_t kCulledKey = 0xFFFFFFFFu;  // depth key of a Gaussian behind the z >= 0.2 plane

// Stage-1 -> stage-2 record, 48 B, three 16-B loads, indexed by the Gaussian's index (original
// index on the whole-path entry, row index on the stage-2 entry).  With Q'' = Q * (-1/2 log2 e):
//   a = (x_pix, y_pix, Q''00, Q''01 + Q''10)   b = (Q''11, log2(opacity factor), r, g)   c = (b, depth, -, -)
struct __attribute__((aligned(16))) Record {
    float4 a, b, c;
};

// Which tiles exist and which of them this call renders.
//   ntx, nty : number of tiles in the frame along x / y (REF_CPU: the last row/column is absent)
//   wx0..wy1 : window [wx0,wx1) x [wy0,wy1) rendered by this call
// Window-local tile id = (tx - wx0) * (wy1 - wy0) + (ty - wy0).
struct TileGrid {
    int32_t tile, ntx, nty;
    int32_t wx0, wx1, wy0, wy1;
    int32_t width, height;  // frame size in pixels (REF_CUDA has partial edge tiles)
    __host__ __device__ int32_t nwx() const { return wx1 - wx0; }
    __host__ __device__ int32_t nwy() const { return wy1 - wy0; }
    __host__ __device__ int64_t count() const { return (int64_t)nwx() * nwy(); }
};

// Tile rectangle of one Gaussian, inclusive, already clamped to the window; empty when x0 > x1.
struct TileRect {
    uint16_t x0, x1, y0, y1;
};

struct OutDesc {
    float *ptr;
    int64_t stride_x, stride_y;  // in floats; the channel stride is 1
    int32_t x0, y0, w, h;        // frame pixel of out(0,0) and the buffer extent in pixels
};

struct StageOneOut {  // PreprocessedScene arrays (splat/schema.py:13-25), depth-sorted
    float *points_xy, *colors, *cov2d, *depths, *inv_cov, *radius, *min_x, *max_x, *min_y, *max_y, *sig_op;
    int32_t *order;
};

struct GaussiansIn {  // splat/gaussians.py:19-33
    const float *means3d, *scales, *quats, *opacity_logit, *colors;
};

struct PreprocessedIn {  // argument list of splat/c/render.cu:90-101
    const float *means, *colors, *inv_cov, *min_x, *max_x, *min_y, *max_y, *opacity;
};

// ---- gsx_project.hip (compiled with -ffp-contract=off)
hipError_t launch_depth_keys(const GsxCamera &cam, const float *means3d, int64_t n, uint32_t *keys,
                             uint32_t *vals, hipStream_t s);
hipError_t launch_count_visible(const uint32_t *sorted_keys, int64_t n, uint32_t *n_visible, hipStream_t s);
// Original order: depth keys + identity values for the sort, records / rects / counts indexed by
// the ORIGINAL Gaussian index.
// bbox (REF_CUDA only, else may be null): (min_x, max_x, min_y, max_y) per Gaussian for the
// per-pixel cull of splat/c/render.cu:55-60.
// tight_rects: GSX_SEM_STD_3DGS without GSX_FLAG_PUBLISHED_RECTS (see include/gsx.h).
// cam_device (may be null): GsxParams.camera_device, read by the kernel instead of `cam`.
hipError_t launch_project_pack(const GsxCamera &cam, const GsxCamera *cam_device, const GaussiansIn &in, int64_t n,
                               const TileGrid &grid, int semantics, bool tight_rects, uint32_t *keys, uint32_t *vals,
                               Record *rec, TileRect *rect, uint32_t *counts, float4 *bbox, hipStream_t s);
hipError_t launch_project_full(const GsxCamera &cam, const GaussiansIn &in, const uint32_t *sorted_keys,
                               const uint32_t *sorted_idx, int64_t n, const StageOneOut &out, hipStream_t s);
hipError_t launch_pack_preprocessed(const PreprocessedIn &in, int64_t n, const TileGrid &grid, int semantics,
                                    Record *rec, TileRect *rect, uint32_t *counts, float4 *bbox, hipStream_t s);
hipError_t launch_covariance3d(const float *scales, const float *quats, int64_t n, float *out, hipStream_t s);
hipError_t launch_project_points(const GsxCamera &cam, const float *means3d, int64_t n, float *points_out,
                                 uint8_t *in_view, hipStream_t s);

// ---- gsx_binning.hip
size_t binning_temp_bytes(int64_t n, int64_t cap);
// Stable radix sort of (depth key, index) pairs, all 32 key bits.  The two buffers of each pair
// are ping-ponged; on return keys_cur / vals_cur point at the sorted data.
hipError_t sort_by_depth(void *temp, size_t temp_bytes, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                         uint32_t *&vals_alt, int64_t n, hipStream_t s);
// offsets[r] = sum over ranks r' < r of counts[order[r']] for r in [0, n]; order == nullptr means
// the identity (rows already in compositing order).  Also delivers the frame counts on the device:
// counts2[0] = visible Gaussians (first culled key of sorted_keys, or n_visible_known when >= 0),
// counts2[1] = offsets[n] = D, as int64 (the first two fields of a GsxFrameStats).
hipError_t scan_counts(void *temp, size_t temp_bytes, const uint32_t *counts, const uint32_t *order,
                       const uint32_t *sorted_keys, uint32_t *offsets, int64_t n, uint32_t *n_visible,
                       int64_t n_visible_known, int64_t *counts2, hipStream_t s);
// Emits one (tile id, Gaussian index) pair per covered tile in rank order, stable-sorts them by
// tile id and fills ranges[t] = [first, last) for every tile of the window.  keys0/keys1/vals0/
// vals1 hold cap 32-bit words each; *sorted_vals points at the sorted Gaussian indices.  The pair
// count D = offsets[n] is only read on the device; pairs beyond cap are dropped.
hipError_t bin_instances(void *temp, size_t temp_bytes, const TileRect *rect, const uint32_t *order,
                         const uint32_t *offsets, int64_t n, int64_t cap, const TileGrid &grid, void *keys0,
                         void *keys1, uint32_t *vals0, uint32_t *vals1, uint2 *ranges, const uint32_t **sorted_vals,
                         hipStream_t s);
// ---- gsx_sort.hip: stable LSD radix sort, 8-bit digits, key bits [0, key_bits).  The element
// count is min(*n_dev, bound) (n_dev == nullptr: bound); grids are sized by `bound`.  Buffers
// ping-pong; on return keys_cur / vals_cur point at the sorted data.  temp: radix_temp_bytes(bound).
size_t radix_temp_bytes(int64_t max_items);
hipError_t radix_sort_pairs_u32(void *temp, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                                uint32_t *&vals_alt, const uint32_t *n_dev, int64_t bound, int key_bits,
                                hipStream_t s);
hipError_t radix_sort_pairs_u16(void *temp, uint16_t *&keys_cur, uint16_t *&keys_alt, uint32_t *&vals_cur,
                                uint32_t *&vals_alt, const uint32_t *n_dev, int64_t bound, int key_bits,
                                hipStream_t s);

// ---- gsx_sh.hip
hipError_t launch_sh_to_rgb(const float *means3d, const float *sh, int degree, int64_t n, const float *center,
                            float *colors, hipStream_t s);

// ---- gsx_blend.hip
// background: 3 floats, read on the host (GSX_SEM_STD_3DGS only); generic: GSX_FLAG_GENERIC_KERNELS.
hipError_t launch_blend(const Record *rec, const float4 *bbox, const uint32_t *sorted_vals, const uint2 *ranges,
                        const TileGrid &grid, const OutDesc &out, int semantics, const float *background,
                        bool generic, hipStream_t s);

}  // namespace gsx
