// Internal declarations shared by the translation units of libgsx.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsx.h"

namespace gsx {

constexpr uint32_t kCulledKey = 0xFFFFFFFFu;  // depth key of a Gaussian behind the z >= 0.2 plane

// Stage-1 -> stage-2 record, 48 B, three 16-B loads.  Indexed by depth rank.
//   a = (x_pix, y_pix, Q00, Q01)   b = (Q10, Q11, opacity factor, radius)   c = (r, g, b, depth)
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
                             uint32_t *vals, uint32_t *n_visible, hipStream_t s);
hipError_t launch_project_pack(const GsxCamera &cam, const GaussiansIn &in, const uint32_t *sorted_keys,
                               const uint32_t *sorted_idx, int64_t n, const TileGrid &grid, int semantics,
                               Record *rec, TileRect *rect, uint32_t *counts, hipStream_t s);
hipError_t launch_project_full(const GsxCamera &cam, const GaussiansIn &in, const uint32_t *sorted_keys,
                               const uint32_t *sorted_idx, int64_t n, const StageOneOut &out, hipStream_t s);
hipError_t launch_pack_preprocessed(const PreprocessedIn &in, int64_t n, const TileGrid &grid, int semantics,
                                    Record *rec, TileRect *rect, uint32_t *counts, hipStream_t s);
hipError_t launch_project_points(const GsxCamera &cam, const float *means3d, int64_t n, float *points_out,
                                 uint8_t *in_view, hipStream_t s);

// ---- gsx_binning.hip
size_t binning_temp_bytes(int64_t n, int64_t cap);
// Stable LSD radix sort of (key, value) pairs on key bits [0, end_bit).  The two buffers of
// each pair are ping-ponged; on return keys_cur / vals_cur point at the sorted data.
hipError_t sort_pairs(void *temp, size_t temp_bytes, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                      uint32_t *&vals_alt, int64_t n, int end_bit, hipStream_t s);
hipError_t scan_counts(void *temp, size_t temp_bytes, const uint32_t *counts, uint32_t *offsets, int64_t n_plus_1,
                       hipStream_t s);
hipError_t launch_emit(const TileRect *rect, const uint32_t *offsets, int64_t n, const TileGrid &grid,
                       uint32_t *tile_keys, uint32_t *tile_vals, hipStream_t s);
hipError_t launch_tile_ranges(const uint32_t *sorted_tile_keys, int64_t d, uint2 *ranges, int64_t n_tiles,
                              hipStream_t s);

// ---- gsx_blend.hip
hipError_t launch_blend(const Record *rec, const uint32_t *sorted_vals, const uint2 *ranges, const TileGrid &grid,
                        const OutDesc &out, int semantics, hipStream_t s);

}  // namespace gsx
