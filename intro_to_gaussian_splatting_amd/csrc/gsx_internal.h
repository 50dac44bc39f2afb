// Internal declarations shared by the translation units of libgsx.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gsx.h"
#include "gsx_plan.h"

namespace gsx {

// Measurement knobs (A/B runs of a kernel variant on the GPU box: tools/, bench.py --test-lib) exist only in
// libgsx_test.so, which is this same source built with -DGSX_TEST_HOOKS; in the shipping library knob() is its
// default and no environment variable is read anywhere.
inline int knob(const char *name, int dflt) {
#ifdef GSX_TEST_HOOKS
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

constexpr uint32_t kCulledKey = 0xFFFFFFFFu;  // depth key of a Gaussian behind the z >= 0.2 plane
constexpr uint32_t kEmptyKey = 0xFFFFFFFEu;   // visible, but it reaches no tile of the window; keys >= this are
                                              // dropped by the first pass of the depth sort

// Stage-1 -> stage-2 record, 48 B, three 16-B loads, indexed by the Gaussian's index (original
// index on the whole-path entry, row index on the stage-2 entry).  With Q'' = Q * (-1/2 log2 e):
//   GSX_SEM_REF_CPU (square completed in y, M = -Q'': r11 = sqrt(M11), h = M01 / r11, D1 = M00 - h^2):
//     a = (x_pix, y_pix, D1, h)   b = (r11, log2(opacity factor), r, g)   c = (b, depth, 0, -)
//     or, when that factorisation does not exist, the monomial form below with c.z = 1
//     or, for an ILL-CONDITIONED footprint (pack_record in gsx_project.hip), the same completed square with c = (b,
//     opacity factor, 2, -) and the raw float32 conic (Q00, Q01, Q10, Q11) in the Gaussian's slot of the per-Gaussian
//     float4 side array (the workspace's `bbox`): the compositing kernels execute the reference's own operation order
//     on such a record wherever its rounding can show (gsx_blend.hip: kKindRefOrder, blend_tile16_ref_kernel)
//   other semantics:
//     a = (x_pix, y_pix, Q''00, Q''01 + Q''10)   b = (Q''11, log2(opacity factor) | opacity, r, g)   c = (b, depth, 0, -)
struct __attribute__((aligned(16))) Record {
    float4 a, b, c;
};
static_assert(sizeof(Record) == kRecordBytes, "gsx_plan.h sizes the workspace with this");

// Tile rectangle of one Gaussian, inclusive, already clamped to the window; empty when x0 > x1.
struct TileRect {
    uint16_t x0, x1, y0, y1;
};
static_assert(sizeof(TileRect) == kTileRectBytes && sizeof(float4) == kBboxBytes && sizeof(uint2) == kRangeBytes,
              "gsx_plan.h sizes the workspace with these");

// Tiles whose list is much longer than the frame's average are composited by FOUR waves (a quarter of the
// tile's pixels each, one pixel per lane, eight records per trip) instead of one: a lone wave walks its list
// at ~430 cycles per record, so one 20 000-entry tile would outlast the rest of the frame several times over.
// The list of such tiles is built on the device by tile_ranges_kernel (count: zeroed by the emit kernel) and
// flagged in bit 31 of ranges[t].y; the compositing launch carries 4 * kMaxLongTiles helper workgroups.
constexpr uint32_t kLongFlag = 0x80000000u;
struct LongTiles {
    uint32_t *count;   // number of long tiles found (may exceed max)
    uint32_t *list;    // window-local tile ids, max entries
    uint32_t max;      // 0: the split is off for this frame
    // When the previous frame of the view left its tiles' costs (GsxParams.hints; header[kHintLens] and
    // header[kHintSched] name this window) a tile is long when it COST more than cost_pct % of a SIMD's share of the
    // frame (sum of all costs / 1024) -- see tile_ranges_kernel; cost == nullptr: by list length alone.
    uint32_t *cost = nullptr;
    const uint32_t *header = nullptr;
    uint32_t cost_pct = 30;
    // A tile that WAS long stays long down to stay_pct % of the threshold: four quarters (each walks its own block's list)
    // and one wave (the longest of four block lists per batch) do not report the same cost, and a tile within that
    // difference of the threshold changed sides on every frame -- nine tiles of the heavy-tailed test scene, whose frames
    // alternated between 0.476 and 0.556 ms (round 6, tools/frame_sequence.py).
    uint32_t stay_pct = 75;
    // one word, zeroed by the emit kernel: tiles and long-tile quarters of the tile-16 REF_CPU compositing launch that met a
    // reference-order record (GsxFrameStats.n_redo; with GSX_FLAG_PLAIN_FOOTPRINTS: that were left undone for it)
    uint32_t *redo = nullptr;
};
__host__ __device__ inline uint32_t long_tile_threshold(uint32_t pairs, uint32_t tiles) {
    const uint32_t mean = tiles ? pairs / tiles : 0u;
    const uint32_t four = mean > 0x3FFFFFFFu ? 0xFFFFFFFFu : 4u * mean;
    return four > 1024u ? four : 1024u;
}

struct StageOneOut {  // PreprocessedScene arrays (splat/schema.py:13-25), depth-sorted
    float *points_xy, *colors, *cov2d, *depths, *inv_cov, *radius, *min_x, *max_x, *min_y, *max_y, *sig_op;
    int32_t *order;
};

struct GaussiansIn {  // splat/gaussians.py:19-33
    const float *means3d, *scales, *quats, *opacity_logit, *colors;
    // GsxParams.original_index (or null): row i holds the Gaussian of that ORIGINAL index -- keys, records and rectangles
    // are filed under it (launch_project_pack pre-fills the keys with kEmptyKey first)
    const int32_t *original_index = nullptr;
    const float4 *block_bounds = nullptr;      // GsxParams.block_bounds (two float4 per GSX_BOUNDS_ROWS rows), or null
};

struct PreprocessedIn {  // argument list of splat/c/render.cu:90-101
    const float *means, *colors, *inv_cov, *min_x, *max_x, *min_y, *max_y, *opacity;
};

// The projection launch's share: one spare workgroup puts the compositing schedule of THIS frame together from the
// tiles' costs the previous frame left (gsx_schedule_device.h; sched == nullptr: not wanted).
struct ScheduleHint {
    const uint32_t *lens;
    uint32_t *sched, *header;
    uint32_t ntiles, nwy;       // tiles of the window, and along its y axis (tile ids are column-major)
};

// ---- gsx_project.hip (compiled with -ffp-contract=off)
// gsx_preprocess, first kernel: depth keys in original order (kCulledKey behind the cull plane), the 11 floats of
// every visible Gaussian that the rank-ordered output kernel gathers (in its record slot), the sort's counters zeroed.
// visible_rows: 0 = as many of the n Gaussians are visible as n suggests (the default), 1 = GSX_FLAG_ONE_VISIBLE,
// 2 = GSX_FLAG_SMALL_BATCH (two or three): which of its BLAS's orders the reference's products over the VISIBLE rows take.
hipError_t launch_project_stage(const GsxCamera &cam, const GaussiansIn &in, int64_t n, uint32_t *keys, Record *stage,
                                uint32_t *counters, int visible_rows, hipStream_t s);
// Original order: depth keys (kCulledKey behind the cull plane, kEmptyKey when no tile of the window is
// reached), records / rects indexed by the ORIGINAL Gaussian index (only written for the Gaussians that
// reach a tile; the depth sort generates the identity values itself).
// bbox (REF_CUDA only, else may be null): (min_x, max_x, min_y, max_y) per Gaussian for the
// per-pixel cull of splat/c/render.cu:55-60.
// tight_rects: GSX_SEM_STD_3DGS without GSX_FLAG_PUBLISHED_RECTS (see include/gsx.h).
// cam_device (may be null): GsxParams.camera_device, read by the kernel instead of `cam`.
// sh_degree >= 0: in.colors holds spherical-harmonics coefficients, evaluated inline (GsxParams.sh).
// counters: 4 words this kernel zeroes for the depth sort (culled count, kept count, ...).
// A frame without splitters from an earlier frame of its view: `wgs` spare workgroups of the projection launch rank a sample
// of `ns` keys they compute themselves (gsx_sample_device.h) into the partition pass' splitters and zero its chunk sums --
// what sample_rank_kernel would do BEHIND the projection.  splitters == nullptr: not wanted.  (depth_presample, below: gsx_sort.hip)
struct SampleHint {
    uint32_t *splitters = nullptr;
    unsigned long long *chunk_sums = nullptr;
    uint32_t nsums = 0, ns = 0;
    const uint32_t *row_of = nullptr;      // GsxParams.row_of_index: sample index (an original index) -> row
};
hipError_t launch_project_pack(const GsxCamera &cam, const GsxCamera *cam_device, const GaussiansIn &in, int64_t n,
                               const TileGrid &grid, int semantics, bool tight_rects, int visible_rows, int sh_degree,
                               uint32_t *keys, Record *rec, TileRect *rect, uint32_t *counters, float4 *bbox,
                               const ScheduleHint &sched, uint8_t *block_scratch, hipStream_t s,
                               const SampleHint &sample = SampleHint());
// (block_scratch: ceil(n / GSX_BOUNDS_ROWS) bytes the launch may use when in.block_bounds is set -- the depth sort's
// temp area, idle until the projection is done; in.original_index: the launch pre-fills the keys itself)
// gsx_preprocess, last kernel: all PreprocessedScene fields in depth order; order[r] = Gaussian of rank r, r < *m_dev.
hipError_t launch_project_full(const Record *stage, const uint32_t *order, const uint32_t *m_dev, int64_t n,
                               const StageOneOut &out, hipStream_t s);
hipError_t launch_pack_preprocessed(const PreprocessedIn &in, int64_t n, const TileGrid &grid, int semantics,
                                    Record *rec, TileRect *rect, float4 *bbox, hipStream_t s);
hipError_t launch_covariance3d(const float *scales, const float *quats, int64_t n, float *out, hipStream_t s);
hipError_t launch_covariance2d(const GsxCamera &cam, const float *points, const float *cov3d, int64_t n, float *out,
                               hipStream_t s);
hipError_t launch_project_points(const GsxCamera &cam, const float *means3d, int64_t n, float *points_out,
                                 uint8_t *in_view, hipStream_t s);

// ---- gsx_binning.hip
// Where a frame's counts go on the device (and, optionally, straight into pinned host memory).
struct BinCounts {
    int64_t *stats2;             // [0] = visible Gaussians, [1] = D, [2] = Gaussians kept by the depth sort
    int64_t *stats2_host;        // device-visible alias of a pinned GsxFrameStats (fields 0, 1 and n_kept), or null
    uint32_t *d32;               // min(D, 2^32 - 1): element count of the tile sort
    uint32_t *long_count;        // zeroed by the emit kernel for tile_ranges_kernel (LongTiles.count)
    uint32_t *redo_count;        // zeroed by the emit kernel for the compositing launch (LongTiles.redo[0])
    const uint32_t *culled_dev;  // Gaussians behind the cull plane (counted by the depth sort), or null
    int64_t n_total;             // n_visible = n_total - *culled_dev
};
// One (tile id, Gaussian index) pair per covered tile, in rank order, from the rank-ordered tile
// rectangles rrect[0 .. m) (m = min(*m_dev, n); m_dev == nullptr: n).  order[r] = Gaussian index of rank
// r (nullptr: the identity).  Also zeroes ranges[] and delivers the frame's counts (bc).  keys0 / vals0
// hold cap 32-bit words each; pairs beyond cap are dropped.  sums_ready: emit_chunk_sums(temp, n, cap) already
// holds the chunk sums (the sampled depth sort left them there).
hipError_t emit_instances(void *temp, const TileRect *rrect, const uint32_t *order, const uint32_t *m_dev, int64_t n,
                          int64_t cap, const TileGrid &grid, void *keys0, uint32_t *vals0, uint2 *ranges,
                          const BinCounts &bc, bool sums_ready, int64_t kept_hint, hipStream_t s);
// Stable sort of the emitted pairs by tile id and ranges[t] = [first, last) for every tile of the
// window; *sorted_vals points at the sorted Gaussian indices.  The pair count is read from *d32.
hipError_t sort_instances(void *temp, int64_t cap, const TileGrid &grid, void *keys0, void *keys1, uint32_t *vals0,
                          uint32_t *vals1, uint2 *ranges, const uint32_t *d32, const LongTiles &lt,
                          const uint32_t **sorted_vals, hipStream_t s);
// counts[t] = length of tile t's list (GsxParams.tile_counts).
hipError_t launch_tile_counts(const uint2 *ranges, int64_t nt, uint32_t *counts, hipStream_t s);
// sched[k] = tile with the k-th longest list (1024 length classes), for launch_blend's balanced hand-out.
hipError_t launch_tile_schedule(const uint2 *ranges, int64_t nt, uint32_t *sched, hipStream_t s);
// the per-XCD schedule of THIS frame's list lengths (REF_CPU tile 16; header: kHintHeaderWords words, sched: hints_sched_entries)
hipError_t launch_tile_schedule_xcd(const uint2 *ranges, int64_t nt, int64_t nwy, uint32_t *sched, uint32_t *header, hipStream_t s);
// ---- gsx_sort.hip: stable LSD radix sort, up to 8 bits per pass, key bits [0, key_bits).  The element
// count is min(*n_dev, bound) (n_dev == nullptr: bound); grids are sized by `bound`.  Buffers
// ping-pong; on return keys_cur / vals_cur point at the sorted data.  temp: radix_temp_bytes(bound).
hipError_t radix_sort_pairs_u32(void *temp, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                                uint32_t *&vals_alt, const uint32_t *n_dev, int64_t bound, int key_bits,
                                hipStream_t s);
hipError_t radix_sort_pairs_u16(void *temp, uint16_t *&keys_cur, uint16_t *&keys_alt, uint32_t *&vals_cur,
                                uint32_t *&vals_alt, const uint32_t *n_dev, int64_t bound, int key_bits,
                                hipStream_t s);
// Depth sort of the whole-path entry point: all 32 key bits, dropping keys >= kEmptyKey in the first
// pass (*m_dev = Gaussians kept, *culled_dev += keys == kCulledKey; culled_dev must hold 0 when the
// first kernel runs) and gathering rect[index] into rrect[rank] in the last one.  On return
// vals_cur[0 .. *m_dev) = Gaussian index of each depth rank (ties: original index).  The values of the
// first pass are the item positions themselves (vals_cur need not be initialised).
// row_of (both depth sorts; or null): GsxParams.row_of_index -- the keys are filed under ORIGINAL indices, the value an item
// gets in the first pass is the ROW that index names (records and rectangles are in row order)
hipError_t sort_depth_compact(void *temp, uint32_t *keys0, uint32_t *keys1, uint32_t *&vals_cur, uint32_t *&vals_alt,
                              int64_t n, uint32_t *m_dev, uint32_t *culled_dev, const TileRect *rect, TileRect *rrect,
                              hipStream_t s, uint32_t *samples_out = nullptr, uint32_t *carry0 = nullptr,
                              uint32_t *carry1 = nullptr, const uint32_t *row_of = nullptr);

// The same contract with 5 kernels instead of 12: sample 2048 / 8192 keys -> 255 / 1023 splitters, ONE stable
// partition pass, one in-LDS sort per bucket (gsx_sort.hip).  depth_sort_route picks the route from the
// Gaussian count and the caller's hint of how many of them reach a tile (GsxParams.kept_hint; 0 = unknown).
// lds_cap: 0 = the bucket kernel's capacity; tests pass a small value to drive buckets through its global-memory path.
enum DepthRoute { kDepthLsd = 0, kDepth256 = 1, kDepth1024 = 2, kDepthOneWorkgroup = 3 };
DepthRoute depth_sort_route(int64_t n, int64_t kept_hint);
SampleHint depth_presample(DepthRoute route, void *temp, int64_t n, uint64_t *chunk_sums, const uint32_t *row_of);
// The depth sort's share of GsxParams.hints (gsx_plan.h: hints_layout); header == nullptr: no hints buffer.
struct SortHints {
    uint32_t *header;           // kHintSplitters says whether `splitters` are there, kHintLens / kHintSched how many tiles
    const uint32_t *splitters;  // 256 words left by the previous frame
    uint32_t *samples;          // kSortSamples words this frame's count kernel fills for the next frame
    bool use;                   // GSX_FLAG_HINTS_VALID: partition with `splitters`, no sample kernel
    // the 256-bucket route (round 6): the bucket kernel knows every kept key's rank and leaves the NEXT frame's splitters
    // itself -- the EXACT 256-quantiles of this frame's keys, in place of the hints' splitters (every reader of the current
    // ones has finished) -- instead of a sample for the compositing launch's spare workgroups to rank; null: not wanted
    uint32_t *next_splitters = nullptr;
};

// The compositing launch's share: spare workgroups rank the samples into the next frame's splitters, every tile
// workgroup leaves the length of its list.  header == nullptr: no hints buffer.
struct BlendHints {
    uint32_t *header;
    const uint32_t *samples;    // kSortSamples words of this frame's count kernel, or nullptr (nothing to rank)
    uint32_t *splitters;
    uint32_t *lens;
    uint32_t xcd_sched;         // != 0: `sched` is hints.sched, the per-XCD schedule (gsx_schedule_device.h) -- trusted
                                // only if header[kHintSched] == number of tiles; 0: tile_schedule_kernel's whole-frame order
    uint32_t rank_last = 0;     // the spare workgroups that rank the samples come last in the grid instead of first
    uint32_t plain = 0;         // GSX_FLAG_PLAIN_FOOTPRINTS: the compositing instance that cannot evaluate reference-order records
};
hipError_t sort_depth_sampled(DepthRoute route, void *temp, uint32_t *keys0, uint32_t *keys1, uint32_t *&vals_cur,
                              uint32_t *&vals_alt, int64_t n, int64_t kept_hint, uint32_t *m_dev, uint32_t *culled_dev,
                              const TileRect *rect, TileRect *rrect, uint32_t lds_cap, uint64_t *chunk_sums,
                              const SortHints &hints, hipStream_t s, const uint32_t *row_of = nullptr, bool presampled = false);
// Where emit_instances keeps its chunk sums inside `temp` (for sort_depth_sampled to fill them in).
uint64_t *emit_chunk_sums(void *temp, int64_t n, int64_t cap);

// ---- gsx_sh.hip
hipError_t launch_sh_to_rgb(const float *means3d, const float *sh, int degree, int64_t n, const float *center,
                            float *colors, hipStream_t s);

// ---- gsx_blend.hip
// background: 3 floats, read on the host (GSX_SEM_STD_3DGS only); generic: GSX_FLAG_GENERIC_KERNELS.
// cp: what the launch zeroes besides compositing its tiles (extra workgroups of the same kernel).
// lt: long tiles split over four waves (GSX_SEM_REF_CPU, tile 16 only; lt.max == 0 otherwise).
// sched: launch_tile_schedule's order, or nullptr (tiles in index order); used by the tile-16 REF_CPU kernel.
// asked: Plan.schedule (1 / 0 from the caller's flags, -1: by the size of the scene).
bool blend_uses_schedule(const TileGrid &grid, int semantics, bool generic, int64_t n, int asked);
// One part of the window (GsxParams.n_substrips): the tiles whose column (axis 0) / row (axis 1) -- absolute tile
// coordinate -- lies in [lo, hi); first: this launch also does what a frame does once (zero fill, next frame's splitters).
struct TileSpan {
    int32_t axis, lo, hi;
    uint32_t first;
    uint32_t index = 0;         // which part (0 .. 15)
};
// can this kernel family composite a window in parts?  (tile-16 REF_CPU kernel only; the others take one launch)
bool blend_in_parts(const TileGrid &grid, int semantics, bool generic);
// part: nullptr = the whole window in one launch
hipError_t launch_blend(const Record *rec, const float4 *bbox, const uint32_t *sorted_vals, const uint2 *ranges,
                        const TileGrid &grid, const OutDesc &out, int semantics, const float *background,
                        bool generic, const ClearPlan &cp, const LongTiles &lt, const uint32_t *sched,
                        const BlendHints &hints, hipStream_t s, const TileSpan *part = nullptr);
bool blend_splits_long_tiles(const TileGrid &grid, int semantics, bool generic);
hipError_t launch_clear(const ClearPlan &cp, float *base, hipStream_t s);   // the zero fill alone
hipError_t launch_zero_words(uint32_t *p, size_t n, hipStream_t s);

}  // namespace gsx
