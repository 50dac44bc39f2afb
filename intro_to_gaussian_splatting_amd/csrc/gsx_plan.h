// Host-side planning of a frame: tile grid and window, output descriptor, workspace carving and capacity,
// the rectangles a frame zeroes.  Pure integer arithmetic with no HIP type in it, so that the same code is
// (a) what libgsx.so runs and (b) what tests/host/plan_sanitize.cpp compiles with g++ under
// -fsanitize=address,undefined and sweeps to the 2^31 limits (GSX_HD is empty there).
#pragma once

#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gsx.h"

#if defined(__HIPCC__)
#define GSX_HD __host__ __device__
#else
#define GSX_HD
#endif

namespace gsx {

// ---- sizes the kernels and the workspace agree on
constexpr int kSortItems = 2048;       // a radix-sort chunk: items of one scatter workgroup (gsx_sort.hip)
constexpr int kSortQuad = 4;           // chunks per count workgroup
constexpr int kSortBins = 256;         // digit-table rows
constexpr int kEmitChunk = 1024;       // depth ranks per workgroup of chunk_sums / emit (gsx_binning.hip)
constexpr uint32_t kMaxLongTiles = 512;
constexpr int kMaxSubstrips = 16;      // GsxParams.n_substrips
constexpr size_t kParamsBytesAbi300 = 104;   // sizeof(GsxParams) before struct_size existed: what struct_size == 0 means
constexpr int kClearFloats = 6144;     // floats zeroed per clear workgroup (gsx_blend.hip)
constexpr size_t kRecordBytes = 48, kTileRectBytes = 8, kBboxBytes = 16, kRangeBytes = 8;

// Which tiles exist and which of them this call renders.
//   ntx, nty : number of tiles in the frame along x / y (REF_CPU: the last row/column is absent)
//   wx0..wy1 : window [wx0,wx1) x [wy0,wy1) rendered by this call
// Window-local tile id = (tx - wx0) * (wy1 - wy0) + (ty - wy0).
struct TileGrid {
    int32_t tile, ntx, nty;
    int32_t wx0, wx1, wy0, wy1;
    int32_t width, height;  // frame size in pixels (REF_CUDA has partial edge tiles)
    GSX_HD int32_t nwx() const { return wx1 - wx0; }
    GSX_HD int32_t nwy() const { return wy1 - wy0; }
    GSX_HD int64_t count() const { return (int64_t)nwx() * nwy(); }
};

struct OutDesc {
    float *ptr;
    int64_t stride_x, stride_y;  // in floats; the channel stride is 1
    int32_t x0, y0, w, h;        // frame pixel of out(0,0) and the buffer extent in pixels
};

// Up to four rectangles of the output buffer that a frame zeroes (everything no rendered tile covers), in
// buffer-local pixels along the (slow, fast) memory axes; first[i] = first clear workgroup of rectangle i,
// first[n] = their total.  pitch: floats per slow-axis step.
struct ClearPlan {
    int32_t n;
    int32_t first[5];
    int32_t s0[4], f0[4], rows[4], fw[4];
    int64_t pitch;
};


// Scratch of one radix / partition pass: digit table [bins][chunks rounded up to a multiple of 4] + `bins` row
// totals + `bins` splitters (sample-partitioned depth sort) + 256 four-chunk totals for each of up to
// kSortQuadTotals count workgroups (passes of up to 512K items, which run without a row-scan launch).
// The LSD passes and the 256-bucket partition use kSortBins rows; the depth sort of more than ~1.5M kept
// Gaussians partitions into kSortBinsMax buckets (gsx_sort.hip), over the n depth keys only.
constexpr int kSortBinsMax = 1024;
constexpr int kSortSamples = 2048, kSortSamplesMax = 8192, kSortQuadTotals = 64;
inline size_t radix_temp_bytes(int64_t max_items, int bins = kSortBins) {
    const size_t nblocks = (size_t)((max_items + kSortItems - 1) / kSortItems) + 9 * kSortQuad;   // (the count grid: whole quads, 8 at a time)
    return ((size_t)bins * (nblocks + kSortQuad) + 2 * (size_t)bins + (size_t)kSortQuadTotals * kSortBins) * sizeof(uint32_t);
}
// Where the 64-bit chunk sums start inside `temp`: behind the radix table for max(n, cap) items (and behind the
// kSortBinsMax-row table of the n depth keys).
inline size_t binning_sums_offset(int64_t n, int64_t cap) {
    const int64_t items = n > cap ? n : cap;
    const size_t a = radix_temp_bytes(items > 1 ? items : 1);   // a request for 0 pairs is sized like 1
    const size_t b = radix_temp_bytes(n > 1 ? n : 1, kSortBinsMax);
    return ((a > b ? a : b) + 255) & ~(size_t)255;
}
inline size_t binning_temp_bytes(int64_t n, int64_t cap) {
    const size_t nchunks = (size_t)((n + kEmitChunk - 1) / kEmitChunk) + 2;
    return binning_sums_offset(n, cap) + ((nchunks * sizeof(uint64_t) + 255) & ~(size_t)255);
}
inline int clear_blocks_for(int64_t rows, int64_t fw) {
    const int64_t floats = rows * fw * 3;
    return (int)((floats + kClearFloats - 1) / kClearFloats);
}

// ---- GsxParams.hints: what one frame leaves for the next frame of the same view (device memory, caller-owned)
//   header (64 words)  [kHintSplitters] 256 when `splitters` holds 255 sorted splitters (+ splitters[0] = 0), else 0
//                      [kHintSamples]   number of entries of `samples` the last depth sort filled (2048) or 0
//                      [kHintLens]      number of tiles `lens` was written for, [kHintSched] the same for `sched`
//   splitters[256]     depth-sort splitters of the last frame (written by spare workgroups of its compositing launch)
//   samples[2048]      regularly spaced KEPT depth keys of the last frame (written by its partition count kernel)
//   lens[max_tiles]    COST of every tile of the window, written by the compositing launch: records staged until the tile
//                      was done + 5 per batch (a dense tile that saturates early is cheap however long its list).  Bit 31:
//                      composited as a long tile, on four helper workgroups (set by tile_ranges_kernel with cost 0, raised
//                      by the helpers to the largest of their four costs -- which is what the tile would have cost on one
//                      wave); the schedule counts such a tile as empty, tile_ranges_kernel decides by the cost who is
//                      long next time
//   sched[..]          per XCD, its tiles by falling list length (eight spare workgroups of the projection launch,
//                      gsx_schedule_device.h; header[kHintXcdTiles + x] = how many tiles XCD x has)
//                      header[kHintXcdCost + x] = sum of the costs of XCD x's tiles (same workgroups)
//                      header[kHintLongPct] = the share of a SIMD's load (in %) from which a tile counts as long
//                      (0 = 30; the compositing launch raises it while more than 192 tiles qualify and lowers it again
//                      below 64: the threshold, not the order of arrival, decides who gets the helper slots)
enum { kHintSplitters = 0, kHintSamples = 1, kHintLens = 2, kHintSched = 3, kHintLongPct = 4, kHintXcdTiles = 8 /* .. 15 */,
       kHintXcdCost = 16 /* .. 23 */, kHintHeaderWords = 64 };
constexpr uint32_t kSchedXcds = 8;
// The window's nt tiles are cut into 8 k chunks of consecutive ids, k = max(1, nt / 256) per XCD, whose sizes differ
// by at most one (the first `rem` chunks hold s + 1 tiles, the others s), and chunk c goes to XCD c % 8: every XCD's
// share of the TILES is within k tiles of an eighth, its share of the list entries as even as ~30 chunks spread over
// the frame make it.  (Chunks of two whole tile columns -- 59.5 of them at 1080p -- gave four XCDs 8 chunks and four
// 7, and the compositing launch waited 18 us for the first four: round 3, measured.)
struct SchedCut {
    uint32_t k, s, rem;        // chunks per XCD, size of the small chunks, number of chunks of size s + 1
};
GSX_HD inline SchedCut sched_cut(uint32_t nt) {
    SchedCut c;
    c.k = nt / (kSchedXcds * 32u) ? nt / (kSchedXcds * 32u) : 1u;
    c.s = nt / (kSchedXcds * c.k);
    c.rem = nt - c.s * kSchedXcds * c.k;
    return c;
}
GSX_HD inline uint32_t sched_cap(uint32_t nt, uint32_t /*nwy*/) {
    const SchedCut c = sched_cut(nt);
    return (c.s + 1u) * c.k;
}
struct HintsLayout {
    size_t splitters, samples, lens, sched, total;   // byte offsets
};
// max_tiles: tiles of the frame (a window has at most as many).  The schedule region holds 8 x sched_cap(nt) entries for
// the LARGEST number any window of up to max_tiles tiles needs: 8 k (s + 1) <= nt + 8 k <= nt + nt / 32 + 8, which
// grows with nt (tests/host/plan_sanitize.cpp sweeps it; round 3 sized it nt + 16 x the longer axis, which a frame of
// more than ~512 tiles along its SHORTER axis overran: a 10 000 x 10 000 frame by 900 entries).
inline size_t hints_sched_entries(int64_t max_tiles) {
    const size_t t = (size_t)(max_tiles > 0 ? max_tiles : 1);
    return t + t / 32 + 64;
}
inline HintsLayout hints_layout(int64_t max_tiles, int64_t /*max_axis*/) {
    HintsLayout h;
    const size_t t = (size_t)(max_tiles > 0 ? max_tiles : 1);
    h.splitters = kHintHeaderWords * 4;
    h.samples = h.splitters + (size_t)kSortBins * 4;
    h.lens = h.samples + (size_t)kSortSamples * 4;
    h.sched = h.lens + ((t * 4 + 255) & ~(size_t)255);
    h.total = h.sched + ((hints_sched_entries(max_tiles) * 4 + 255) & ~(size_t)255);
    return h;
}

namespace plan {

constexpr size_t kAlign = 256;
constexpr int64_t kMaxPairs = ((int64_t)1 << 31) - 1;
inline size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

// Fills the first `bytes` bytes of a caller's GsxParams (what ITS header says the struct has; never more than this
// library's own struct, never less than the ABI-300 one) and states that size in struct_size: nothing behind the
// caller's struct is written here or read later (make_plan).
inline void default_params(GsxParams *params, size_t bytes = sizeof(GsxParams)) {
    if (bytes > sizeof(GsxParams)) bytes = sizeof(GsxParams);
    if (bytes < kParamsBytesAbi300) bytes = kParamsBytesAbi300;
    GsxParams d;
    memset(&d, 0, sizeof d);
    d.struct_size = (int32_t)bytes;
    d.semantics = GSX_SEM_REF_CPU;
    d.layout = GSX_LAYOUT_WH3;
    d.tile_x1 = -1;
    d.tile_y1 = -1;
    d.stats_size = (int32_t)sizeof(GsxFrameStats);      // (only a struct that reaches the field receives it)
    memcpy(params, &d, bytes);
}

// GsxParams.original_index of a caller's struct (null when the struct ends before the field): for the entry points that
// read single fields instead of a whole Plan (gsx_preprocess).
inline const int32_t *original_index_of(const GsxParams *params) {
    if (!params) return nullptr;
    const size_t have = params->struct_size ? (size_t)params->struct_size : kParamsBytesAbi300;
    return have >= offsetof(GsxParams, original_index) + sizeof(const int32_t *) ? params->original_index : nullptr;
}

// Number of tiles along an axis.  REF_CPU iterates range(0, extent - tile, tile)
// (splat/gaussian_scene.py:208,214): the last row/column is never rendered.
// REF_CUDA covers the frame (splat/c/render.cu:119-120).
inline int32_t tiles_along(int32_t extent, int32_t tile, int semantics) {
    if (semantics == GSX_SEM_REF_CPU) return extent > tile ? (extent - tile + tile - 1) / tile : 0;
    return (extent + tile - 1) / tile;
}

struct Carve {
    size_t keys0, keys1, vals0, vals1;      // n x u32: depth keys / original indices (ping-pong)
    size_t rec, rect, rrect, bbox;          // per Gaussian: record, tile rectangle by index / by depth rank
    size_t tkeys0, tkeys1, tvals0, tvals1;  // cap x u32: tile ids / Gaussian indices (ping-pong)
    size_t ranges, longs, redo, sched_header, sched, counters, temp, temp_bytes, total;
};

// The 64-byte `counters` block: what the kernels of one frame hand to each other on the device.
//   u32 [0] Gaussians behind the cull plane   [1] Gaussians kept by the depth sort (M)
//       [2] min(D, 2^32 - 1)                   [3] long tiles found (LongTiles.count)
//   i64 at byte 16: n_visible, D (the first two fields of a GsxFrameStats), M again (GsxFrameStats.n_kept)
enum { kCtrCulled = 0, kCtrKept = 1, kCtrPairs = 2, kCtrLong = 3 };

inline Carve carve(int64_t n, int64_t cap, int64_t max_tiles, size_t temp_bytes) {
    Carve c;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = align_up(off + bytes);
        return at;
    };
    size_t nn = (size_t)(n > 0 ? n : 1), cc = (size_t)(cap > 0 ? cap : 1);
    c.keys0 = take(nn * 4); c.keys1 = take(nn * 4); c.vals0 = take(nn * 4); c.vals1 = take(nn * 4);
    c.rec = take(nn * kRecordBytes);
    c.rect = take(nn * kTileRectBytes);
    c.rrect = take(nn * kTileRectBytes);
    c.bbox = take(nn * kBboxBytes);
    c.tkeys0 = take(cc * 4); c.tkeys1 = take(cc * 4); c.tvals0 = take(cc * 4); c.tvals1 = take(cc * 4);
    c.ranges = take((size_t)(max_tiles > 0 ? max_tiles : 1) * kRangeBytes);
    c.longs = take(kMaxLongTiles * sizeof(uint32_t));
    // GsxFrameStats.n_redo on the device (LongTiles.redo)
    c.redo = take(256);
    // a frame's own compositing schedule (no hints to take it from): per XCD, tiles by falling list length -- the layout of
    // the hints' schedule region (hints_sched_entries) with a header of its own in front (kHintHeaderWords words)
    c.sched_header = take(kHintHeaderWords * sizeof(uint32_t));
    c.sched = take(hints_sched_entries(max_tiles) * sizeof(uint32_t));
    c.counters = take(64);
    c.temp = take(temp_bytes);
    c.temp_bytes = temp_bytes;
    c.total = off;
    return c;
}

inline int64_t max_tiles_of(int32_t width, int32_t height, int32_t tile) {
    return (int64_t)((width + tile - 1) / tile) * ((height + tile - 1) / tile);
}
inline int64_t max_axis_tiles_of(int32_t width, int32_t height, int32_t tile) {
    const int64_t a = (width + tile - 1) / tile, b = (height + tile - 1) / tile;
    return a > b ? a : b;
}

// Largest pair capacity whose carve fits `bytes` (the size of a carve grows monotonically with the capacity:
// bisection, exact -- what gsx_workspace_bytes(n, .., cap) asks for always yields at least cap); -1 when not
// even the per-Gaussian part fits.  The kernels index pairs with 32 bits and gsx_workspace_bytes sizes for
// < 2^31 pairs: a larger buffer (a 288 GB part can hand over 68 GB and more) does not raise the capacity
// beyond that.
inline int64_t capacity_for(size_t bytes, int64_t n, int64_t max_tiles) {
    auto fits = [&](int64_t cap) { return carve(n, cap, max_tiles, binning_temp_bytes(n, cap)).total <= bytes; };
    if (!fits(1)) return -1;
    int64_t lo = 1, hi = kMaxPairs;       // invariant: fits(lo), and hi is an upper bound of the answer
    if (fits(hi)) return hi;
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (fits(mid))
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

struct Plan {
    TileGrid grid;
    OutDesc out;
    int semantics;
    bool timing;
    bool no_sync;  // GSX_FLAG_NO_SYNC: nothing waits for the device
    bool generic;  // GSX_FLAG_GENERIC_KERNELS
    bool tight;    // GSX_SEM_STD_3DGS without GSX_FLAG_PUBLISHED_RECTS
    bool split;    // long tiles on four waves (not GSX_FLAG_NO_LONG_TILE_SPLIT)
    bool plain;         // GSX_FLAG_PLAIN_FOOTPRINTS
    size_t stats_bytes; // how far GsxFrameStats is written: GsxParams.stats_size, 64 when the caller did not state it
    int small_batch;    // 1: GSX_FLAG_ONE_VISIBLE, 2: GSX_FLAG_SMALL_BATCH, 0: neither
    int schedule;  // tiles handed out by list length: 1 GSX_FLAG_TILE_SCHEDULE, 0 GSX_FLAG_NO_TILE_SCHEDULE, -1 by size
    const GsxCamera *camera_device;
    uint32_t *tile_counts;
    const float *sh;
    const int32_t *original_index;   // GsxParams.original_index (or null)
    const float *block_bounds;       // GsxParams.block_bounds (or null; only with original_index)
    const int32_t *row_of_index;     // GsxParams.row_of_index (the inverse of original_index)
    int sh_degree;   // -1: RGB colours
    int64_t kept_hint;   // GsxParams.kept_hint (0: unknown)
    char *hints;         // GsxParams.hints (or null)
    bool hints_valid;    // GSX_FLAG_HINTS_VALID
    float background[3];
    // GsxParams.n_substrips: the compositing launch in parts (0: one launch)
    int n_parts, part_axis;
    int32_t part_bounds[kMaxSubstrips + 1];
    void *part_events[kMaxSubstrips];
};

inline int make_plan(int32_t width, int32_t height, int32_t tile, float *out_image, const GsxParams *params, Plan &p,
                     char *msg, size_t msg_bytes) {
    auto fail = [&](int code, const char *fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(msg, msg_bytes, fmt, ap);
        va_end(ap);
        return code;
    };
    GsxParams d;
    default_params(&d);
    d.stats_size = GSX_FRAME_STATS_BYTES_ABI300;      // not stated (NULL params, a struct that ends before the field): the ABI-300 struct
    if (params) {
        // GsxParams.struct_size: only the bytes the caller's struct has are read; fields behind them keep their defaults
        // (include/gsx.h).  0 = the ABI-300 struct, which ends behind `hints`.
        const int32_t have = params->struct_size;
        if (have != 0 && (have < (int32_t)offsetof(GsxParams, kept_hint) || have > (int32_t)sizeof(GsxParams) || (have & 7) != 0))
            return fail(GSX_ERR_INVALID_ARGUMENT, "GsxParams.struct_size %d is not a size this library knows (%zu)", have,
                        sizeof(GsxParams));
        memcpy(&d, params, have ? (size_t)have : kParamsBytesAbi300);
    }
    if (width <= 0 || height <= 0) return fail(GSX_ERR_INVALID_ARGUMENT, "image size %dx%d is not positive", width, height);
    if (tile <= 0 || tile > 1024) return fail(GSX_ERR_INVALID_ARGUMENT, "tile size %d out of range [1,1024]", tile);
    if (d.semantics != GSX_SEM_REF_CPU && d.semantics != GSX_SEM_REF_CUDA && d.semantics != GSX_SEM_STD_3DGS)
        return fail(GSX_ERR_UNSUPPORTED, "unknown semantics %d", d.semantics);
    for (int i = 0; i < 3; ++i) p.background[i] = d.background[i];
    p.camera_device = d.camera_device;
    p.tile_counts = d.tile_counts;
    p.sh = d.sh;
    p.original_index = d.original_index;
    p.block_bounds = d.block_bounds;
    p.row_of_index = d.row_of_index;
    if (d.block_bounds && !d.original_index) return fail(GSX_ERR_INVALID_ARGUMENT, "block_bounds needs original_index");
    if (d.block_bounds && (reinterpret_cast<uintptr_t>(d.block_bounds) & 15u) != 0) return fail(GSX_ERR_INVALID_ARGUMENT, "block_bounds must be 16-byte aligned");
    p.sh_degree = d.sh ? d.sh_degree : -1;
    p.kept_hint = d.kept_hint > 0 ? d.kept_hint : 0;
    p.hints = (char *)d.hints;
    p.hints_valid = d.hints != nullptr && (d.flags & GSX_FLAG_HINTS_VALID) != 0;
    if (d.hints && (reinterpret_cast<uintptr_t>(d.hints) & 255u) != 0)
        return fail(GSX_ERR_INVALID_ARGUMENT, "hints must be 256-byte aligned");
    if (d.sh && (d.sh_degree < 0 || d.sh_degree > 3)) return fail(GSX_ERR_INVALID_ARGUMENT, "SH degree %d outside [0,3]", d.sh_degree);
    if (d.layout != GSX_LAYOUT_WH3 && d.layout != GSX_LAYOUT_HW3) return fail(GSX_ERR_INVALID_ARGUMENT, "unknown layout %d", d.layout);
    if (!out_image) return fail(GSX_ERR_INVALID_ARGUMENT, "out_image is NULL");
    p.semantics = d.semantics;
    p.timing = (d.flags & GSX_FLAG_TIMING) != 0;
    p.no_sync = (d.flags & GSX_FLAG_NO_SYNC) != 0 && !p.timing;
    p.generic = (d.flags & GSX_FLAG_GENERIC_KERNELS) != 0;
    p.tight = d.semantics == GSX_SEM_STD_3DGS && (d.flags & GSX_FLAG_PUBLISHED_RECTS) == 0;
    p.split = (d.flags & GSX_FLAG_NO_LONG_TILE_SPLIT) == 0;
    p.plain = (d.flags & GSX_FLAG_PLAIN_FOOTPRINTS) != 0;
    if (d.stats_size == 0) d.stats_size = GSX_FRAME_STATS_BYTES_ABI300;     // (a zeroed struct: not stated)
    if (d.stats_size != GSX_FRAME_STATS_BYTES_ABI300 && (d.stats_size < (int32_t)sizeof(GsxFrameStats) || (d.stats_size & 7) != 0))
        return fail(GSX_ERR_INVALID_ARGUMENT, "GsxParams.stats_size %d is neither %d nor >= %zu", d.stats_size,
                    GSX_FRAME_STATS_BYTES_ABI300, sizeof(GsxFrameStats));
    p.stats_bytes = d.stats_size > (int32_t)sizeof(GsxFrameStats) ? sizeof(GsxFrameStats) : (size_t)d.stats_size;
    if (p.plain && p.stats_bytes < offsetof(GsxFrameStats, n_redo) + sizeof(int64_t))
        return fail(GSX_ERR_INVALID_ARGUMENT, "GSX_FLAG_PLAIN_FOOTPRINTS needs GsxParams.stats_size >= %zu: n_redo must reach the caller",
                    offsetof(GsxFrameStats, n_redo) + sizeof(int64_t));
    p.small_batch = (d.flags & GSX_FLAG_ONE_VISIBLE) ? 1 : ((d.flags & GSX_FLAG_SMALL_BATCH) ? 2 : 0);
    p.schedule = (d.flags & GSX_FLAG_NO_TILE_SCHEDULE) ? 0 : ((d.flags & GSX_FLAG_TILE_SCHEDULE) ? 1 : -1);
    TileGrid &g = p.grid;
    g.tile = tile;
    g.ntx = tiles_along(width, tile, d.semantics);
    g.nty = tiles_along(height, tile, d.semantics);
    g.width = width;
    g.height = height;
    if (g.ntx > 65535 || g.nty > 65535) return fail(GSX_ERR_UNSUPPORTED, "more than 65535 tiles along an axis");
    g.wx0 = d.tile_x0 < 0 ? 0 : d.tile_x0;
    g.wy0 = d.tile_y0 < 0 ? 0 : d.tile_y0;
    // tile_x1 / tile_y1 < 0: "to the end" (the default); x1 == x0 is an EMPTY window, also at tile 0
    g.wx1 = (d.tile_x1 < 0 || d.tile_x1 > g.ntx) ? g.ntx : d.tile_x1;
    g.wy1 = (d.tile_y1 < 0 || d.tile_y1 > g.nty) ? g.nty : d.tile_y1;
    if (g.wx0 > g.wx1) g.wx0 = g.wx1;
    if (g.wy0 > g.wy1) g.wy0 = g.wy1;
    OutDesc &o = p.out;
    o.ptr = out_image;
    o.x0 = d.out_w > 0 ? d.out_x0 : 0;
    o.y0 = d.out_h > 0 ? d.out_y0 : 0;
    o.w = d.out_w > 0 ? d.out_w : width;
    o.h = d.out_h > 0 ? d.out_h : height;
    // the kernels that zero and write the buffer index its floats with 32 bits per rectangle: 2^30 pixels
    // (a 12.9 GB float frame, e.g. 32768 x 32768) is the most one call renders into
    if ((int64_t)o.w * o.h > ((int64_t)1 << 30))
        return fail(GSX_ERR_UNSUPPORTED, "output buffer of %dx%d pixels is larger than 2^30 pixels", o.w, o.h);
    p.n_parts = 0;
    p.part_axis = 0;
    if (d.n_substrips > 1) {
        if (d.n_substrips > kMaxSubstrips) return fail(GSX_ERR_INVALID_ARGUMENT, "n_substrips %d > %d", d.n_substrips, kMaxSubstrips);
        if (d.substrip_axis != 0 && d.substrip_axis != 1) return fail(GSX_ERR_INVALID_ARGUMENT, "substrip_axis %d is neither 0 nor 1", d.substrip_axis);
        if (!d.substrip_bounds || !d.substrip_events) return fail(GSX_ERR_INVALID_ARGUMENT, "substrip_bounds / substrip_events is NULL");
        const int32_t lo = d.substrip_axis == 0 ? g.wx0 : g.wy0, hi = d.substrip_axis == 0 ? g.wx1 : g.wy1;
        for (int k = 0; k <= d.n_substrips; ++k) {
            p.part_bounds[k] = d.substrip_bounds[k];
            if (k && p.part_bounds[k] < p.part_bounds[k - 1]) return fail(GSX_ERR_INVALID_ARGUMENT, "substrip_bounds are not ascending");
        }
        if (p.part_bounds[0] != lo || p.part_bounds[d.n_substrips] != hi)
            return fail(GSX_ERR_INVALID_ARGUMENT, "substrip_bounds span [%d, %d), the window spans [%d, %d) along axis %d",
                        p.part_bounds[0], p.part_bounds[d.n_substrips], lo, hi, d.substrip_axis);
        for (int k = 0; k < d.n_substrips; ++k) {
            p.part_events[k] = d.substrip_events[k];
            if (!p.part_events[k]) return fail(GSX_ERR_INVALID_ARGUMENT, "substrip_events[%d] is NULL", k);
        }
        p.n_parts = d.n_substrips;
        p.part_axis = d.substrip_axis;
    }
    if (d.layout == GSX_LAYOUT_WH3) {
        o.stride_x = (int64_t)o.h * 3;
        o.stride_y = 3;
    } else {
        o.stride_x = 3;
        o.stride_y = (int64_t)o.w * 3;
    }
    if (g.count() > 0) {
        // every rendered tile must lie inside the output buffer
        int64_t px0 = (int64_t)g.wx0 * tile, px1 = (int64_t)g.wx1 * tile, py0 = (int64_t)g.wy0 * tile, py1 = (int64_t)g.wy1 * tile;
        px1 = px1 > width ? width : px1;   // partial edge tiles (REF_CUDA) end at the frame border
        py1 = py1 > height ? height : py1;
        if (px0 < o.x0 || px1 > (int64_t)o.x0 + o.w || py0 < o.y0 || py1 > (int64_t)o.y0 + o.h)
            return fail(GSX_ERR_INVALID_ARGUMENT, "tile window [%d,%d)x[%d,%d) does not fit the %dx%d output buffer at (%d,%d)",
                        g.wx0, g.wx1, g.wy0, g.wy1, o.w, o.h, o.x0, o.y0);
    }
    return GSX_OK;
}

// The pixels of the output buffer that no tile of the window covers (the compositing kernel writes every
// pixel of every tile it owns, empty tiles included): up to four rectangles, zeroed by extra workgroups
// of the compositing launch instead of a whole-frame memset (25 MB at 1080p).  whole = true: the entire
// buffer (nothing is rendered).
inline ClearPlan make_clear_plan(const Plan &p, bool whole) {
    const OutDesc &o = p.out;
    const int T = p.grid.tile;
    ClearPlan cp;
    memset(&cp, 0, sizeof cp);
    // window in buffer-local pixel coordinates along (slow, fast) memory axes
    const bool wh3 = o.stride_y < o.stride_x;  // x is the slow axis
    const int64_t slow_n = wh3 ? o.w : o.h, fast_n = wh3 ? o.h : o.w;
    cp.pitch = fast_n * 3;
    auto rect = [&](int64_t s0, int64_t s1, int64_t f0, int64_t f1) {
        if (s1 <= s0 || f1 <= f0) return;
        const int i = cp.n++;
        cp.s0[i] = (int32_t)s0; cp.rows[i] = (int32_t)(s1 - s0);
        cp.f0[i] = (int32_t)f0; cp.fw[i] = (int32_t)(f1 - f0);
        cp.first[i + 1] = cp.first[i] + clear_blocks_for(s1 - s0, f1 - f0);
    };
    if (whole || p.grid.count() == 0) {
        // row by row blocks of at most 2^31 floats each: one rectangle per quarter keeps 32-bit indices safe
        const int64_t q = (slow_n + 3) / 4;
        for (int k = 0; k < 4; ++k) rect(k * q, (k + 1) * q < slow_n ? (k + 1) * q : slow_n, 0, fast_n);
        return cp;
    }
    int64_t ws0 = (int64_t)(wh3 ? p.grid.wx0 : p.grid.wy0) * T - (wh3 ? o.x0 : o.y0);
    int64_t ws1 = (int64_t)(wh3 ? p.grid.wx1 : p.grid.wy1) * T;
    int64_t wf1 = (int64_t)(wh3 ? p.grid.wy1 : p.grid.wx1) * T;
    ws1 = (ws1 > (wh3 ? p.grid.width : p.grid.height) ? (wh3 ? p.grid.width : p.grid.height) : ws1) - (wh3 ? o.x0 : o.y0);
    wf1 = (wf1 > (wh3 ? p.grid.height : p.grid.width) ? (wh3 ? p.grid.height : p.grid.width) : wf1) - (wh3 ? o.y0 : o.x0);
    int64_t wf0 = (int64_t)(wh3 ? p.grid.wy0 : p.grid.wx0) * T - (wh3 ? o.y0 : o.x0);
    rect(0, ws0, 0, fast_n);
    rect(ws1, slow_n, 0, fast_n);
    rect(ws0, ws1, 0, wf0);
    rect(ws0, ws1, wf1, fast_n);
    return cp;
}


}  // namespace plan
}  // namespace gsx
