// C ABI of libgsx.so (see include/gsx.h): argument validation, workspace carving and the
// stream-ordered launch sequence.  No device memory is allocated here and nothing is retained
// between calls; errors are returned as codes with a thread-local message.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "gsx_internal.h"

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define GSX_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(GSX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

using namespace gsx;
using namespace gsx::plan;   // Carve, Plan, carve(), capacity_for(), make_plan(), make_clear_plan(): gsx_plan.h

// Optional per-stage timing with HIP events on the launch stream (GSX_FLAG_TIMING).
struct StageTimer {
    bool on = false;
    hipStream_t s = nullptr;
    hipEvent_t ev[6];
    int n = 0;
    void begin(bool enable, hipStream_t stream) {
        on = enable;
        s = stream;
        mark();
    }
    void mark() {
        if (!on || n >= 6) return;
        if (hipEventCreate(&ev[n]) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(ev[n], s);
        ++n;
    }
    // marks: 0 start | 1 project (+keys) | 2 depth sort | 3 scan | 4 bin | 5 blend
    void finish(GsxFrameStats *st) {
        if (n > 0) (void)hipEventSynchronize(ev[n - 1]);
        if (st) {
            for (int i = 0; i < 6; ++i) st->stage_ms[i] = 0.0f;
            for (int i = 1; i < n; ++i) {
                float ms = 0.0f;
                (void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
                st->stage_ms[i - 1] = ms;
            }
            if (n > 1) {
                float ms = 0.0f;
                (void)hipEventElapsedTime(&ms, ev[0], ev[n - 1]);
                st->stage_ms[GSX_STAGE_TOTAL] = ms;
            }
        }
        for (int i = 0; i < n; ++i) (void)hipEventDestroy(ev[i]);
        n = 0;
    }
    ~StageTimer() { finish(nullptr); }
};

int make_plan_or_fail(int32_t width, int32_t height, int32_t tile, float *out_image, const GsxParams *params, Plan &p) {
    char msg[256];
    const int rc = make_plan(width, height, tile, out_image, params, p, msg, sizeof msg);
    return rc == GSX_OK ? GSX_OK : fail(rc, "%s", msg);
}

// Steps shared by both render entry points once records and rank-ordered tile rectangles exist.
// `order` = Gaussian index of each depth rank (nullptr: rows are already in compositing order);
// `m_dev` = number of ranks on the device (nullptr: n); `culled_dev` = Gaussians behind the cull plane.
// The pair count D never has to reach the host for the frame to be enqueued: the kernels read
// it from device memory and their grids are sized by the workspace capacity.  The normal call
// synchronises ONCE, after the last launch, to report the counts and to detect D > capacity;
// GSX_FLAG_NO_SYNC skips even that (the counts then arrive in pinned memory on their own).
// fh: the frame's use of GsxParams.hints (FrameHints below; all null / false without a hints buffer).
struct FrameHints {
    gsx::BlendHints blend = gsx::BlendHints{nullptr, nullptr, nullptr, nullptr, 0u};
    const uint32_t *sched = nullptr;   // hints.sched, put together from the previous frame's list lengths, or null
    uint32_t *sched_region = nullptr;  // the hints buffer's schedule region (whether or not it holds a schedule yet), or null
};

int bin_and_blend(const Plan &p, const Carve &c, char *ws, int64_t n, int64_t cap, const gsx::TileRect *rrect,
                  const uint32_t *order, const uint32_t *m_dev, const uint32_t *culled_dev, bool sums_ready,
                  GsxFrameStats *stats, StageTimer &tm, hipStream_t s, const FrameHints &fh = FrameHints()) {
    uint32_t *counters = (uint32_t *)(ws + c.counters);
    void *temp = ws + c.temp;
    int64_t *dev2 = (int64_t *)(counters + 4);
    uint2 *ranges = (uint2 *)(ws + c.ranges);
    bool counts_on_device = false, counts_in_host = false;
    bool redo_counted = false;      // the compositing launch counts tiles with reference-order records: GsxFrameStats.n_redo
    const bool has_redo = p.stats_bytes >= offsetof(GsxFrameStats, n_redo) + sizeof(int64_t);
    if (p.plain && !stats) return fail(GSX_ERR_INVALID_ARGUMENT, "GSX_FLAG_PLAIN_FOOTPRINTS needs stats_host: n_redo must reach the caller");
    bool parts_marked = false;      // GsxParams.substrip_events recorded (every path records them once, behind its last launch at the latest)
    // no Gaussians: every tile's list is empty -- GsxParams.tile_counts says so (an empty WINDOW has no entries)
    if (n == 0 && p.tile_counts && p.grid.count() > 0)
        GSX_HIP(gsx::launch_zero_words(p.tile_counts, (size_t)p.grid.count(), s));
    if (p.grid.count() > 0 && n == 0 && p.semantics == GSX_SEM_STD_3DGS) {
        // no Gaussians: every pixel of the window is the background colour
        tm.mark();  // 3: scan + emit
        GSX_HIP(gsx::launch_zero_words((uint32_t *)ranges, (size_t)p.grid.count() * 2, s));
        GSX_HIP(gsx::launch_blend((const gsx::Record *)(ws + c.rec), nullptr, (const uint32_t *)(ws + c.tvals0), ranges,
                                  p.grid, p.out, p.semantics, p.background, p.generic, make_clear_plan(p, false),
                                  gsx::LongTiles{nullptr, nullptr, 0u}, nullptr, gsx::BlendHints{nullptr, nullptr, nullptr, nullptr, 0u}, s));
    } else if (n == 0) {
        tm.mark();
        GSX_HIP(gsx::launch_clear(make_clear_plan(p, true), p.out.ptr, s));
    } else {
        // the counts are produced by the emit kernel even when no tile is rendered (an empty window
        // still reports n_visible; its pair count is 0 because every rectangle was clamped away)
        gsx::BinCounts bc{dev2, nullptr, counters + kCtrPairs, counters + kCtrLong, (uint32_t *)(ws + c.redo), culled_dev, n};
        if (p.no_sync && stats) {
            // pinned host memory is device-visible: the emit kernel stores the two counts there itself
            void *alias = nullptr;
            if (hipHostGetDevicePointer(&alias, stats, 0) == hipSuccess && alias) {
                bc.stats2_host = (int64_t *)alias;
                counts_in_host = true;
            } else {
                (void)hipGetLastError();
            }
        }
        GSX_HIP(gsx::emit_instances(temp, rrect, order, m_dev, n, cap, p.grid, ws + c.tkeys0, (uint32_t *)(ws + c.tvals0),
                                    ranges, bc, sums_ready, p.kept_hint, s));
        counts_on_device = true;
        tm.mark();  // 3: scan + emit
        if (p.grid.count() == 0) {
            GSX_HIP(gsx::launch_clear(make_clear_plan(p, true), p.out.ptr, s));
        } else {
            const uint32_t *sorted_vals = nullptr;
            const bool split = p.split && cap > 0 && gsx::blend_splits_long_tiles(p.grid, p.semantics, p.generic);
            gsx::LongTiles lt{counters + kCtrLong, (uint32_t *)(ws + c.longs), split ? gsx::kMaxLongTiles : 0u};
            lt.redo = (uint32_t *)(ws + c.redo);
            if (fh.blend.lens) {        // GsxParams.hints: the tiles' costs decide who is long (tile_ranges_kernel)
                lt.cost = fh.blend.lens;
                lt.header = fh.blend.header;
                lt.cost_pct = (uint32_t)gsx::knob("GSX_LONG_COST_PCT", 30);      // (test library only)
                lt.stay_pct = (uint32_t)gsx::knob("GSX_LONG_STAY_PCT", 75);
                if (!fh.sched || lt.cost_pct == 0) lt.header = nullptr;
            }
            GSX_HIP(gsx::sort_instances(temp, cap, p.grid, ws + c.tkeys0, ws + c.tkeys1, (uint32_t *)(ws + c.tvals0),
                                        (uint32_t *)(ws + c.tvals1), ranges, counters + kCtrPairs, lt, &sorted_vals, s));
            if (p.tile_counts) GSX_HIP(gsx::launch_tile_counts(ranges, p.grid.count(), p.tile_counts, s));
            const uint32_t *sched = fh.sched;      // handed over by the previous frame (GsxParams.hints): no kernel
            gsx::BlendHints bh = fh.blend;
            bh.xcd_sched = fh.sched ? 1u : 0u;
            if (!sched && cap > 0 && gsx::blend_uses_schedule(p.grid, p.semantics, p.generic, n, p.schedule)) {
                if (p.semantics == GSX_SEM_REF_CPU && gsx::knob("GSX_COLD_XCD_SCHED", 1) != 0) {
                    // no schedule on file (a view's first frame, a caller without a hints buffer): the same per-XCD schedule
                    // from THIS frame's list lengths, into the hints buffer's region when there is one (its header then says so)
                    uint32_t *hdr = bh.header ? bh.header : (uint32_t *)(ws + c.sched_header);
                    uint32_t *region = (bh.header && fh.sched_region) ? fh.sched_region : (uint32_t *)(ws + c.sched);
                    if (!(bh.header && !fh.sched_region)) {
                        GSX_HIP(gsx::launch_tile_schedule_xcd(ranges, p.grid.count(), p.grid.nwy(), region, hdr, s));
                        sched = region;
                        bh.header = hdr;
                        bh.xcd_sched = 1u;
                    }
                }
                if (!sched) {       // (the other rule sets: one workgroup ranks the whole frame)
                    sched = (uint32_t *)(ws + c.sched);
                    GSX_HIP(gsx::launch_tile_schedule(ranges, p.grid.count(), (uint32_t *)(ws + c.sched), s));
                }
            }
            tm.mark();  // 4: tile sort (+ the compositing schedule)
            bh.plain = p.plain ? 1u : 0u;
            redo_counted = p.semantics == GSX_SEM_REF_CPU && p.grid.tile == 16 && !p.generic;
            // The 128 spare workgroups hold 16 wave slots of every XCD for ~15 us.  In front of the tiles that is free
            // -- unless the window's tiles fill the chip's 8 192 wave slots just about once (1080p: 7 973 tiles): then
            // some tiles find no slot until the spare workgroups are done.  Behind the tiles they are dispatched as
            // slots free up -- late, so only a frame whose compositing lasts several times their 15 us takes them
            // there (pairs per tile, from the pair capacity = last frame's count).  Measured under rocprofv3, same box:
            // uniform 1M / 1080p 239 us either way, heavy-tailed 1M / 1080p 382 -> 368 us; a 100k frame (36 us of
            // compositing) loses 10 us with them last.
            const int64_t ntl = p.grid.count();
            bh.rank_last = (fh.sched && ntl > 7400 && ntl <= 8192 && (int64_t)cap >= 128 * ntl) ? 1u : 0u;
            if (gsx::knob("GSX_RANK_LAST", -1) >= 0) bh.rank_last = (uint32_t)gsx::knob("GSX_RANK_LAST", -1);   // test library only
            if (p.n_parts > 1 && gsx::blend_in_parts(p.grid, p.semantics, p.generic)) {
                // GsxParams.n_substrips: the same launch once per part of the window, an event of the caller's behind each
                // (a part without tiles is not launched; its event still marks "everything before it is done")
                bool first = true;
                for (int k = 0; k < p.n_parts; ++k) {
                    const gsx::TileSpan span{p.part_axis, p.part_bounds[k], p.part_bounds[k + 1], first ? 1u : 0u, (uint32_t)k};
                    if (span.hi > span.lo || (first && k == p.n_parts - 1)) {
                        GSX_HIP(gsx::launch_blend((const gsx::Record *)(ws + c.rec), (const float4 *)(ws + c.bbox), sorted_vals,
                                                  ranges, p.grid, p.out, p.semantics, p.background, p.generic,
                                                  make_clear_plan(p, false), lt, sched, bh, s, &span));
                        first = false;
                    }
                    GSX_HIP(hipEventRecord((hipEvent_t)p.part_events[k], s));
                }
                parts_marked = true;
            } else {
                GSX_HIP(gsx::launch_blend((const gsx::Record *)(ws + c.rec), (const float4 *)(ws + c.bbox), sorted_vals,
                                          ranges, p.grid, p.out, p.semantics, p.background, p.generic,
                                          make_clear_plan(p, false), lt, sched, bh, s));
            }
            tm.mark();  // 5: blend
        }
    }
    if (!parts_marked)
        for (int k = 0; k < p.n_parts; ++k) GSX_HIP(hipEventRecord((hipEvent_t)p.part_events[k], s));
    if (p.no_sync) {
        if (stats) {
            // stats must be pinned host memory; the two counts land when the stream gets here
            if (!counts_on_device) {
                stats->n_visible = 0;
                stats->n_instances = 0;
                stats->n_kept = 0;
            } else if (!counts_in_host) {
                GSX_HIP(hipMemcpyAsync(stats, dev2, 16, hipMemcpyDeviceToHost, s));
                GSX_HIP(hipMemcpyAsync(&stats->n_kept, dev2 + 2, 8, hipMemcpyDeviceToHost, s));
            }
            if (has_redo) {             // GsxParams.stats_size: the caller's struct reaches n_redo
                stats->n_redo = 0;      // (the upper half stays: the device count has 32 bits)
                if (redo_counted) GSX_HIP(hipMemcpyAsync(&stats->n_redo, ws + c.redo, 4, hipMemcpyDeviceToHost, s));
            }
            stats->n_tiles = p.grid.count();
            stats->reserved = cap;  // > 0: counts are delivered asynchronously; value = pair capacity used
        }
        return GSX_OK;
    }
    int64_t host2[3] = {0, 0, 0};
    uint32_t redo_host = 0;
    if (counts_on_device) GSX_HIP(hipMemcpyAsync(host2, dev2, 24, hipMemcpyDeviceToHost, s));
    if (redo_counted) GSX_HIP(hipMemcpyAsync(&redo_host, ws + c.redo, 4, hipMemcpyDeviceToHost, s));
    GSX_HIP(hipStreamSynchronize(s));
    if (stats) {
        stats->n_visible = host2[0];
        stats->n_instances = host2[1];
        stats->n_kept = host2[2];
        if (has_redo) stats->n_redo = redo_host;
        stats->n_tiles = p.grid.count();
        stats->reserved = 0;
    }
    tm.finish(stats);
    if (host2[1] > cap)
        return fail(GSX_ERR_WORKSPACE_TOO_SMALL, "frame needs %lld tile instances, workspace holds %lld",
                    (long long)host2[1], (long long)cap);
    return GSX_OK;
}

int check_workspace(void *workspace, size_t bytes, int64_t n, int64_t max_tiles, Carve &c, int64_t &cap) {
    if (!workspace) return fail(GSX_ERR_INVALID_ARGUMENT, "workspace is NULL");
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return fail(GSX_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    cap = capacity_for(bytes, n, max_tiles);   // never above 2^31 - 1 pairs, whatever the buffer size
    if (cap < 0)
        return fail(GSX_ERR_WORKSPACE_TOO_SMALL, "workspace of %zu bytes cannot hold %lld Gaussians", bytes, (long long)n);
    c = carve(n, cap, max_tiles, binning_temp_bytes(n, cap));
    return GSX_OK;
}

}  // namespace

extern "C" {

int gsx_version(void) { return GSX_VERSION; }

const char *gsx_last_error(void) { return g_error; }

void gsx_default_params_sized(GsxParams *params, size_t struct_size) {
    if (params) default_params(params, struct_size);
}

// the symbol binaries built against the ABI 300 / 301 headers call (there a plain function, the struct 104 bytes):
// the name is parenthesised because this header makes it a macro
void (gsx_default_params)(GsxParams *params) {
    if (params) default_params(params, kParamsBytesAbi300);
}

size_t gsx_hints_bytes(int32_t width, int32_t height, int32_t tile) {
    if (width <= 0 || height <= 0 || tile <= 0) return 0;
    return gsx::hints_layout(max_tiles_of(width, height, tile), max_axis_tiles_of(width, height, tile)).total;
}

size_t gsx_workspace_bytes(int64_t n, int32_t width, int32_t height, int32_t tile, int64_t max_instances) {
    if (n < 0 || width <= 0 || height <= 0 || tile <= 0 || max_instances < 0) return 0;
    if (n >= (int64_t)1 << 31 || max_instances >= (int64_t)1 << 31) return 0;
    size_t temp = binning_temp_bytes(n, max_instances);
    if (temp == 0) return 0;
    return carve(n, max_instances, max_tiles_of(width, height, tile), temp).total;
}

int gsx_preprocess(const GsxCamera *camera, const float *means3d, const float *scales, const float *quats,
                   const float *opacity_logit, const float *colors, int64_t n, float *points_xy, float *colors_out,
                   float *covariance_2d, float *depths, float *inverse_covariance_2d, float *radius, float *min_x,
                   float *max_x, float *min_y, float *max_y, float *sigmoid_opacity, int32_t *order,
                   int64_t *n_visible_host, const GsxParams *params, void *workspace, size_t workspace_bytes,
                   void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!camera) return fail(GSX_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (n < 0 || n >= (int64_t)1 << 31) return fail(GSX_ERR_INVALID_ARGUMENT, "n = %lld out of range", (long long)n);
    const int small_batch = !params ? 0 : ((params->flags & GSX_FLAG_ONE_VISIBLE) ? 1 : ((params->flags & GSX_FLAG_SMALL_BATCH) ? 2 : 0));
    if (n_visible_host) *n_visible_host = 0;
    if (n == 0) return GSX_OK;
    if (!means3d || !scales || !quats || !opacity_logit || !colors) return fail(GSX_ERR_INVALID_ARGUMENT, "an input array is NULL");
    if (!points_xy || !colors_out || !covariance_2d || !depths || !inverse_covariance_2d || !radius || !min_x || !max_x ||
        !min_y || !max_y || !sigmoid_opacity)
        return fail(GSX_ERR_INVALID_ARGUMENT, "an output array is NULL");
    Carve c;
    int64_t cap;
    int rc = check_workspace(workspace, workspace_bytes, n, 1, c, cap);
    if (rc != GSX_OK) return rc;
    char *ws = (char *)workspace;
    uint32_t *k0 = (uint32_t *)(ws + c.keys0), *k1 = (uint32_t *)(ws + c.keys1);
    uint32_t *v0 = (uint32_t *)(ws + c.vals0), *v1 = (uint32_t *)(ws + c.vals1);
    uint32_t *counters = (uint32_t *)(ws + c.counters);
    // Three steps.  (1) Original order, coalesced: depth key + the projected quantities of every visible Gaussian in
    // its record slot (the kernel also zeroes the sort's counters).  (2) The SAME compacting depth sort as the
    // whole-path entry: it drops -- and counts -- the culled Gaussians in its first step and leaves the index of every
    // depth rank (the rectangles it carries along are not needed here: it moves whatever the workspace holds).
    // (3) Rank order: one 48-byte gather per rank, the derived fields, eleven coalesced output arrays.  Round 2 ran a
    // memset node, a key kernel, a 12-launch LSD sort over all n keys, a counting kernel and a projection that
    // gathered the five input arrays by rank.
    gsx::GaussiansIn in{means3d, scales, quats, opacity_logit, colors, original_index_of(params)};
    gsx::Record *stage = (gsx::Record *)(ws + c.rec);
    GSX_HIP(gsx::launch_project_stage(*camera, in, n, k0, stage, counters, small_batch, s));
    const gsx::DepthRoute route = gsx::depth_sort_route(n, 0);
    if (route != gsx::kDepthLsd)
        GSX_HIP(gsx::sort_depth_sampled(route, ws + c.temp, k0, k1, v0, v1, n, 0, counters + kCtrKept, counters + kCtrCulled,
                                        (const gsx::TileRect *)(ws + c.rect), (gsx::TileRect *)(ws + c.rrect), 0, nullptr,
                                        gsx::SortHints{nullptr, nullptr, nullptr, false}, s));
    else
        GSX_HIP(gsx::sort_depth_compact(ws + c.temp, k0, k1, v0, v1, n, counters + kCtrKept, counters + kCtrCulled,
                                        (const gsx::TileRect *)(ws + c.rect), (gsx::TileRect *)(ws + c.rrect), s));
    gsx::StageOneOut out{points_xy, colors_out, covariance_2d, depths, inverse_covariance_2d, radius,
                         min_x, max_x, min_y, max_y, sigmoid_opacity, order};
    GSX_HIP(gsx::launch_project_full(stage, v0, counters + kCtrKept, n, out, s));
    uint32_t nv = 0;
    GSX_HIP(hipMemcpyAsync(&nv, counters + kCtrKept, 4, hipMemcpyDeviceToHost, s));
    GSX_HIP(hipStreamSynchronize(s));
    if (n_visible_host) *n_visible_host = nv;
    return GSX_OK;
}

int gsx_render_preprocessed(int32_t image_height, int32_t image_width, int32_t tile_size, const float *point_means,
                            const float *point_colors, const float *inverse_covariance_2d, const float *min_x,
                            const float *max_x, const float *min_y, const float *max_y, const float *opacity,
                            int64_t n, float *out_image, const GsxParams *params, GsxFrameStats *stats_host,
                            void *workspace, size_t workspace_bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    Plan p;
    int rc = make_plan_or_fail(image_width, image_height, tile_size, out_image, params, p);
    if (rc != GSX_OK) return rc;
    if (p.semantics == GSX_SEM_STD_3DGS)
        return fail(GSX_ERR_UNSUPPORTED, "GSX_SEM_STD_3DGS has its own stage 1: use gsx_render_forward");
    if (n < 0 || n >= (int64_t)1 << 31) return fail(GSX_ERR_INVALID_ARGUMENT, "n = %lld out of range", (long long)n);
    if (n > 0 && (!point_means || !point_colors || !inverse_covariance_2d || !min_x || !max_x || !min_y || !max_y || !opacity))
        return fail(GSX_ERR_INVALID_ARGUMENT, "an input array is NULL");
    Carve c;
    int64_t cap;
    rc = check_workspace(workspace, workspace_bytes, n, max_tiles_of(image_width, image_height, tile_size), c, cap);
    if (rc != GSX_OK) return rc;
    char *ws = (char *)workspace;
    StageTimer tm;
    tm.begin(p.timing, s);
    tm.mark();  // 1: (no depth sort on this entry point)
    gsx::PreprocessedIn in{point_means, point_colors, inverse_covariance_2d, min_x, max_x, min_y, max_y, opacity};
    GSX_HIP(gsx::launch_pack_preprocessed(in, n, p.grid, p.semantics, (gsx::Record *)(ws + c.rec),
                                          (gsx::TileRect *)(ws + c.rect),
                                          p.semantics != GSX_SEM_STD_3DGS ? (float4 *)(ws + c.bbox) : nullptr, s));
    tm.mark();  // 2: pack
    // rows are already in compositing order: the rectangles by row ARE the rectangles by rank
    return bin_and_blend(p, c, ws, n, cap, (const gsx::TileRect *)(ws + c.rect), nullptr, nullptr, nullptr, false, stats_host,
                         tm, s);
}

int gsx_render_forward(const GsxCamera *camera, const float *means3d, const float *scales, const float *quats,
                       const float *opacity_logit, const float *colors, int64_t n, int32_t tile_size,
                       float *out_image, const GsxParams *params, GsxFrameStats *stats_host, void *workspace,
                       size_t workspace_bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!camera) return fail(GSX_ERR_INVALID_ARGUMENT, "camera is NULL");
    Plan p;
    int rc = make_plan_or_fail(camera->width, camera->height, tile_size, out_image, params, p);
    if (rc != GSX_OK) return rc;
    if (n < 0 || n >= (int64_t)1 << 31) return fail(GSX_ERR_INVALID_ARGUMENT, "n = %lld out of range", (long long)n);
    if (p.n_parts > 0) {
        // substrip_events are recorded with plain hipEventRecord: inside a stream capture they would become nodes of
        // the graph, and the caller's hipStreamWaitEvent on another stream would fail or invalidate the capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        GSX_HIP(hipStreamIsCapturing(s, &cs));
        if (cs != hipStreamCaptureStatusNone)
            return fail(GSX_ERR_UNSUPPORTED, "n_substrips cannot be combined with stream capture (the part events are recorded on the stream)");
    }
    if (n > 0 && (!means3d || !scales || !quats || !opacity_logit || (!colors && !p.sh)))
        return fail(GSX_ERR_INVALID_ARGUMENT, "an input array is NULL");
    if (p.original_index && !p.row_of_index) return fail(GSX_ERR_INVALID_ARGUMENT, "original_index needs row_of_index (its inverse)");
    const uint32_t *row_of = p.original_index ? (const uint32_t *)p.row_of_index : nullptr;
    Carve c;
    int64_t cap;
    rc = check_workspace(workspace, workspace_bytes, n, max_tiles_of(camera->width, camera->height, tile_size), c, cap);
    if (rc != GSX_OK) return rc;
    char *ws = (char *)workspace;
    uint32_t *k0 = (uint32_t *)(ws + c.keys0), *k1 = (uint32_t *)(ws + c.keys1);
    uint32_t *v0 = (uint32_t *)(ws + c.vals0), *v1 = (uint32_t *)(ws + c.vals1);
    StageTimer tm;
    tm.begin(p.timing, s);
    gsx::GaussiansIn in{means3d, scales, quats, opacity_logit, p.sh ? p.sh : colors, p.original_index, (const float4 *)p.block_bounds};
    uint32_t *counters = (uint32_t *)(ws + c.counters);
    // GsxParams.hints: what the previous frame of this view left (splitters, tile-list lengths) and what this one
    // leaves.  Only where every producer and consumer exists: the tile-16 REF_CPU compositing kernel (lengths,
    // ranked samples), the 256-bucket depth sort (splitters; the LSD passes of larger scenes only take the schedule).
    const gsx::DepthRoute route = gsx::depth_sort_route(n, p.kept_hint);
    const bool sampled = route != gsx::kDepthLsd;
    FrameHints fh;
    gsx::SortHints sh{nullptr, nullptr, nullptr, false};
    gsx::ScheduleHint sched_hint{nullptr, nullptr, nullptr, 0u};
    if (p.hints && (route == gsx::kDepth256 || route == gsx::kDepthLsd) && n >= gsx::kSortSamples && p.grid.count() > 0 &&
        gsx::blend_splits_long_tiles(p.grid, p.semantics, p.generic)) {
        const gsx::HintsLayout hl = gsx::hints_layout(max_tiles_of(camera->width, camera->height, tile_size),
                                                      max_axis_tiles_of(camera->width, camera->height, tile_size));
        uint32_t *hdr = (uint32_t *)p.hints;
        // (the LSD passes of larger scenes have no use for splitters, but they leave the sample all the same: the next
        // frame of the view may take the 256-bucket route -- a rank's strip does once its kept count is known)
        sh = gsx::SortHints{hdr, (const uint32_t *)(p.hints + hl.splitters), (uint32_t *)(p.hints + hl.samples),
                            p.hints_valid && route == gsx::kDepth256};
        // the 256-bucket route leaves the next frame's splitters itself -- the exact quantiles of this frame's keys (bucket_sort_kernel)
        // -- and the compositing launch has no sample to rank; the LSD route still leaves a sample for it
        const bool exact = route == gsx::kDepth256 && gsx::knob("GSX_EXACT_SPLITTERS", 1) != 0;
        if (exact) sh.next_splitters = (uint32_t *)(p.hints + hl.splitters);
        fh.blend = gsx::BlendHints{hdr, exact ? nullptr : sh.samples, (uint32_t *)(p.hints + hl.splitters), (uint32_t *)(p.hints + hl.lens), 0u};
        fh.sched_region = (uint32_t *)(p.hints + hl.sched);
        // the schedule costs nothing here (a spare workgroup of the projection launch): every window of more than two
        // tiles per SIMD gets one, unless told not to
        if (p.hints_valid && p.schedule != 0 && p.grid.count() > 2048) {
            sched_hint = gsx::ScheduleHint{(const uint32_t *)(p.hints + hl.lens), (uint32_t *)(p.hints + hl.sched), hdr,
                                           (uint32_t)p.grid.count(), (uint32_t)p.grid.nwy()};
            fh.sched = sched_hint.sched;
        }
    }
    // no splitters on file (a view's first frame, a caller without a hints buffer): the projection launch ranks a sample itself
    gsx::SampleHint presample;
    if (sampled && !sh.use && gsx::knob("GSX_PRESAMPLE", 1) != 0)
        presample = gsx::depth_presample(route, ws + c.temp, n, gsx::emit_chunk_sums(ws + c.temp, n, cap), row_of);
    GSX_HIP(gsx::launch_project_pack(*camera, p.camera_device, in, n, p.grid, p.semantics, p.tight, p.small_batch, p.sh_degree, k0, (gsx::Record *)(ws + c.rec),
                                     (gsx::TileRect *)(ws + c.rect), counters,
                                     p.semantics != GSX_SEM_STD_3DGS ? (float4 *)(ws + c.bbox) : nullptr, sched_hint,
                                     c.temp_bytes >= (size_t)(n / GSX_BOUNDS_ROWS + 1) ? (uint8_t *)(ws + c.temp) : nullptr, s, presample));
    tm.mark();  // 1: project (+ depth keys)
    // the sampled routes also leave the per-chunk tile counts the pair emission starts from (one kernel less)
    if (sampled)
        GSX_HIP(gsx::sort_depth_sampled(route, ws + c.temp, k0, k1, v0, v1, n, p.kept_hint, counters + kCtrKept,
                                        counters + kCtrCulled, (const gsx::TileRect *)(ws + c.rect),
                                        (gsx::TileRect *)(ws + c.rrect), 0, gsx::emit_chunk_sums(ws + c.temp, n, cap), sh, s, row_of,
                                        presample.splitters != nullptr));
    else {
        // the LSD passes carry the rectangles along, packed into 4 bytes, when tile coordinates fit 8 bits; the two
        // arrays they travel in are the pair lists' value arrays, which nothing uses before the emission
        const bool carry = p.grid.ntx <= 256 && p.grid.nty <= 256 && cap >= n && gsx::knob("GSX_LSD_CARRY", 1) != 0;
        GSX_HIP(gsx::sort_depth_compact(ws + c.temp, k0, k1, v0, v1, n, counters + kCtrKept, counters + kCtrCulled,
                                        (const gsx::TileRect *)(ws + c.rect), (gsx::TileRect *)(ws + c.rrect), s, sh.samples,
                                        carry ? (uint32_t *)(ws + c.tvals0) : nullptr,
                                        carry ? (uint32_t *)(ws + c.tvals1) : nullptr, row_of));
    }
    tm.mark();  // 2: depth sort (drops what reaches no tile, leaves the rectangles in rank order)
    return bin_and_blend(p, c, ws, n, cap, (const gsx::TileRect *)(ws + c.rrect), v0, counters + kCtrKept,
                         counters + kCtrCulled, sampled, stats_host, tm, s, fh);
}

int gsx_sh_to_rgb(const float *means3d, const float *sh, int32_t degree, int64_t n, const float *camera_center_host,
                  float *colors_out, void *stream) {
    if (degree < 0 || degree > 3) return fail(GSX_ERR_INVALID_ARGUMENT, "SH degree %d outside [0,3]", degree);
    if (n < 0) return fail(GSX_ERR_INVALID_ARGUMENT, "n is negative");
    if (!camera_center_host) return fail(GSX_ERR_INVALID_ARGUMENT, "camera_center_host is NULL");
    if (n > 0 && (!means3d || !sh || !colors_out)) return fail(GSX_ERR_INVALID_ARGUMENT, "an array is NULL");
    GSX_HIP(gsx::launch_sh_to_rgb(means3d, sh, degree, n, camera_center_host, colors_out, (hipStream_t)stream));
    return GSX_OK;
}

int gsx_covariance_3d(const float *scales, const float *quats, int64_t n, float *covariance_out, void *stream) {
    if (n < 0) return fail(GSX_ERR_INVALID_ARGUMENT, "n is negative");
    if (n > 0 && (!scales || !quats || !covariance_out)) return fail(GSX_ERR_INVALID_ARGUMENT, "an array is NULL");
    GSX_HIP(gsx::launch_covariance3d(scales, quats, n, covariance_out, (hipStream_t)stream));
    return GSX_OK;
}

int gsx_covariance_2d(const GsxCamera *camera, const float *points, const float *covariance_3d, int64_t n,
                      float *covariance_2d_out, void *stream) {
    if (!camera) return fail(GSX_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (n < 0) return fail(GSX_ERR_INVALID_ARGUMENT, "n is negative");
    if (n > 0 && (!points || !covariance_3d || !covariance_2d_out)) return fail(GSX_ERR_INVALID_ARGUMENT, "an array is NULL");
    GSX_HIP(gsx::launch_covariance2d(*camera, points, covariance_3d, n, covariance_2d_out, (hipStream_t)stream));
    return GSX_OK;
}

int gsx_project_points(const GsxCamera *camera, const float *means3d, int64_t n, float *points_out,
                       uint8_t *in_view_out, void *stream) {
    if (!camera) return fail(GSX_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (n < 0) return fail(GSX_ERR_INVALID_ARGUMENT, "n is negative");
    if (n > 0 && (!means3d || !points_out || !in_view_out)) return fail(GSX_ERR_INVALID_ARGUMENT, "an array is NULL");
    GSX_HIP(gsx::launch_project_points(*camera, means3d, n, points_out, in_view_out, (hipStream_t)stream));
    return GSX_OK;
}

}  // extern "C"
