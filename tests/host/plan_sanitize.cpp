// Host-side planning arithmetic of libgsx (intro_to_gaussian_splatting_amd/csrc/gsx_plan.h) under
// AddressSanitizer + UndefinedBehaviorSanitizer: carve / capacity_for / make_plan / make_clear_plan swept over
// Gaussian counts, pair capacities, tile sizes, frames and windows up to the 2^31 limits.  No GPU, no HIP:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I include \
//       -I intro_to_gaussian_splatting_amd/csrc tests/host/plan_sanitize.cpp -o plan_sanitize && ./plan_sanitize
// (tests/test_host_sanitize.py does exactly this.)  Exit code 0 and "ok" = every invariant held.
#include <stdlib.h>

#include <vector>

#include "gsx_plan.h"

using namespace gsx;
using namespace gsx::plan;

static long long g_checks = 0;
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        ++g_checks;                                                                      \
        if (!(cond)) {                                                                   \
            fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond);    \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static int64_t rnd_in(int64_t lo, int64_t hi) { return lo + (int64_t)(rnd() % (uint64_t)(hi - lo + 1)); }

// Regions of a carve: disjoint, 256-byte aligned, inside [0, total), each large enough for what it holds.
static void check_carve(int64_t n, int64_t cap, int64_t max_tiles) {
    const size_t temp = binning_temp_bytes(n, cap);
    const Carve c = carve(n, cap, max_tiles, temp);
    const size_t nn = (size_t)(n > 0 ? n : 1), cc = (size_t)(cap > 0 ? cap : 1), tt = (size_t)(max_tiles > 0 ? max_tiles : 1);
    struct R { size_t off, bytes; };
    const R regs[] = {{c.keys0, nn * 4}, {c.keys1, nn * 4}, {c.vals0, nn * 4}, {c.vals1, nn * 4},
                      {c.rec, nn * kRecordBytes}, {c.rect, nn * kTileRectBytes}, {c.rrect, nn * kTileRectBytes},
                      {c.bbox, nn * kBboxBytes}, {c.tkeys0, cc * 4}, {c.tkeys1, cc * 4}, {c.tvals0, cc * 4},
                      {c.tvals1, cc * 4}, {c.ranges, tt * kRangeBytes}, {c.longs, kMaxLongTiles * 4},
                      {c.counters, 64}, {c.temp, temp}};
    size_t prev_end = 0;
    for (const R &r : regs) {
        CHECK(r.off % 256 == 0);
        CHECK(r.off >= prev_end);
        CHECK(r.off + r.bytes <= c.total);
        prev_end = r.off + r.bytes;
    }
    // the digit table of the larger sort and the chunk sums both fit `temp`
    const int64_t items = n > cap ? n : cap;
    const size_t nquads = (size_t)((items + kSortItems - 1) / kSortItems + kSortQuad - 1) / kSortQuad;
    CHECK(((size_t)kSortBins * nquads * kSortQuad + 2 * kSortBins + (size_t)kSortQuadTotals * kSortBins) * 4 <= binning_sums_offset(n, cap));
    // ... and so does the kSortBinsMax-row table of the depth keys (1024-bucket partition of large scenes)
    const size_t dquads = (size_t)((n + kSortItems - 1) / kSortItems + kSortQuad - 1) / kSortQuad;
    CHECK(((size_t)kSortBinsMax * dquads * kSortQuad + 2 * kSortBinsMax) * 4 <= binning_sums_offset(n, cap));
    const size_t nchunks = (size_t)((n + kEmitChunk - 1) / kEmitChunk);
    CHECK(binning_sums_offset(n, cap) + (nchunks + 1) * 8 <= temp);
}

static void check_capacity(int64_t n, int64_t cap, int32_t w, int32_t h, int32_t tile) {
    const int64_t max_tiles = max_tiles_of(w, h, tile);
    const Carve want = carve(n, cap, max_tiles, binning_temp_bytes(n, cap));
    const int64_t got = capacity_for(want.total, n, max_tiles);
    CHECK(got >= cap);                    // the bytes gsx_workspace_bytes asks for hold the pairs asked for
    CHECK(got <= kMaxPairs);
    const Carve fit = carve(n, got, max_tiles, binning_temp_bytes(n, got));
    CHECK(fit.total <= want.total);       // and the capacity derived from them fits them
    if (want.total > 4096) CHECK(capacity_for(want.total - 4096, n, max_tiles) <= got);
    CHECK(capacity_for(0, n, max_tiles) == -1);
    CHECK(cap == 0 || got == cap || carve(n, got + 1, max_tiles, binning_temp_bytes(n, got + 1)).total > want.total || got == kMaxPairs);
    check_carve(n, got, max_tiles);
}

static void check_plan(int32_t w, int32_t h, int32_t tile, int sem, int layout, const int32_t win[4], bool strip_buffer) {
    GsxParams prm;
    default_params(&prm);
    prm.semantics = sem;
    prm.layout = layout;
    prm.tile_x0 = win[0]; prm.tile_x1 = win[1]; prm.tile_y0 = win[2]; prm.tile_y1 = win[3];
    float dummy = 0.0f;
    Plan p;
    char msg[256] = "";
    // a strip-sized output buffer that exactly covers the window (what strips.py hands over)
    const int32_t ntx = tiles_along(w, tile, sem), nty = tiles_along(h, tile, sem);
    int32_t x0 = win[0] < 0 ? 0 : win[0], y0 = win[2] < 0 ? 0 : win[2];
    int32_t x1 = (win[1] < 0 || win[1] > ntx) ? ntx : win[1], y1 = (win[3] < 0 || win[3] > nty) ? nty : win[3];
    if (x0 > x1) x0 = x1;
    if (y0 > y1) y0 = y1;
    if (strip_buffer && x1 > x0 && y1 > y0) {
        const int64_t px1 = (int64_t)x1 * tile > w ? w : (int64_t)x1 * tile, py1 = (int64_t)y1 * tile > h ? h : (int64_t)y1 * tile;
        prm.out_x0 = x0 * tile; prm.out_y0 = y0 * tile;
        prm.out_w = (int32_t)(px1 - (int64_t)x0 * tile); prm.out_h = (int32_t)(py1 - (int64_t)y0 * tile);
    }
    const int rc = make_plan(w, h, tile, &dummy, &prm, p, msg, sizeof msg);
    const int64_t out_px = (int64_t)(prm.out_w > 0 ? prm.out_w : w) * (prm.out_h > 0 ? prm.out_h : h);
    if (ntx > 65535 || nty > 65535 || out_px > ((int64_t)1 << 30)) {
        CHECK(rc == GSX_ERR_UNSUPPORTED);
        return;
    }
    CHECK(rc == GSX_OK);
    const TileGrid &g = p.grid;
    CHECK(g.ntx == ntx && g.nty == nty);
    CHECK(0 <= g.wx0 && g.wx0 <= g.wx1 && g.wx1 <= g.ntx && 0 <= g.wy0 && g.wy0 <= g.wy1 && g.wy1 <= g.nty);
    CHECK(g.wx0 == x0 && g.wx1 == x1 && g.wy0 == y0 && g.wy1 == y1);
    CHECK(g.count() == (int64_t)(x1 - x0) * (y1 - y0));
    // the rectangles a frame zeroes + the pixels its tiles write = the whole output buffer, none twice
    const ClearPlan cp = make_clear_plan(p, false);
    CHECK(cp.n >= 0 && cp.n <= 4);
    int64_t cleared = 0;
    const bool wh3 = p.out.stride_y < p.out.stride_x;
    const int64_t slow_n = wh3 ? p.out.w : p.out.h, fast_n = wh3 ? p.out.h : p.out.w;
    for (int i = 0; i < cp.n; ++i) {
        CHECK(cp.rows[i] > 0 && cp.fw[i] > 0 && cp.s0[i] >= 0 && cp.f0[i] >= 0);
        CHECK((int64_t)cp.s0[i] + cp.rows[i] <= slow_n && (int64_t)cp.f0[i] + cp.fw[i] <= fast_n);
        CHECK(cp.first[i + 1] - cp.first[i] == clear_blocks_for(cp.rows[i], cp.fw[i]));
        CHECK((int64_t)(cp.first[i + 1] - cp.first[i]) * kClearFloats >= (int64_t)cp.rows[i] * cp.fw[i] * 3);
        cleared += (int64_t)cp.rows[i] * cp.fw[i];
        for (int j = 0; j < i; ++j) {     // pairwise disjoint
            const bool sep = cp.s0[i] + cp.rows[i] <= cp.s0[j] || cp.s0[j] + cp.rows[j] <= cp.s0[i] ||
                             cp.f0[i] + cp.fw[i] <= cp.f0[j] || cp.f0[j] + cp.fw[j] <= cp.f0[i];
            CHECK(sep);
        }
    }
    int64_t tile_px = 0;
    if (g.count() > 0) {
        const int64_t px1 = (int64_t)g.wx1 * tile > w ? w : (int64_t)g.wx1 * tile, py1 = (int64_t)g.wy1 * tile > h ? h : (int64_t)g.wy1 * tile;
        tile_px = (px1 - (int64_t)g.wx0 * tile) * (py1 - (int64_t)g.wy0 * tile);
    }
    CHECK(cleared + tile_px == (int64_t)p.out.w * p.out.h);
    const ClearPlan whole = make_clear_plan(p, true);
    int64_t all = 0;
    for (int i = 0; i < whole.n; ++i) all += (int64_t)whole.rows[i] * whole.fw[i];
    CHECK(all == (int64_t)p.out.w * p.out.h);
}

int main() {
    // ---- workspace carving and capacity, hand-picked edges and a random sweep up to the 2^31 limits
    const int64_t big = ((int64_t)1 << 31) - 1;
    const int64_t ns[] = {0, 1, 2047, 2048, 2049, 131072, 131073, 1000000, 5000000, 20000000, 500000000, big};
    const int64_t caps[] = {0, 1, 4095, 4096, 2048 * 64, 2048 * 64 + 1, 4700000, 85000000, 1000000000, big};
    for (int64_t n : ns)
        for (int64_t cap : caps) {
            check_carve(n, cap, 32400);
            check_capacity(n, cap, 3840, 2160, 16);
        }
    for (int it = 0; it < 200000; ++it) {
        const int shift_n = (int)rnd_in(0, 31), shift_c = (int)rnd_in(0, 31);
        const int64_t n = rnd_in(0, ((int64_t)1 << shift_n) - 1 + (shift_n == 31 ? 0 : 0));
        const int64_t cap = rnd_in(0, ((int64_t)1 << shift_c) - 1);
        const int32_t tile = (int32_t)rnd_in(1, 64), w = (int32_t)rnd_in(1, 8192), h = (int32_t)rnd_in(1, 8192);
        check_capacity(n, cap, w, h, tile);
    }
    // a buffer far larger than any frame needs (a 288 GB part can hand over > 68 GB): the capacity stays < 2^31
    CHECK(capacity_for((size_t)200 << 30, 1000000, 8100) == kMaxPairs);
    // ---- GsxParams.hints: the per-XCD schedule of ANY window of up to max_tiles tiles fits the region hints_layout
    //      reserves (the projection launch's spare workgroups write 8 x sched_cap(nt) entries behind h.sched)
    {
        auto fits = [&](int64_t max_tiles, int64_t nt) {
            const HintsLayout h = hints_layout(max_tiles, 0);
            const size_t need = (size_t)kSchedXcds * sched_cap((uint32_t)nt, 0u) * sizeof(uint32_t);
            CHECK(h.sched + need <= h.total);
            CHECK(h.lens + (size_t)max_tiles * 4 <= h.sched);
        };
        for (int64_t nt = 1; nt <= 70000; ++nt) fits(nt, nt);                       // every window size of small frames
        const int64_t frames[] = {8100, 32400, 389376, 625 * 625, 1024 * 1024, 4096 * 4096};
        for (int64_t mt : frames) {
            fits(mt, mt);
            for (int it = 0; it < 20000; ++it) fits(mt, rnd_in(1, mt));
        }
        for (int it = 0; it < 200000; ++it) {
            const int64_t mt = rnd_in(1, (int64_t)1 << 26);
            fits(mt, rnd_in(1, mt));
            fits(mt, mt);
        }
    }
    // ---- plans: frames, tile sizes, semantics, layouts, windows (empty ones at tile 0 included)
    const int sems[] = {GSX_SEM_REF_CPU, GSX_SEM_REF_CUDA, GSX_SEM_STD_3DGS};
    for (int it = 0; it < 300000; ++it) {
        const int32_t tile = (int32_t)rnd_in(1, 1024);
        const int32_t w = (int32_t)rnd_in(1, it % 7 == 0 ? 2000000000 : 9000), h = (int32_t)rnd_in(1, it % 11 == 0 ? 2000000000 : 9000);
        const int sem = sems[rnd() % 3], layout = (int)(rnd() % 2);
        const int32_t ntx = tiles_along(w, tile, sem), nty = tiles_along(h, tile, sem);
        int32_t win[4] = {0, -1, 0, -1};
        switch (rnd() % 5) {
            case 0: break;                                            // the default: the whole frame
            case 1: win[0] = win[1] = (int32_t)rnd_in(0, ntx); break;  // an empty window, possibly at tile 0
            default:
                win[0] = (int32_t)rnd_in(-2, (int64_t)ntx + 2); win[1] = (int32_t)rnd_in(-2, (int64_t)ntx + 2);
                win[2] = (int32_t)rnd_in(-2, (int64_t)nty + 2); win[3] = (int32_t)rnd_in(-2, (int64_t)nty + 2);
        }
        const bool huge = (int64_t)w * h > (int64_t)1 << 40;   // rectangle areas still fit int64; nothing else is formed
        (void)huge;
        check_plan(w, h, tile, sem, layout, win, rnd() % 2 == 0);
    }
    // an empty window that starts at tile 0 renders nothing (it used to mean "to the end")
    {
        GsxParams prm;
        default_params(&prm);
        prm.tile_x0 = 0; prm.tile_x1 = 0;
        float dummy;
        Plan p;
        char msg[256];
        CHECK(make_plan(1920, 1080, 16, &dummy, &prm, p, msg, sizeof msg) == GSX_OK && p.grid.count() == 0);
        prm.tile_x1 = -1;
        CHECK(make_plan(1920, 1080, 16, &dummy, &prm, p, msg, sizeof msg) == GSX_OK && p.grid.count() == 119 * 67);
        CHECK(make_plan(0, 1080, 16, &dummy, &prm, p, msg, sizeof msg) == GSX_ERR_INVALID_ARGUMENT);
        CHECK(make_plan(1920, 1080, 0, &dummy, &prm, p, msg, sizeof msg) == GSX_ERR_INVALID_ARGUMENT);
        CHECK(make_plan(1920, 1080, 16, nullptr, &prm, p, msg, sizeof msg) == GSX_ERR_INVALID_ARGUMENT);
        prm.tile_x0 = 0; prm.tile_x1 = 10; prm.out_w = 16; prm.out_h = 1080;       // window wider than the strip buffer
        CHECK(make_plan(1920, 1080, 16, &dummy, &prm, p, msg, sizeof msg) == GSX_ERR_INVALID_ARGUMENT);
    }
    printf("ok: %lld checks\n", g_checks);
    return 0;
}
