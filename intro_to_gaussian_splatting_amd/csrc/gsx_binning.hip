// Depth ordering and tile binning on gfx950.
//
// The reference sorts all visible Gaussians once by view depth (splat/gaussian_scene.py:117) and
// then, for every tile, boolean-masks that sorted list (:209-218), so each tile's list is in
// global depth order.  Here the same lists are produced with two stable radix sorts:
//   1. N keys  (depth bits, value = original index)  -> depth rank of every Gaussian that reaches a
//      tile of the window (gsx_sort.hip: the first pass drops the others, the last pass leaves the
//      tile rectangles in rank order);
//   2. D keys  (window-local tile id, value = Gaussian index), emitted in rank order, so a
//      STABLE sort on the tile id alone (13 bits at 1080p, 2 radix passes instead of 6 for a
//      64-bit tile|depth key) leaves every tile's entries in depth order, ties broken by
//      original index exactly like a stable argsort.
// The radix sort (gsx_sort.hip) and the scan below are ours; no library primitive is left on the
// path.  D never leaves the device: every kernel after the scan reads it from device memory and is
// launched on a grid sized by the workspace capacity.
//
// Between the two sorts (all reads coalesced, no per-rank offset array):
//   chunk sums   the tile counts of every 1024 consecutive depth ranks (from the rank-ordered rectangles), 64
//                bit: left behind by the last kernel of the sampled depth sort (gsx_sort.hip, up to 2^20
//                Gaussians), by chunk_sums_kernel otherwise;
//   emit         every workgroup (same chunking) adds up the sums before its chunk, scans its own 1024
//                counts in LDS and writes its (tile id, Gaussian index) pairs in RUNS: a thread takes one run
//                of consecutive pairs, finds the Gaussian of the first one by binary search in the LDS offsets
//                and walks from there (rows, columns, Gaussians), so stores are wide and contiguous whatever
//                the mix of footprints is (one frame-filling splat is spread over the whole workgroup by
//                construction).
// After the second sort: tile_ranges_kernel ([first, last) of every tile's list, long tiles flagged) and, for
// large scenes, tile_schedule_kernel (the tiles by falling list length, for the compositing launch).

#include "gsx_internal.h"
#include "gsx_schedule_device.h"

namespace gsx {
namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 4;
constexpr int kChunk = kBlock * kPerThread;   // depth ranks per workgroup in chunk_sums / emit
static_assert(kChunk == kEmitChunk, "gsx_plan.h sizes the chunk sums with this");
// Up to this many chunks (2M Gaussians) the emit kernel adds up the raw chunk sums itself; beyond,
// scan_sums_kernel turns them into prefixes first (one more launch where frames take milliseconds).
constexpr int kSelfScanChunks = 2048;

__device__ __forceinline__ uint32_t load_count(const uint32_t *n_dev, uint32_t bound) {
    if (!n_dev) return bound;
    const uint32_t n = *n_dev;
    return n < bound ? n : bound;
}

__device__ __forceinline__ uint32_t tiles_of(const TileRect &r) {
    return r.x0 > r.x1 ? 0u : (uint32_t)(r.x1 - r.x0 + 1) * (uint32_t)(r.y1 - r.y0 + 1);
}

// A thread's four consecutive rectangles: two 16-byte loads.
__device__ __forceinline__ void load_rects(const TileRect *__restrict__ rrect, uint32_t first, uint32_t m,
                                           TileRect (&r)[kPerThread]) {
    if (first + kPerThread <= m) {
        const uint4 *src = reinterpret_cast<const uint4 *>(rrect + first);   // first is a multiple of 4: 32-B aligned
        const uint4 a = src[0], b = src[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            r[k].x0 = (uint16_t)(w[2 * k] & 0xFFFFu);
            r[k].x1 = (uint16_t)(w[2 * k] >> 16);
            r[k].y0 = (uint16_t)(w[2 * k + 1] & 0xFFFFu);
            r[k].y1 = (uint16_t)(w[2 * k + 1] >> 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            if (first + k < m) {
                r[k] = rrect[first + k];
            } else {
                r[k].x0 = 1; r[k].x1 = 0; r[k].y0 = 1; r[k].y1 = 0;
            }
        }
    }
}

__device__ __forceinline__ uint64_t block_sum64(uint64_t v, uint64_t *wsum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_down((long long)v, o);
    if (lane == 0) wsum[w] = v;
    __syncthreads();
    const uint64_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return total;
}

// sums[c] = tile instances of depth ranks [1024 c, 1024 (c+1)).
__global__ void __launch_bounds__(kBlock)
    chunk_sums_kernel(const TileRect *__restrict__ rrect, const uint32_t *__restrict__ m_dev, uint32_t bound,
                      uint64_t *__restrict__ sums) {
    __shared__ uint64_t wsum[4];
    const uint32_t m = load_count(m_dev, bound);
    const uint32_t first = blockIdx.x * (uint32_t)kChunk + threadIdx.x * (uint32_t)kPerThread;
    uint64_t mine = 0;
    if (first < m) {
        TileRect r[kPerThread];
        load_rects(rrect, first, m, r);
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) mine += tiles_of(r[k]);
    }
    const uint64_t total = block_sum64(mine, wsum);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// Large inputs only: sums[0 .. nchunks] <- exclusive prefix (entry nchunks = grand total), one workgroup.
// Every thread owns a run of consecutive entries (all loaded before the first is needed), the runs' totals are
// scanned with wave shuffles.
__global__ void __launch_bounds__(kBlock) scan_sums_kernel(uint64_t *__restrict__ sums, int nchunks) {
    __shared__ uint64_t wsum[kBlock / 64];
    constexpr int kKeep = 32;                                   // entries a thread keeps in registers (8192 chunks = 8M Gaussians)
    const int per = (nchunks + kBlock - 1) / kBlock;
    const int i0 = threadIdx.x * per, i1 = min(nchunks, i0 + per);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t v[kKeep], mine = 0;
#pragma unroll
    for (int k = 0; k < kKeep; ++k) {
        v[k] = (k < per && i0 + k < i1) ? sums[i0 + k] : 0ull;
        mine += v[k];
    }
    for (int i = i0 + kKeep; i < i1; ++i) mine += sums[i];      // (beyond 8M Gaussians: read again below)
    uint64_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = (uint64_t)__shfl_up((long long)x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint64_t run = x - mine;
    for (int k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
    for (int k = 0; k < kKeep; ++k) {
        if (k < per && i0 + k < i1) {
            sums[i0 + k] = run;
            run += v[k];
        }
    }
    for (int i = i0 + kKeep; i < i1; ++i) {
        const uint64_t u = sums[i];
        sums[i] = run;
        run += u;
    }
    if (i0 < nchunks && i1 == nchunks) sums[nchunks] = run;   // exactly one thread owns the last entry
}

// counts: what the frame reports and what the tile sort reads on the device.
//   stats2[0] = visible Gaussians, stats2[1] = D as int64 (the first two fields of a GsxFrameStats), stats2[2] =
//   Gaussians kept by the depth sort;
//   *d32 = min(D, 2^32 - 1): element count of the tile sort.
struct EmitCounts {
    int64_t *stats2;
    int64_t *stats2_host;        // device-visible alias of pinned host memory, or null
    uint32_t *d32;
    uint32_t *long_count;
    uint32_t *redo_count;
    const uint32_t *culled_dev;  // Gaussians behind the cull plane (whole-path entry), or null
    int64_t n_total;             // n_visible = n_total - *culled_dev
};

// What the walk below needs of one Gaussian of the chunk, in one 16-byte LDS word.
struct EmitSlot {
    uint32_t xs, ys;   // x0 | x1 << 16, y0 | y1 << 16 (tile rectangle)
    uint32_t index;    // Gaussian index (the value of its pairs)
    uint32_t end;      // local offset of its last pair + 1 (= offs[g + 1])
};

template <typename Key, bool PREFIXED>
__global__ void __launch_bounds__(kBlock)
    emit_kernel(const TileRect *__restrict__ rrect, const uint32_t *__restrict__ order,
                const uint32_t *__restrict__ m_dev, uint32_t bound, const uint64_t *__restrict__ sums, int nchunks,
                TileGrid g, uint32_t limit, Key *__restrict__ keys, uint32_t *__restrict__ vals,
                uint2 *__restrict__ ranges, EmitCounts ec, uint32_t eager) {
    __shared__ uint64_t wsum[4];
    __shared__ uint32_t offs[kChunk + 1];
    __shared__ __attribute__((aligned(16))) EmitSlot slot[kChunk + 2];
    {   // tiles without pairs keep an empty range
        const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (t < g.count()) ranges[t] = make_uint2(0u, 0u);
    }
    if ((int)blockIdx.x >= nchunks) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * (uint32_t)kChunk + threadIdx.x * (uint32_t)kPerThread;

    // ---- everything this workgroup reads from memory is requested at once -- the kept count, the chunk sums
    //      before this chunk, its rectangles and indices (bounded by the host-known `bound`; what lies beyond
    //      the kept count is masked afterwards): one trip to memory instead of three dependent ones.  With one
    //      workgroup per 1024 ranks and all of them resident the kernel's duration IS that chain.
    // A workgroup whose ranks lie below `eager` -- the number of kept Gaussians the caller expects (GsxParams.kept_hint,
    // with a margin; without a hint: all n) -- requests its rectangles and indices before it knows the kept count; one
    // beyond reads the count first and, nearly always, finds it has nothing to fetch: on a rank that owns 1/8 of the
    // frame the eager form read 60 MB of rectangles and indices where 7.7 MB are in use (round 3, PMC).
    TileRect r[kPerThread];
    uint32_t gidx[kPerThread];
    const bool early = blockIdx.x * (uint32_t)kChunk < eager;
    uint32_t m = 0;
    if (!early) m = load_count(m_dev, bound);
    const uint32_t fetch_bound = early ? bound : m;
    load_rects(rrect, first, fetch_bound, r);
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) gidx[k] = first + k < fetch_bound ? (order ? order[first + k] : first + k) : 0u;
    if (early) m = load_count(m_dev, bound);
    uint64_t base;
    const bool last = (int)blockIdx.x == nchunks - 1;
    if (PREFIXED) {
        base = sums[blockIdx.x];
    } else {
        uint64_t before = 0;
        for (int k = threadIdx.x; k < (int)blockIdx.x; k += kBlock) before += sums[k];
        base = block_sum64(before, wsum);
    }

    // ---- this chunk's counts and local offsets
    uint32_t c[kPerThread];
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        c[k] = first + k < m ? tiles_of(r[k]) : 0u;
        mine += c[k];
    }
    uint64_t x = mine;  // inclusive scan over the wave, then over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = (uint64_t)__shfl_up((long long)x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint64_t run = x - mine;
    for (int k = 0; k < w; ++k) run += wsum[k];
    const uint64_t chunk_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // local offsets saturate at 2^32 - 1: whatever lies beyond is beyond the pair capacity anyway
    auto sat = [](uint64_t v) -> uint32_t { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; };
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        offs[threadIdx.x * kPerThread + k] = sat(run);
        run += c[k];
        slot[threadIdx.x * kPerThread + k] = EmitSlot{(uint32_t)r[k].x0 | ((uint32_t)r[k].x1 << 16),
                                                      (uint32_t)r[k].y0 | ((uint32_t)r[k].y1 << 16), gidx[k], sat(run)};
    }
    if (threadIdx.x == kBlock - 1) offs[kChunk] = sat(run);
    if (threadIdx.x < 2) slot[kChunk + threadIdx.x] = EmitSlot{0u, 0u, 0u, 0xFFFFFFFFu};   // the walk stops here
    __syncthreads();

    if (last && threadIdx.x == 0) {
        const uint64_t d = PREFIXED ? sums[nchunks] : base + chunk_total;
        const int64_t nvis = ec.n_total - (ec.culled_dev ? (int64_t)*ec.culled_dev : 0);
        ec.stats2[0] = nvis;
        ec.stats2[1] = (int64_t)d;
        ec.stats2[2] = (int64_t)m;              // Gaussians the depth sort kept (GsxFrameStats.n_kept)
        if (ec.stats2_host) {                    // a pinned GsxFrameStats: n_visible, n_instances, ..., n_kept at byte 56
            ec.stats2_host[0] = nvis;
            ec.stats2_host[1] = (int64_t)d;
            ec.stats2_host[7] = (int64_t)m;
        }
        *ec.d32 = d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d;
        *ec.long_count = 0u;
        *ec.redo_count = 0u;
    }

    // ---- the chunk's pairs, in groups of 8 consecutive pairs aligned in the GLOBAL pair index (8 keys and 8 values
    //      are whole 16-byte words: wide stores, every byte written once).  A thread takes groups t, t + 256, ...: for
    //      each it finds the Gaussian of the first pair by binary search (largest rank with offs <= p) and walks --
    //      next row of the rectangle, next column, next Gaussian (one 16-byte LDS word, fetched one Gaussian ahead)
    //      -- at ~8 instructions per pair.  (A search and an integer division per pair, as in the first version of
    //      this kernel, were ~100.)  Neighbouring lanes hold neighbouring groups, so ONE store instruction of a wave
    //      covers 1 KB of keys / 2 x 1 KB of values without a gap.  Round 2 gave every thread one run of 8 L
    //      consecutive pairs (one search instead of L): a store instruction then wrote 16 of every 16 L bytes, the rest
    //      of each sector followed an iteration of the walk later -- too late to be merged on the way out: 39.0 MB
    //      reached HBM for 25.3 (1M Gaussians, PMC WRITE_SIZE), 195 MB for 128 at 5M, where the kernel is
    //      bandwidth-bound.  Now 24.9 and 125 MB; 17.7 -> 15.0 us and 65.8 -> 41.4 us (same box, rocprofv3).
    if (base >= (uint64_t)limit) return;   // speculative mode: the pair count exceeded the caller's hint
    const uint64_t room = (uint64_t)limit - base;
    const uint32_t npairs = (uint32_t)(chunk_total < room ? chunk_total : room);
    if (npairs == 0) return;
    const uint32_t out0 = (uint32_t)base, out1 = out0 + npairs;      // limit < 2^32: no wrap
    const uint32_t nwy = (uint32_t)g.nwy();
    const uint32_t q_first = out0 & ~7u;
    const uint32_t groups = (out1 - q_first + 7u) >> 3;
    const uint4 *slots = reinterpret_cast<const uint4 *>(slot);
    for (uint32_t gi = threadIdx.x; gi < groups; gi += (uint32_t)kBlock) {
        const uint32_t qb = q_first + gi * 8u;
        const uint32_t qa = max(qb, out0), qe = min(qb + 8u, out1);   // this group's pairs: global [qa, qe)
        uint32_t p = qa - out0;
        uint32_t lo = 0, hi = kChunk;            // invariant: offs[lo] <= p < offs[hi]
#pragma unroll
        for (int s_ = 0; s_ < 10; ++s_) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool right = offs[mid] <= p;
            lo = right ? mid : lo;
            hi = right ? hi : mid;
        }
        uint4 cur = slots[lo], ahead = slots[lo + 1];     // (xs, ys, index, end)
        uint32_t x0 = cur.x & 0xFFFFu, y0 = cur.y & 0xFFFFu, y1 = cur.y >> 16;
        uint32_t tx, ty;
        {
            const uint32_t k = p - offs[lo], h = y1 - y0 + 1u;
            tx = x0 + k / h;
            ty = y0 + k % h;
        }
        uint32_t key = (tx - (uint32_t)g.wx0) * nwy + (ty - (uint32_t)g.wy0);
        for (uint32_t q8 = qb; q8 < qe; q8 += 8u) {
            uint32_t kk[8], vv[8];
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t q = q8 + i;
                kk[i] = key;
                vv[i] = cur.z;
                if (q >= qa && q + 1u < qe) {        // step to pair p + 1 (there is one)
                    ++p;
                    if (p >= cur.w) {                // the next Gaussian that has tiles
                        do {
                            cur = ahead;
                            ++lo;
                            ahead = slots[lo + 1];
                        } while (p >= cur.w);
                        x0 = cur.x & 0xFFFFu;
                        y0 = cur.y & 0xFFFFu;
                        y1 = cur.y >> 16;
                        tx = x0;
                        ty = y0;
                        key = (tx - (uint32_t)g.wx0) * nwy + (ty - (uint32_t)g.wy0);
                    } else if (ty == y1) {           // next column of the rectangle
                        ty = y0;
                        ++tx;
                        key += nwy - (y1 - y0);
                    } else {
                        ++ty;
                        ++key;
                    }
                }
            }
            if (q8 >= qa && q8 + 8u <= qe) {
                if (sizeof(Key) == 2) {
                    *reinterpret_cast<uint4 *>(keys + q8) =
                        make_uint4(kk[0] | (kk[1] << 16), kk[2] | (kk[3] << 16), kk[4] | (kk[5] << 16), kk[6] | (kk[7] << 16));
                } else {
                    uint4 *dk = reinterpret_cast<uint4 *>(keys + q8);
                    dk[0] = make_uint4(kk[0], kk[1], kk[2], kk[3]);
                    dk[1] = make_uint4(kk[4], kk[5], kk[6], kk[7]);
                }
                uint4 *dv = reinterpret_cast<uint4 *>(vals + q8);
                dv[0] = make_uint4(vv[0], vv[1], vv[2], vv[3]);
                dv[1] = make_uint4(vv[4], vv[5], vv[6], vv[7]);
            } else {                                  // the ragged first / last 8 of the chunk
#pragma unroll
                for (uint32_t i = 0; i < 8u; ++i) {
                    const uint32_t q = q8 + i;
                    if (q >= qa && q < qe) {
                        keys[q] = (Key)kk[i];
                        vals[q] = vv[i];
                    }
                }
            }
        }
    }
}

// ranges[t] = [first, last+1) of tile t inside the tile-sorted pair list; untouched (zeroed by
// the emit kernel) for tiles with no entries.  8 consecutive keys per thread.
// The grid covers the workspace capacity; the true pair count is read from device memory.
// The thread that closes a tile's run also decides whether the tile is LONG (more entries than
// long_tile_threshold: the key that many positions back is still this tile's), appends it to lt.list and
// flags it in bit 31 of ranges[t].y -- see LongTiles in gsx_internal.h.
template <typename Key>
__global__ void __launch_bounds__(kBlock)
    tile_ranges_kernel(const Key *__restrict__ keys, const uint32_t *__restrict__ d_dev, uint32_t cap,
                       uint2 *__restrict__ ranges, LongTiles lt, uint32_t ntiles) {
    constexpr int kPer = 8;
    const uint32_t d = min(*d_dev, cap);
    const uint32_t j0 = (blockIdx.x * (uint32_t)kBlock + threadIdx.x) * kPer;
    if (j0 >= d) return;
    const uint32_t long_len = long_tile_threshold(d, ntiles);
    // By cost (see below): the threshold is put together from the hints' header HERE, with the first keys' loads in
    // flight, not where a thread closes a tile -- one trip to memory less on the path of a kernel that is all latency
    // (measured: 7.2 instead of 5.3 us at 1M Gaussians with the header read at the point of use).
    bool by_cost = false;
    uint32_t cost_thr = 0xFFFFFFFFu;
    if (lt.max && lt.cost && lt.header) {
        uint32_t total = 0;
#pragma unroll
        for (int xcd = 0; xcd < 8; ++xcd) total += lt.header[kHintXcdCost + xcd];
        by_cost = lt.header[kHintLens] == ntiles && lt.header[kHintSched] == ntiles;
        // (the percentage: the caller's, or what the compositing launch has raised it to -- see gsx_plan.h)
        const uint32_t pct = max(lt.cost_pct, lt.header[kHintLongPct]);
        cost_thr = max(256u, (uint32_t)(((uint64_t)(total >> 10) * pct) / 100u));
    }
    Key k[kPer + 2];                 // k[0] = the key before this thread's run, k[kPer + 1] = the one after it
    k[0] = j0 > 0 ? keys[j0 - 1] : (Key)0;
    const uint32_t cnt = min((uint32_t)kPer, d - j0);
    if (cnt == kPer && sizeof(Key) == 2) {
        const uint4 q = *reinterpret_cast<const uint4 *>(keys + j0);
        const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            k[1 + 2 * e] = (Key)(w4[e] & 0xFFFFu);
            k[2 + 2 * e] = (Key)(w4[e] >> 16);
        }
    } else {
#pragma unroll
        for (int e = 0; e < kPer; ++e) k[1 + e] = (uint32_t)e < cnt ? keys[j0 + e] : (Key)0;
    }
    k[kPer + 1] = j0 + kPer < d ? keys[j0 + kPer] : (Key)0;
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
        if ((uint32_t)e >= cnt) break;
        const uint32_t j = j0 + e;
        const Key cur = k[e + 1];
        if (j == 0 || k[e] != cur) ranges[cur].x = j;
        if (j == d - 1 || k[e + 2] != cur) {           // j closes the run of tile `cur`
            uint32_t end = j + 1;
            bool is_long = !by_cost && lt.max && j >= long_len && keys[j - long_len] == cur;
            if (by_cost) {
                // What the tile COST last time decides (records staged until it was done: a dense tile that saturates
                // early is cheap however long its list), against a SIMD's share of the whole frame: a wave that is
                // alone on its SIMD at the end of the launch walks ~3x slower than the SIMD's eight waves together, so
                // a tile above a third of that share is still running when everything else has finished.  (By length,
                // 4x the mean: the heavy-tailed test scene ended with ~100 single waves of 1 500 .. 2 500 records each
                // running alone for 150 us; lowering the length threshold instead put 500 tiles on four waves, most
                // of them cheap, and cost 50 %.)  A long tile's cost is the largest of its four helpers' -- about what one
                // wave would have walked; a tile that was long stays long down to lt.stay_pct % of the threshold, so that
                // a tile does not change sides from frame to frame (LongTiles in gsx_internal.h).
                const uint32_t was = lt.cost[cur];
                const uint32_t thr_now = (was >> 31) ? (uint32_t)(((uint64_t)cost_thr * lt.stay_pct) / 100u) : cost_thr;
                is_long = (was & 0x7FFFFFFFu) >= thr_now && j >= 127u && keys[j - 127u] == cur;     // (and 128 entries now)
            }
            if (is_long) {
                const uint32_t slot = atomicAdd(lt.count, 1u);
                if (slot < lt.max) {
                    lt.list[slot] = (uint32_t)cur;
                    end |= kLongFlag;
                    if (lt.cost) lt.cost[cur] = 0x80000000u;      // the helpers raise it to their largest cost
                }
            }
            ranges[cur].y = end;
        }
    }
}

__global__ void __launch_bounds__(kBlock)
    tile_counts_kernel(const uint2 *__restrict__ ranges, int64_t nt, uint32_t *__restrict__ counts) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t < nt) counts[t] = (ranges[t].y & ~kLongFlag) - ranges[t].x;
}

// sched[k] = the tile with the k-th longest list (ties and near-ties in any order: lists are ranked by
// 1024 length classes between the shortest and the longest list of the frame).  The compositing kernel
// hands the tiles out so that every SIMD of the chip gets the same share of every length class (see
// blend_tile16_kernel): with the tiles in index order the busiest SIMD of a 1M-Gaussian 1080p frame had 11 %
// more list entries than the average one and the kernel waited for it (298 -> 261 us).  Long tiles
// (composited by helper workgroups) count as empty.  One workgroup; a thread keeps its first 32 lengths in
// registers (all loads in flight at once; a loop of dependent trips to memory cost 1 us per 1024 tiles).  (Run by the last workgroup of tile_ranges_kernel instead, this cost 100+ us: every workgroup of
// that kernel then needs a device-scope release fence, an L2 write-back on this part.)
constexpr int kSchedThreads = 1024, kSchedClasses = 1024, kSchedKeep = 32;   // 32 768 tiles (a 4K frame) in registers
__global__ void __launch_bounds__(kSchedThreads)
    tile_schedule_kernel(const uint2 *__restrict__ ranges, uint32_t nt, uint32_t *__restrict__ sched) {
    __shared__ uint32_t hist[kSchedClasses];
    __shared__ uint32_t wsum[kSchedThreads / 64];
    __shared__ uint32_t s_min, s_max;
    // the order is put together in LDS and written out in one coalesced sweep: 4-byte stores scattered straight
    // to memory are one 64-byte transaction each, and a single CU issues about one per clock (15 us at 32 768 tiles)
    __shared__ uint32_t staged[kSchedKeep * kSchedThreads];
    const bool stage = nt <= (uint32_t)(kSchedKeep * kSchedThreads);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto length = [&](uint32_t t) -> uint32_t {
        const uint2 r = ranges[t];
        return (r.y & kLongFlag) ? 0u : r.y - r.x;
    };
    uint32_t len[kSchedKeep];
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
    for (int k = 0; k < kSchedKeep; ++k) {
        const uint32_t t = (uint32_t)k * kSchedThreads + threadIdx.x;
        len[k] = t < nt ? length(t) : 0u;
        if (t < nt) {
            mn = min(mn, len[k]);
            mx = max(mx, len[k]);
        }
    }
    for (uint32_t t = (uint32_t)kSchedKeep * kSchedThreads + threadIdx.x; t < nt; t += kSchedThreads) {
        const uint32_t l = length(t);
        mn = min(mn, l);
        mx = max(mx, l);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    }
    hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        s_min = 0xFFFFFFFFu;
        s_max = 0;
    }
    __syncthreads();
    if (lane == 0 && mn <= mx) {
        atomicMin(&s_min, mn);
        atomicMax(&s_max, mx);
    }
    __syncthreads();
    const uint32_t shortest = s_min;
    const float per_entry = (float)(kSchedClasses - 1) / (float)max(s_max - shortest, 1u);
    auto cls = [&](uint32_t l) -> uint32_t {   // class 0 = the longest lists
        return (uint32_t)(kSchedClasses - 1) - min((uint32_t)((float)(l - shortest) * per_entry), (uint32_t)(kSchedClasses - 1));
    };
#pragma unroll
    for (int k = 0; k < kSchedKeep; ++k)
        if ((uint32_t)k * kSchedThreads + threadIdx.x < nt) atomicAdd(&hist[cls(len[k])], 1u);
    for (uint32_t t = (uint32_t)kSchedKeep * kSchedThreads + threadIdx.x; t < nt; t += kSchedThreads)
        atomicAdd(&hist[cls(length(t))], 1u);
    __syncthreads();
    const uint32_t mine = hist[threadIdx.x];
    uint32_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < w; ++k) before += wsum[k];
    hist[threadIdx.x] = before + x - mine;   // first slot of this class
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSchedKeep; ++k) {
        const uint32_t t = (uint32_t)k * kSchedThreads + threadIdx.x;
        if (t < nt) (stage ? staged : sched)[atomicAdd(&hist[cls(len[k])], 1u)] = t;
    }
    for (uint32_t t = (uint32_t)kSchedKeep * kSchedThreads + threadIdx.x; t < nt; t += kSchedThreads)
        sched[atomicAdd(&hist[cls(length(t))], 1u)] = t;
    if (stage) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nt; k += kSchedThreads) sched[k] = staged[k];
    }
}

// A frame WITHOUT a schedule from an earlier frame of its view (REF_CPU, tile 16): the per-XCD schedule the hinted frames get
// from the projection launch's spare workgroups (gsx_schedule_device.h: an XCD's tiles -- ~32-tile chunks dealt round robin
// -- by falling cost, handed to its 128 SIMDs round by round), here from THIS frame's list lengths, eight workgroups behind
// tile_ranges_kernel.  Round 6: replaces the one-workgroup ranking of the whole frame on this path -- 10.1 -> ~4 us, and the
// compositing launch keeps its tiles' neighbours in one L2 (208 -> ~200 us at 1M Gaussians): a view's first frame is what
// `value_cold_frame` reports.
__global__ void __launch_bounds__(kBlock) tile_schedule_xcd_kernel(const uint2 *__restrict__ ranges, SchedJob job) {
    schedule_xcd(job, blockIdx.x, [&](uint32_t t) -> uint32_t {
        const uint2 r = ranges[t];
        const uint32_t len = min((r.y & ~kLongFlag) - r.x, 0x7FFFFFFFu);
        return (r.y & kLongFlag) ? (len | 0x80000000u) : len;
    });
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

hipError_t launch_tile_schedule_xcd(const uint2 *ranges, int64_t nt, int64_t nwy, uint32_t *sched, uint32_t *header, hipStream_t s) {
    if (nt <= 0) return hipSuccess;
    const SchedJob job{nullptr, sched, header, (uint32_t)nt, (uint32_t)nwy, sched_cap((uint32_t)nt, (uint32_t)nwy)};
    tile_schedule_xcd_kernel<<<kSchedXcds, kBlock, 0, s>>>(ranges, job);
    return hipGetLastError();
}

hipError_t launch_tile_schedule(const uint2 *ranges, int64_t nt, uint32_t *sched, hipStream_t s) {
    if (nt <= 0) return hipSuccess;
    tile_schedule_kernel<<<1, kSchedThreads, 0, s>>>(ranges, (uint32_t)nt, sched);
    return hipGetLastError();
}

hipError_t launch_tile_counts(const uint2 *ranges, int64_t nt, uint32_t *counts, hipStream_t s) {
    if (nt <= 0) return hipSuccess;
    tile_counts_kernel<<<blocks_for(nt), kBlock, 0, s>>>(ranges, nt, counts);
    return hipGetLastError();
}

static uint64_t *sums_of(void *temp, int64_t n, int64_t cap) {
    return (uint64_t *)((char *)temp + binning_sums_offset(n, cap));
}

// (tile id, Gaussian index) pairs in rank order.  Tile ids fit 16 bits for any frame up to 65536 tiles
// (4K has 32 026), which halves the key traffic of the sort; larger frames use 32-bit ids.
// keys0 / vals0 hold `cap` 32-bit words each; the pair count D stays on the device (*bc.d32), pairs
// beyond `cap` are dropped (the caller compares D with cap afterwards).
template <typename Key>
hipError_t emit_impl(void *temp, const TileRect *rrect, const uint32_t *order, const uint32_t *m_dev, int64_t n,
                     int64_t cap, const TileGrid &grid, void *keys0, uint32_t *vals0, uint2 *ranges, const BinCounts &bc,
                     bool sums_ready, uint32_t eager, hipStream_t s) {
    const int64_t nt = grid.count();
    const int nchunks = (int)((n + kChunk - 1) / kChunk);
    uint64_t *sums = sums_of(temp, n, cap);
    if (!sums_ready) chunk_sums_kernel<<<nchunks, kBlock, 0, s>>>(rrect, m_dev, (uint32_t)n, sums);
    const EmitCounts ec{bc.stats2, bc.stats2_host, bc.d32, bc.long_count, bc.redo_count, bc.culled_dev, bc.n_total};
    const unsigned tiles_grid = blocks_for(nt);
    const unsigned egrid = (unsigned)nchunks > tiles_grid ? (unsigned)nchunks : tiles_grid;
    if (nchunks <= kSelfScanChunks) {
        emit_kernel<Key, false><<<egrid, kBlock, 0, s>>>(rrect, order, m_dev, (uint32_t)n, sums, nchunks, grid,
                                                         (uint32_t)cap, (Key *)keys0, vals0, ranges, ec, eager);
    } else {
        scan_sums_kernel<<<1, kBlock, 0, s>>>(sums, nchunks);
        emit_kernel<Key, true><<<egrid, kBlock, 0, s>>>(rrect, order, m_dev, (uint32_t)n, sums, nchunks, grid,
                                                        (uint32_t)cap, (Key *)keys0, vals0, ranges, ec, eager);
    }
    return hipGetLastError();
}

// Stable sort of the pairs by tile id + every tile's [first, last) range.
template <typename Key>
hipError_t sort_impl(void *temp, int64_t cap, void *keys0, void *keys1, uint32_t *vals0, uint32_t *vals1, uint2 *ranges,
                     int key_bits, const uint32_t *d32, const LongTiles &lt, uint32_t ntiles,
                     const uint32_t **sorted_vals, hipStream_t s) {
    Key *kc = (Key *)keys0, *ka = (Key *)keys1;
    uint32_t *vc = vals0, *va = vals1;
    hipError_t e;
    if (sizeof(Key) == 2)
        e = radix_sort_pairs_u16(temp, (uint16_t *&)kc, (uint16_t *&)ka, vc, va, d32, cap, key_bits, s);
    else
        e = radix_sort_pairs_u32(temp, (uint32_t *&)kc, (uint32_t *&)ka, vc, va, d32, cap, key_bits, s);
    if (e != hipSuccess) return e;
    tile_ranges_kernel<Key><<<blocks_for((cap + 7) / 8), kBlock, 0, s>>>(kc, d32, (uint32_t)cap, ranges, lt, ntiles);
    *sorted_vals = vc;
    return hipGetLastError();
}

hipError_t emit_instances(void *temp, const TileRect *rrect, const uint32_t *order, const uint32_t *m_dev, int64_t n,
                          int64_t cap, const TileGrid &grid, void *keys0, uint32_t *vals0, uint2 *ranges,
                          const BinCounts &bc, bool sums_ready, int64_t kept_hint, hipStream_t s) {
    if (n <= 0) return hipErrorInvalidValue;   // callers handle the empty scene themselves
    // ranks expected to exist: the hint with 2 % + a chunk of margin (a larger count is still handled, one trip later)
    const int64_t expect = kept_hint > 0 && kept_hint < n ? kept_hint + kept_hint / 50 + kChunk : n;
    const uint32_t eager = (uint32_t)(expect < n ? expect : n);
    if (grid.count() <= 65536)
        return emit_impl<uint16_t>(temp, rrect, order, m_dev, n, cap, grid, keys0, vals0, ranges, bc, sums_ready, eager, s);
    return emit_impl<uint32_t>(temp, rrect, order, m_dev, n, cap, grid, keys0, vals0, ranges, bc, sums_ready, eager, s);
}

uint64_t *emit_chunk_sums(void *temp, int64_t n, int64_t cap) { return sums_of(temp, n, cap); }

hipError_t sort_instances(void *temp, int64_t cap, const TileGrid &grid, void *keys0, void *keys1, uint32_t *vals0,
                          uint32_t *vals1, uint2 *ranges, const uint32_t *d32, const LongTiles &lt,
                          const uint32_t **sorted_vals, hipStream_t s) {
    *sorted_vals = vals0;
    if (cap <= 0) return hipSuccess;           // nothing fits: every range stays empty
    const int64_t nt = grid.count();
    int bits = 1;
    while (((int64_t)1 << bits) < nt) ++bits;
    if (nt <= 65536)
        return sort_impl<uint16_t>(temp, cap, keys0, keys1, vals0, vals1, ranges, bits, d32, lt, (uint32_t)nt, sorted_vals, s);
    return sort_impl<uint32_t>(temp, cap, keys0, keys1, vals0, vals1, ranges, bits, d32, lt, (uint32_t)nt, sorted_vals, s);
}

}  // namespace gsx
