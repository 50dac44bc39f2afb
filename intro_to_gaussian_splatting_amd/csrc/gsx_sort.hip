// Stable LSD radix sort of (key, value) pairs for the two orderings of the frame pipeline:
// depth rank (N 32-bit keys) and tile binning (D tile ids, whose count D exists only in device
// memory when the sort is enqueued).
//
// Why not the library sort: rocPRIM's onesweep is built for large inputs; at the sizes of this
// path (1M depth keys, ~4M 16-bit tile keys) its passes are latency bound (26.7 us per pass at 1M,
// 325 workgroups in a look-back chain) and every call adds ~7 small fill kernels (94 us of
// __amd_rocclr_fillBufferAligned per frame, rocprofv3 profiles/r1c).  It also needs the element
// count on the HOST, which forced a device->host read-back in the middle of every frame.
//
// This implementation is the classic three-kernel LSD pass (two kernels for small inputs, see
// SELF_SCAN below), sized for these inputs:
//   count    each workgroup (1024 threads) histograms FOUR consecutive 2048-item chunks side by side by the
//            pass's digit in LDS and stores table[digit][chunk .. chunk+3] as one 16-byte word per digit
//            (round 1 stored 256 scattered 4-byte words per chunk: 9x write amplification at the memory side);
//   scan     one workgroup per digit row turns its row into an exclusive prefix and records the
//            row total; every thread loads its 16 consecutive counts up front (one memory round trip);
//   scatter  each workgroup re-reads its items, ranks them STABLY inside the workgroup (the lanes of a
//            wave that hold the same digit find each other through a 64-bit LDS word per digit, per-wave
//            running digit counters in LDS, waves and rounds in item
//            order), parks them in LDS in digit-major order, and streams them out: item j of the
//            parked order goes to digit base (exclusive sum of the row totals, recomputed in LDS)
//            + row prefix + (j - first j of its digit), so neighbouring lanes write neighbouring
//            addresses.
// No inter-workgroup communication inside a kernel (the hazards of MI355X's non-coherent per-XCD
// L2s never arise), no fills, no host-side count: `n_dev` points at the element count in device
// memory and the grid is sized by a host-known upper bound; workgroups past the count exit.
//
// Two pass modes exist for the depth sort of the whole-path entry point (sort_depth_compact):
//   FIRST  pass 0 drops the Gaussians that reach no tile of the window (keys >= kEmptyKey): the
//          sort compacts while it sorts, later passes and every rank-ordered stage after it only
//          see the M Gaussians that matter (1/8 of them on a rank that owns 1/8 of the frame);
//   FINAL  the last pass knows each Gaussian's final depth rank when it streams it out, and
//          gathers its 8-byte tile rectangle into rank order on the way (the scan and the pair
//          emission then read coalesced arrays; round 1 gathered counts and rectangles by index
//          in two later kernels, 5-8x the algorithmic traffic); the sorted keys are not written.
#include <stdlib.h>

#include <type_traits>

#include "gsx_internal.h"
#include "gsx_sample_device.h"

namespace gsx {
namespace {

constexpr int kThreads = 256;                      // 4 wavefronts
constexpr int kBins = kSortBins;                   // table rows; a pass uses the first 1 << bits of them
constexpr int kRounds = 8;                         // items per lane
constexpr int kItems = kThreads * kRounds;         // a chunk: 2048 consecutive items, one scatter workgroup
constexpr int kQuad = kSortQuad;                   // chunks per count workgroup
static_assert(kItems == kSortItems, "gsx_plan.h sizes the digit table with this");
// Up to this many chunks (131 072 items) a pass has no row-scan launch: the count kernel writes the
// table chunk-major and every scatter workgroup adds up the counts before its own chunk (measured
// round 1, one frame in flight: 2 000 Gaussians 96 -> 85 us, 100 000 Gaussians 166 -> 161 us; beyond
// ~100 chunks the per-workgroup table walk costs more than the launch it saves).
constexpr int kSelfScanBlocks = 64;
// Up to this many count workgroups (4 chunks each: 512K items) a pass has no row-scan launch either: the count
// kernel also stores every workgroup's four-chunk totals and a scatter workgroup adds up the totals of the
// workgroups before its own (coalesced, 8 loads in flight) plus at most three counts of its own quad.  Measured:
// the scatter kernel gains 0.04 us per count workgroup (52 of them: 6.1 -> 8.2 us, 123: 14.5 -> 19.7 us) and the
// launch it replaces costs 4.8 us, so the tile sort of a 100 000-Gaussian frame wins 2.7 us per pass and the
// partition of 1M keys would win nothing.
constexpr int kQuadScanQuads = kSortQuadTotals;
constexpr int kScanRows = 0, kScanSelf = 1, kScanQuads = 2;   // who turns the counts into prefixes
constexpr bool kXcdChunks = true;      // scatter_kernel: an XCD's workgroups take consecutive chunks (see there)

constexpr int kModePlain = 0, kModeFirst = 1, kModeFinal = 2;
constexpr int kSamples = kSortSamples;   // sample keys of the sample-partitioned depth sort (8 per bucket)

__device__ __forceinline__ uint32_t load_count(const uint32_t *n_dev, uint32_t bound) {
    if (!n_dev) return bound;
    uint32_t n = *n_dev;
    return n < bound ? n : bound;
}

// SPLIT passes (sort_depth_sampled): the "digit" of a key is its bucket among NB - 1 sorted splitters,
// bucket(k) = #{j in 1..NB-1 : spl[j] <= k}, spl[0] = 0 -- monotone in k, equal keys share a bucket.
template <int NB>
__device__ __forceinline__ uint32_t bucket_of(const uint32_t *spl, uint32_t k) {
    uint32_t lo = 0, hi = NB;          // spl[lo] <= k < spl[hi], spl[NB] = +inf
#pragma unroll
    for (int s = NB; s > 1; s >>= 1) {
        const uint32_t mid = (lo + hi) >> 1;
        const bool right = spl[mid] <= k;
        lo = right ? mid : lo;
        hi = right ? hi : mid;
    }
    return lo;
}

// The same for N keys at once: the N searches advance in lock step, so their LDS reads are in flight together
// (eight or ten dependent reads in a row per key otherwise).
template <int NB, int N>
__device__ __forceinline__ void buckets_of(const uint32_t *spl, const uint32_t (&k)[N], uint32_t (&b)[N]) {
    uint32_t lo[N], hi[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        lo[i] = 0;
        hi[i] = NB;
    }
#pragma unroll
    for (int s = NB; s > 1; s >>= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t mid = (lo[i] + hi[i]) >> 1;
            const bool right = spl[mid] <= k[i];
            lo[i] = right ? mid : lo[i];
            hi[i] = right ? hi[i] : mid;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = lo[i];
}

// Splitters handed over by the previous frame (GsxParams.hints): used when their header word says they are there;
// otherwise 255 splitters evenly spaced over the depth codes of 0.2 .. 1000 (roughly log-uniform in depth: a
// stand-in that keeps the buckets usable for ordinary scenes -- any monotone set is CORRECT, the bucket kernel sorts
// what it is given).  hdr == nullptr: `splitters` were computed by this frame's own sample kernel.
__device__ __forceinline__ uint32_t splitter_at(const uint32_t *__restrict__ splitters, const uint32_t *__restrict__ hdr, uint32_t k) {
    if (k == 0) return 0u;
    if (hdr && hdr[kHintSplitters] != (uint32_t)kBins) {
        const uint32_t lo = 0x3E4CCCCDu, hi = 0x447A0000u;      // 0.2f, 1000.0f
        return lo + (uint32_t)(((uint64_t)(hi - lo) * k) >> 8);
    }
    return splitters[k];
}

// Each thread owns kRounds CONSECUTIVE items of one chunk (one or two 16-byte loads) -- the histogram does
// not care about order, so the count kernel reads wide; the scatter kernel needs the wave-striped order.
// CHUNK_MAJOR (small inputs): 256 threads, one chunk, table[chunk][digit] (what the self-scanning scatter
// reads, coalesced over the digits).  Otherwise: 1024 threads, four chunks side by side (one per group of
// four waves), table[digit][chunk .. chunk+3] stored as ONE 16-byte word per digit (nbp = row pitch, a
// multiple of 4).  FIRST: keys >= kEmptyKey are not counted (they are dropped by this pass) and the
// culled ones among them (== kCulledKey) are added to *culled (zeroed by an earlier kernel).
template <typename Key, bool CHUNK_MAJOR, bool FIRST, int SPLIT = 0>   // SPLIT: 0, or the number of buckets (256 / 1024)
__global__ void __launch_bounds__(CHUNK_MAJOR ? kThreads : kQuad * kThreads)
    count_kernel(const Key *__restrict__ keys, const uint32_t *__restrict__ n_dev, uint32_t bound, int shift,
                 uint32_t mask, uint32_t *__restrict__ table, int nbp, uint32_t *__restrict__ culled,
                 const uint32_t *__restrict__ splitters = nullptr, uint32_t *__restrict__ quad_totals = nullptr,
                 const uint32_t *__restrict__ hint_hdr = nullptr, uint32_t *__restrict__ samples_out = nullptr,
                 uint32_t sample_step = 0, unsigned long long *__restrict__ zero_sums = nullptr, uint32_t nsums = 0) {
    constexpr int kLanes = CHUNK_MAJOR ? 1 : kQuad;   // chunks per workgroup
    constexpr int NB = SPLIT ? SPLIT : kBins;         // histogram rows
    constexpr int kPerVec = 16 / sizeof(Key), kVecs = kRounds / kPerVec;   // 8 x u16 or 4 x u32 per 16 B
    static_assert(kRounds % kPerVec == 0, "a thread's items must fill whole 16-byte vectors");
    static_assert(NB == kBins || !CHUNK_MAJOR, "the chunk-major table (small inputs) has 256 rows");
    const int c = CHUNK_MAJOR ? 0 : (int)(threadIdx.x >> 8);
    const uint32_t t = threadIdx.x & 255u;
    // the keys are requested before the splitters are loaded -- but not before the element count is known: the
    // grid covers the capacity `bound`, which may be many times the count
    const uint32_t n = load_count(n_dev, bound);
    // With a row-scan launch behind it (large inputs) XCD x -- blockIdx % 8 == x -- counts the x-th eighth of the quads
    // that hold items: the 16 bytes a workgroup leaves in every digit row then meet the 16 bytes of its neighbours in
    // ONE L2 and leave it as a whole sector (see scatter_kernel; the quads behind them hold nothing and keep their
    // own number).  The grid is a multiple of 8 workgroups (plan_for).
    uint32_t quad = blockIdx.x;
    if (!CHUNK_MAJOR && !quad_totals && kXcdChunks) {
        const uint32_t quads = (n + (uint32_t)(kItems * kQuad) - 1u) / (uint32_t)(kItems * kQuad), per = (quads + 7u) >> 3;
        if ((blockIdx.x >> 3) < per) quad = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    const uint32_t chunk = quad * (uint32_t)kLanes + (uint32_t)c;
    const uint32_t first = chunk * (uint32_t)kItems + t * (uint32_t)kRounds;
    uint4 q[kVecs];
    if (first + kRounds <= n) {
        const uint4 *src = reinterpret_cast<const uint4 *>(keys + first);   // first is a multiple of kRounds
#pragma unroll
        for (int v = 0; v < kVecs; ++v) q[v] = src[v];
    }
    __shared__ uint32_t h[kLanes][NB];
    __shared__ uint32_t s_culled;
    __shared__ uint32_t spl[SPLIT ? SPLIT : 1];

    if (SPLIT)
        for (uint32_t k = threadIdx.x; k < (uint32_t)NB; k += blockDim.x) spl[k] = splitter_at(splitters, hint_hdr, k);   // visible after the barrier below
    if (FIRST && SPLIT && zero_sums)   // (no sample kernel ran: the chunk sums the bucket kernel adds to start from zero here)
        for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nsums; k += gridDim.x * blockDim.x) zero_sums[k] = 0ull;
    if (FIRST && sizeof(Key) == 4 && samples_out && sample_step) {
        // What the NEXT frame's splitters are made from (GsxParams.hints): kSamples regularly spaced keys, kept ones
        // where possible -- the thread whose 8 keys hold position k * step hands over the first of them, from that
        // position on, that the sort keeps (a rank's strip keeps 1 key in 8: a plain regular sample would be 7/8 void)
        const uint32_t k = (first + sample_step - 1u) / sample_step, pos = k * sample_step;
        if (k < (uint32_t)kSamples && pos < first + (uint32_t)kRounds) {
            uint32_t pick = kCulledKey;
            if (first + kRounds <= n) {
                const uint32_t w8[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[kVecs > 1 ? 1 : 0].x, q[kVecs > 1 ? 1 : 0].y,
                                        q[kVecs > 1 ? 1 : 0].z, q[kVecs > 1 ? 1 : 0].w};
#pragma unroll
                for (int e = kRounds - 1; e >= 0; --e)
                    if ((uint32_t)e >= pos - first && w8[e] < kEmptyKey) pick = w8[e];
            } else {
                for (int e = kRounds - 1; e >= 0; --e)
                    if (first + e < n && (uint32_t)e >= pos - first && (uint32_t)keys[first + e] < kEmptyKey) pick = (uint32_t)keys[first + e];
            }
            samples_out[k] = pick;
        }
    }
    auto digit = [&](uint32_t k) -> uint32_t { return SPLIT ? bucket_of<NB>(spl, k) : ((k >> shift) & mask); };
#pragma unroll
    for (int k = 0; k < NB / 256; ++k) h[c][t + 256u * k] = 0;
    if (FIRST && threadIdx.x == 0) s_culled = 0;
    __syncthreads();
    uint32_t my_culled = 0;
    if (first + kRounds <= n) {
#pragma unroll
        for (int v = 0; v < kVecs; ++v) {
            const uint32_t w4[4] = {q[v].x, q[v].y, q[v].z, q[v].w};
            uint32_t d4[4] = {0, 0, 0, 0};
            if (sizeof(Key) == 4 && SPLIT) buckets_of<NB, 4>(spl, w4, d4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (sizeof(Key) == 4) {
                    if (FIRST && w4[e] >= kEmptyKey)
                        my_culled += w4[e] == kCulledKey;
                    else
                        atomicAdd(&h[c][SPLIT ? d4[e] : digit(w4[e])], 1u);
                } else {
                    atomicAdd(&h[c][((w4[e] & 0xFFFFu) >> shift) & mask], 1u);
                    atomicAdd(&h[c][((w4[e] >> 16) >> shift) & mask], 1u);
                }
            }
        }
    } else {
        for (int r = 0; r < kRounds; ++r)
            if (first + r < n) {
                const uint32_t k = (uint32_t)keys[first + r];
                if (FIRST && k >= kEmptyKey)
                    my_culled += k == kCulledKey;
                else
                    atomicAdd(&h[c][digit(k)], 1u);
            }
    }
    if (FIRST && my_culled) atomicAdd(&s_culled, my_culled);
    __syncthreads();
    if (CHUNK_MAJOR) {
        table[(size_t)blockIdx.x * kBins + t] = h[0][t];
    } else if (threadIdx.x < NB) {
        const uint32_t d = threadIdx.x;     // (1024 threads: one digit row each, whatever NB is)
        const uint4 c4 = make_uint4(h[0][d], h[kLanes > 1 ? 1 : 0][d], h[kLanes > 2 ? 2 : 0][d], h[kLanes > 3 ? 3 : 0][d]);
        reinterpret_cast<uint4 *>(table + (size_t)d * nbp)[quad] = c4;
        if (NB == kBins && quad_totals) quad_totals[(size_t)blockIdx.x * kBins + d] = c4.x + c4.y + c4.z + c4.w;   // kScanQuads
    }
    if (FIRST && threadIdx.x == 0 && s_culled) atomicAdd(culled, s_culled);
}

// One workgroup per digit row: in-place exclusive scan of table[d][0..nbp), total -> totals[d].
// A thread owns 16 consecutive counts (4 x 16-byte loads, all in flight at once); rows longer than
// 4096 chunks (> 8M items) take further trips with a carry.
__global__ void __launch_bounds__(kThreads) row_scan_kernel(uint32_t *__restrict__ table, int nbp,
                                                            uint32_t *__restrict__ totals) {
    constexpr int kPer = 16;
    __shared__ uint32_t wave_sum[4];
    uint32_t *row = table + (size_t)blockIdx.x * nbp;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int seg = 0; seg < nbp; seg += kThreads * kPer) {
        const int i0 = seg + threadIdx.x * kPer;
        uint4 q[4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
            q[v] = i0 + 4 * v < nbp ? reinterpret_cast<const uint4 *>(row + i0)[v] : make_uint4(0u, 0u, 0u, 0u);
        uint32_t e[kPer] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w,
                            q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {   // exclusive scan inside the thread
            const uint32_t v = e[k];
            e[k] = mine;
            mine += v;
        }
        uint32_t x = mine;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up((int)x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_sum[w] = x;
        __syncthreads();
        uint32_t before = carry;
        for (int k = 0; k < w; ++k) before += wave_sum[k];
        const uint32_t seg_total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        before += x - mine;
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (i0 + 4 * v < nbp)
                reinterpret_cast<uint4 *>(row + i0)[v] = make_uint4(before + e[4 * v], before + e[4 * v + 1],
                                                                    before + e[4 * v + 2], before + e[4 * v + 3]);
        carry += seg_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// SELF_SCAN: there is no row-scan launch; `table` holds the raw per-chunk counts (chunk-major) and
// every workgroup adds up, for each digit, the counts of the chunks before it and the row total itself
// (nblocks loads per thread).  Pays for small inputs, where a pass is three launch latencies and
// the table is a few KiB.
// MODE: kModeFirst / kModeFinal, see the head of this file.  m_out (FIRST): number of items this pass
// keeps, i.e. the element count of every later pass.  rect / rrect (FINAL): per-Gaussian tile
// rectangles by index / by depth rank.
// CARRY (the LSD depth sort of a large scene, tile grids of up to 256 x 256): every item carries its tile rectangle
// along, packed into 4 bytes -- read in index order by the FIRST pass (cin unused), moved cin -> cout by the
// passes between, unpacked into rrect by the FINAL pass -- instead of the FINAL pass gathering rect[index] by
// Gaussian index: at 5M keys that gather read 128 bytes of a 40 MB table per key, 640 MB, and made the last pass 102 us
// of the sort's 220; carried, the rectangles cost 160 MB over the four passes.
template <typename Key, int SCAN, int MODE, int BITS, int SPLIT = 0, bool CARRY = false>   // SPLIT: 0, or the number of buckets (256 / 1024)
__global__ void __launch_bounds__(kThreads)
    scatter_kernel(const Key *__restrict__ kin, const uint32_t *__restrict__ vin, Key *__restrict__ kout,
                   uint32_t *__restrict__ vout, const uint32_t *__restrict__ n_dev, uint32_t bound, int shift,
                   const uint32_t *__restrict__ table, const uint32_t *__restrict__ totals, int nbp,
                   uint32_t *__restrict__ m_out, const TileRect *__restrict__ rect, TileRect *__restrict__ rrect,
                   const uint32_t *__restrict__ splitters = nullptr, const uint32_t *__restrict__ quad_totals = nullptr,
                   const uint32_t *__restrict__ hint_hdr = nullptr, const uint32_t *__restrict__ cin = nullptr,
                   uint32_t *__restrict__ cout = nullptr) {
    constexpr bool SELF_SCAN = SCAN == kScanSelf;
    constexpr int kWaveItems = kItems / 4;
    constexpr int NB = SPLIT ? SPLIT : kBins;     // digit rows the tables of this workgroup hold
    constexpr int DPT = NB / kThreads;            // digits per thread in the per-digit steps: thread t owns t * DPT ..
    static_assert(DPT == 1 || SCAN == kScanRows, "the passes without a row-scan launch have 256 digit rows");
    typedef typename std::conditional<(NB > 256), uint16_t, uint8_t>::type Dig;
    __shared__ uint32_t spl[SPLIT ? SPLIT : 1];
    __shared__ Dig sdig[SPLIT ? kItems : 1];     // SPLIT: the bucket of every parked item (not derivable by a shift)
    if (SPLIT)
        for (int k = threadIdx.x; k < NB; k += kThreads) spl[k] = splitter_at(splitters, hint_hdr, (uint32_t)k);   // visible after the barrier below
    __shared__ uint32_t cnt[4][NB];      // per-wave running digit counts, then per-wave LDS bases
    __shared__ uint32_t gbase[NB];       // global address of parked item j of digit d = gbase[d] + j
    __shared__ uint32_t wsum[4], lsum[4];
    __shared__ Key skey[kItems];
    __shared__ __attribute__((aligned(16))) uint32_t sval[kItems];   // 8 KB: also the 4 x 256 match words of the ranking
    __shared__ uint32_t scarry[CARRY ? kItems : 1];
    static_assert(!CARRY || (SCAN == kScanRows && !SPLIT && sizeof(Key) == 4), "the carried rectangle belongs to the LSD depth sort");
    static_assert(kItems * 4 == 4 * kBins * 8, "sval doubles as the per-wave match words");
    // The grid covers the capacity `bound`; the element count comes first: a workgroup beyond it must not touch
    // memory (requesting the keys before the count is known saved nothing measurable and cost 100 us on a frame
    // whose capacity was 20x its pair count).
    const uint32_t n = load_count(n_dev, bound);
    // Which chunk is this workgroup's?  With a row-scan launch behind it (large inputs): XCD x -- the workgroups with
    // blockIdx % 8 == x -- takes the x-th eighth of the chunks that hold items, so that the workgroups resident on one
    // XCD at a time work on CONSECUTIVE chunks.  A chunk leaves a run of ~8 items per digit (16 B of keys, 32 B of
    // values), and the runs of consecutive chunks are neighbours in memory: written from one XCD they meet in its L2
    // and leave it as whole 64-byte sectors; dealt round robin over the eight L2s (chunk = blockIdx) every run reached
    // HBM as a partial sector of its own -- 2.0x the algorithmic bytes written, and as much again read for the
    // read-modify-write, in the 4K frame's tile sort (PMC, round 3).
    uint32_t chunk_id = blockIdx.x;
    if (kXcdChunks) {       // (every mode: the small passes without a row-scan launch write the same short runs)
        const uint32_t chunks = (n + (uint32_t)kItems - 1u) / (uint32_t)kItems, per = (chunks + 7u) >> 3;
        const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        if (i >= per) return;
        chunk_id = x * per + i;
    }
    const uint32_t block_base = chunk_id * (uint32_t)kItems;
    if (block_base >= n) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr uint32_t nbins = SPLIT ? (uint32_t)SPLIT : (1u << BITS), mask = (1u << BITS) - 1u;
    const uint32_t wave_base = block_base + (uint32_t)w * kWaveItems;
    Key key[kRounds];
    uint32_t val[kRounds];
    uint32_t carry[CARRY ? kRounds : 1];
    bool ok[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const uint32_t i = wave_base + (uint32_t)r * 64 + lane;
        ok[r] = i < n;
        key[r] = ok[r] ? kin[i] : (Key)0;
        // FIRST: the value of an item is its position (the Gaussian index): nothing to read -- unless the positions are
        // original indices of reordered rows (GsxParams.row_of_index, handed in as `vin`): then it is the row
        val[r] = (MODE & kModeFirst) ? (vin ? (ok[r] ? vin[i] : 0u) : i) : (ok[r] ? vin[i] : 0u);
        if (CARRY) {
            carry[r] = 0u;
            if (ok[r]) {
                if (MODE & kModeFirst) {
                    const TileRect t = rect[val[r]];  // (index order: coalesced; a gather under row_of_index)
                    carry[r] = (uint32_t)t.x0 | ((uint32_t)t.x1 << 8) | ((uint32_t)t.y0 << 16) | ((uint32_t)t.y1 << 24);
                } else {
                    carry[r] = cin[i];
                }
            }
        }
        if (MODE & kModeFirst) ok[r] = ok[r] && (uint32_t)key[r] < kEmptyKey;
    }
    // this thread's digits' row totals and row prefixes: needed after the ranking
    uint32_t t_pre[DPT], before_pre[DPT];
#pragma unroll
    for (int q = 0; q < DPT; ++q) t_pre[q] = before_pre[q] = 0;
    if (SCAN == kScanRows) {
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const uint32_t d = (uint32_t)threadIdx.x * DPT + q;
            if (d < nbins) {
                t_pre[q] = totals[d];
                before_pre[q] = table[(size_t)d * nbp + chunk_id];
            }
        }
    }
    if (SCAN == kScanQuads && (uint32_t)threadIdx.x < nbins) {
        // no row scan ran: this digit's total = the sum of all count workgroups' four-chunk totals, its prefix =
        // the totals of the workgroups before this chunk's + the (raw) counts of the chunks before it in its quad
        const int nquads = nbp / kQuad, quad = (int)(chunk_id / kQuad);
        int q = 0;
        for (; q + 8 <= nquads; q += 8) {      // 8 independent loads in flight (coalesced over the digits)
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = quad_totals[(size_t)(q + u) * kBins + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                before_pre[0] += q + u < quad ? v[u] : 0u;
                t_pre[0] += v[u];
            }
        }
        for (; q < nquads; ++q) {
            const uint32_t v = quad_totals[(size_t)q * kBins + threadIdx.x];
            before_pre[0] += q < quad ? v : 0u;
            t_pre[0] += v;
        }
        for (uint32_t c = (uint32_t)quad * kQuad; c < chunk_id; ++c) before_pre[0] += table[(size_t)threadIdx.x * nbp + c];
        if (chunk_id == 0) const_cast<uint32_t *>(totals)[threadIdx.x] = t_pre[0];   // what bucket_sort_kernel reads
    }
    for (int k = threadIdx.x; k < 4 * NB; k += kThreads) (&cnt[0][0])[k] = 0;
    for (int k = threadIdx.x; k < kItems / 4; k += kThreads) reinterpret_cast<uint4 *>(sval)[k] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // ---- stable rank of every item among the same-digit items of its wave's consecutive slice
    const unsigned long long lt = (1ull << lane) - 1ull;
    // The lanes of a wave that hold the same digit find each other through the LDS: every lane ORs its bit
    // into the wave's 64-bit word of the digit, reads the word back and clears it -- three DS operations that
    // one wave executes in issue order -- instead of BITS ballots at ~6 VALU instructions each.  The words
    // live in sval, which is dead until the items are parked.  With 1024 buckets the words are those of the
    // digit's low 8 bits and two ballots tell apart the lanes that differ in the upper two (256 words per wave
    // is what sval holds; 1024 would cost the kernel two resident workgroups per CU).
    // (Round 3 re-measured the ballot form on the throughput-bound scatters of large frames -- six workgroups share a
    // CU's LDS pipe there --: 16-bit tile keys, 7-bit digits at 1M Gaussians 16.0 -> 20.0 us, 8-bit at 5M 86.1 -> 85.5
    // us, 100k 8.3 -> 9.5 us.  The LDS words stay.)
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u64 *wm = (lds_u64 *)(reinterpret_cast<unsigned long long *>(sval) + (size_t)w * kBins);
    lds_u32 *wc = (lds_u32 *)&cnt[w][0];
    const unsigned long long me = 1ull << lane;
    uint16_t rank[kRounds];
    Dig dig[kRounds];
    {
        uint32_t kq[kRounds], dq[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            kq[r] = (uint32_t)key[r];
            dq[r] = ((uint32_t)key[r] >> shift) & mask;
        }
        if (SPLIT) buckets_of<NB, kRounds>(spl, kq, dq);
#pragma unroll
        for (int r = 0; r < kRounds; ++r) dig[r] = (Dig)dq[r];
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const uint32_t d = dig[r];
        rank[r] = 0;
        unsigned long long hi0 = 0ull, hi1 = 0ull;
        if (NB > 256) {       // lanes whose bucket has bit 8 / bit 9 set
            hi0 = __ballot(ok[r] && (d & 256u));
            hi1 = __ballot(ok[r] && (d & 512u));
        }
        if (ok[r]) {
            __hip_atomic_fetch_or(&wm[d & 255u], me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            unsigned long long peers = wm[d & 255u];
            const uint32_t before = wc[d];
            __builtin_amdgcn_wave_barrier();
            wm[d & 255u] = 0ull;
            if (NB > 256) peers &= ((d & 256u) ? hi0 : ~hi0) & ((d & 512u) ? hi1 : ~hi1);
            if ((peers & lt) == 0ull) wc[d] = before + (uint32_t)__popcll(peers);   // the first of them
            __builtin_amdgcn_wave_barrier();
            rank[r] = (uint16_t)(before + (uint32_t)__popcll(peers & lt));
        }
    }
    __syncthreads();

    // ---- per digit d (thread d / DPT): where its run starts in the parked (digit-major) order and in
    //      the global output: smaller digits (row totals) + this digit in earlier chunks (table)
    {
        uint32_t cw[DPT][4], l[DPT], t[DPT], before[DPT];
        uint32_t tsum = 0, lsum_ = 0;
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int d = threadIdx.x * DPT + q;
#pragma unroll
            for (int k = 0; k < 4; ++k) cw[q][k] = cnt[k][d];
            if (SELF_SCAN) {
                t[q] = 0;
                before[q] = 0;
                int b = 0;
                for (; b + 8 <= nbp; b += 8) {      // 8 independent loads in flight (chunk-major: coalesced over d)
                    uint32_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = table[(size_t)(b + u) * kBins + d];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        before[q] += b + u < (int)chunk_id ? v[u] : 0u;
                        t[q] += v[u];
                    }
                }
                for (; b < nbp; ++b) {
                    const uint32_t v = table[(size_t)b * kBins + d];
                    before[q] += b < (int)chunk_id ? v : 0u;
                    t[q] += v;
                }
            } else {
                t[q] = t_pre[q];
                before[q] = before_pre[q];
            }
            l[q] = cw[q][0] + cw[q][1] + cw[q][2] + cw[q][3];
            tsum += t[q];
            lsum_ += l[q];
        }
        uint32_t x = tsum, y = lsum_;  // inclusive wave scans of the global totals and of the local counts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t xu = __shfl_up((int)x, o), yu = __shfl_up((int)y, o);
            if (lane >= o) {
                x += xu;
                y += yu;
            }
        }
        if (lane == 63) {
            wsum[w] = x;
            lsum[w] = y;
        }
        __syncthreads();
        uint32_t gb = 0, lb = 0;
        for (int k = 0; k < w; ++k) {
            gb += wsum[k];
            lb += lsum[k];
        }
        uint32_t lstart = lb + y - lsum_;                       // first parked slot of this thread's first digit
        uint32_t gstart = gb + x - tsum;                        // keys of smaller digits, all chunks
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int d = threadIdx.x * DPT + q;
            gbase[d] = gstart + before[q] - lstart;
            cnt[0][d] = lstart;
            cnt[1][d] = lstart + cw[q][0];
            cnt[2][d] = lstart + cw[q][0] + cw[q][1];
            cnt[3][d] = lstart + cw[q][0] + cw[q][1] + cw[q][2];
            lstart += l[q];
            gstart += t[q];
        }
        if ((MODE & kModeFirst) && chunk_id == 0 && threadIdx.x == 0) *m_out = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    __syncthreads();
    const uint32_t live = lsum[0] + lsum[1] + lsum[2] + lsum[3];   // items of this chunk that the pass keeps

    // ---- park in LDS, digit-major, stable
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        if (ok[r]) {
            const uint32_t d = dig[r];
            const uint32_t pos = cnt[w][d] + rank[r];
            skey[pos] = key[r];
            sval[pos] = val[r];
            if (CARRY) scarry[pos] = carry[r];
            if (SPLIT) sdig[pos] = dig[r];
        }
    }
    __syncthreads();

    // ---- stream out: consecutive j of one digit -> consecutive addresses
    constexpr bool with_rect = ((MODE & kModeFinal) && !CARRY) || SPLIT;
    if (with_rect) {
        // FINAL: the one gather by Gaussian index of the LSD depth sort.  SPLIT (the partition pass of the sampled
        // sort): v lies in this chunk's own 2048 indices -- 16 KB of rect, read once -- and the rectangles travel
        // with their items, so that the bucket sort finds them side by side.  All gathers are issued before
        // the first store.
        uint32_t dst[kRounds], v[kRounds];
        Key k[kRounds];
        TileRect t[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const uint32_t j = (uint32_t)r * kThreads + threadIdx.x;
            if (j < live) {
                k[r] = skey[j];
                v[r] = sval[j];
                dst[r] = gbase[SPLIT ? (uint32_t)sdig[j] : (((uint32_t)k[r] >> shift) & mask)] + j;
                t[r] = rect[v[r]];
            }
        }
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const uint32_t j = (uint32_t)r * kThreads + threadIdx.x;
            if (j < live) {
                if (!(MODE & kModeFinal)) kout[dst[r]] = k[r];
                vout[dst[r]] = v[r];
                rrect[dst[r]] = t[r];
            }
        }
    } else {
        for (uint32_t j = threadIdx.x; j < live; j += kThreads) {
            const Key k = skey[j];
            const uint32_t dst = gbase[((uint32_t)k >> shift) & mask] + j;
            if (!(MODE & kModeFinal)) kout[dst] = k;
            vout[dst] = sval[j];
            if (CARRY) {
                const uint32_t c = scarry[j];
                if (MODE & kModeFinal)
                    rrect[dst] = TileRect{(uint16_t)(c & 255u), (uint16_t)((c >> 8) & 255u), (uint16_t)((c >> 16) & 255u),
                                          (uint16_t)(c >> 24)};
                else
                    cout[dst] = c;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// Sample-partitioned depth sort (sort_depth_sampled): 6 kernels instead of the 12 of four LSD passes.
//   sample    2048 regularly spaced keys are ranked among themselves on all CUs (sample_rank_kernel); the
//             samples that sit on the 255 regular quantiles are the splitters;
//   partition ONE stable pass of the count / row_scan / scatter machinery above with the bucket among the
//             splitters as the "digit" (FIRST mode: drops what reaches no tile, values = positions);
//   buckets   one 1024-thread workgroup per bucket sorts its items in LDS -- only over the key bytes that
//             vary inside the bucket -- and streams them out with the rank-ordered rectangle gather.
// A bucket larger than the LDS capacity (possible, never seen: the samples keep the sizes within
// a few 10 % of the mean -- with 8 samples per bucket a bucket 4x the mean, which is what the LDS holds at 1M
// keys, has probability ~1e-9; adversarial key layouts are tested) is sorted by the same workgroup through
// global memory, tile by tile.  Equal keys share a bucket and every step is stable: ties keep index order.
constexpr int kBigThreads = 1024, kBigWaves = kBigThreads / 64;
constexpr int kBucketRounds = 16, kBucketCap = kBigThreads * kBucketRounds;   // 16 384 items in LDS

constexpr int kLocalBits = 9, kLocalBins = 1 << kLocalBits;   // digits of the in-LDS passes: 17 varying bits = 2 passes
struct RankShared {
    uint16_t cnt[kBigWaves][kLocalBins];   // per-wave running digit counts, then per-wave positions (< 2^16)
    uint32_t lstart[kLocalBins + 1];       // first position of every digit in the digit-major order
    uint32_t wsum[kLocalBins / 64];
};

// Stable digit-major positions of up to ROUNDS x 1024 items held in registers: item (wave w, round r, lane)
// is the (w * L + r * 64 + lane)-th of the tile (L = the per-wave slice, a multiple of 64).  Digit = DBITS bits
// of the key from `shift`.  All 1024 threads must call this (barriers inside).  On return pos[r] is the
// item's position, sh.lstart[d] the first position of digit d, sh.lstart[1 << DBITS] the number of valid items.
//
// The lanes of a wave that hold the same digit find each other through the LDS, not through DBITS ballots
// (6 VALU instructions per bit and round; a wave64 instruction occupies the SIMD for 4 cycles): every lane ORs
// its bit into the wave's 64-bit word of the digit, reads the word back, and clears it for the next round --
// three LDS operations.  The DS instructions of one wave execute in issue order, so the read sees the ORs of
// all 64 lanes and the clear comes after every read.  `masks`: kBigWaves x kLocalBins 64-bit words (64 KB),
// scratch that callers alias with the item buffers (dead while positions are being computed).
template <int ROUNDS, int DBITS>
__device__ __forceinline__ void rank_items(const uint32_t (&key)[ROUNDS], const bool (&ok)[ROUNDS], int shift,
                                           RankShared &sh, unsigned long long *masks, uint32_t (&pos)[ROUNDS]) {
    constexpr int NB = 1 << DBITS;
    constexpr uint32_t DM = NB - 1;
    static_assert(NB <= kLocalBins, "the tables hold kLocalBins digits");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long me = 1ull << lane, lt = me - 1ull;
    {
        uint4 *z = reinterpret_cast<uint4 *>(masks);
        for (int k = threadIdx.x; k < kBigWaves * kLocalBins * 8 / 16; k += kBigThreads) z[k] = make_uint4(0, 0, 0, 0);
        uint4 *c = reinterpret_cast<uint4 *>(&sh.cnt[0][0]);
        for (int k = threadIdx.x; k < kBigWaves * kLocalBins * 2 / 16; k += kBigThreads) c[k] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // explicit LDS pointers: the DS instructions below must stay DS instructions, in this order
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    lds_u64 *wm = (lds_u64 *)(masks + (size_t)w * kLocalBins);
    lds_u16 *wc = (lds_u16 *)&sh.cnt[w][0];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool valid = ok[r];
        const uint32_t d = (key[r] >> shift) & DM;
        pos[r] = 0;
        if (__ballot(valid) == 0ull) continue;   // wave-uniform: nothing left in this wave's slice
        if (valid) {
            __hip_atomic_fetch_or(&wm[d], me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            const unsigned long long peers = wm[d];
            const uint32_t before = wc[d];
            __builtin_amdgcn_wave_barrier();
            wm[d] = 0ull;
            if ((peers & lt) == 0ull) wc[d] = (uint16_t)(before + (uint32_t)__popcll(peers));   // the first of them
            __builtin_amdgcn_wave_barrier();
            pos[r] = before + (uint32_t)__popcll(peers & lt);
        }
    }
    __syncthreads();
    uint32_t total = 0, x = 0;
    if (threadIdx.x < NB) {   // thread d: exclusive prefix of digit d over the waves, then over the digits
        const int d = threadIdx.x;
#pragma unroll
        for (int k = 0; k < kBigWaves; ++k) {
            const uint32_t c = sh.cnt[k][d];
            sh.cnt[k][d] = (uint16_t)total;
            total += c;
        }
        x = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) sh.wsum[w] = x;
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        uint32_t before = 0;
        for (int k = 0; k < w; ++k) before += sh.wsum[k];
        const uint32_t start = before + x - total;
        sh.lstart[threadIdx.x] = start;
        if (threadIdx.x == NB - 1) sh.lstart[NB] = start + total;
#pragma unroll
        for (int k = 0; k < kBigWaves; ++k) sh.cnt[k][threadIdx.x] = (uint16_t)(sh.cnt[k][threadIdx.x] + start);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        if (ok[r]) pos[r] += sh.cnt[w][(key[r] >> shift) & DM];
}

// The in-LDS LSD passes over the bits set in [lo_bit, hi_bit) for up to R x 1024 items held in registers
// (item (wave w, round r, lane) = the (w * L + r * 64 + lane)-th): 9 bits per pass, items parked in digit-major
// order in (skey, sval) and read back position-major.  COMPACT: invalid items are dropped by the first pass and
// the count of valid ones is returned (the kept items are dense afterwards).  R is a template parameter
// because every round costs instructions at 6 code sites whether the wave has items in it or not: a bucket
// of 3 000 items (4 rounds) through the 16-round code took 11 us per pass, through the 4-round code 5.
template <int R, bool COMPACT>
__device__ __forceinline__ uint32_t lds_passes(uint32_t (&key)[R], uint32_t (&val)[R], bool (&ok)[R], uint32_t L,
                                               int lo_bit, int hi_bit, RankShared &sh, uint32_t *skey, uint32_t *sval) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t m = 0;
    for (int shift = lo_bit; shift < hi_bit; shift += kLocalBits) {
        uint32_t pos[R];
        rank_items<R, kLocalBits>(key, ok, shift, sh, reinterpret_cast<unsigned long long *>(skey), pos);
        m = sh.lstart[kLocalBins];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (ok[r]) {
                skey[pos[r]] = key[r];
                sval[pos[r]] = val[r];
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = (uint32_t)w * L + (uint32_t)r * 64 + lane;
            if (COMPACT) ok[r] = (uint32_t)r * 64 < L && i < m;
            if (ok[r]) {
                key[r] = skey[i];
                val[r] = sval[i];
            }
        }
        __syncthreads();
    }
    return m;
}

// chunk_sums[c] += tile counts of the ranks of chunk c (kEmitChunk consecutive depth ranks: what the pair
// emission needs of the rank-ordered rectangles before it can start; fused into the sort's last kernel it saves
// the frame a launch).  A wave's ranks are the L <= 1024 consecutive ones from `wave_first`, i.e. at most two
// chunks: every lane adds up its own rounds (ChunkTally::add), then ONE pair of wave reductions and at most two
// atomics per wave (ChunkTally::flush; a reduction per round cost 4 us in dependent cross-lane steps).
struct ChunkTally {
    uint32_t chunk0, a, b;      // 64 x 16 x 65 535 tiles fit 32 bits
    __device__ __forceinline__ explicit ChunkTally(uint32_t wave_first) : chunk0(wave_first / (uint32_t)kEmitChunk), a(0), b(0) {}
    __device__ __forceinline__ void add(uint32_t rank, bool have, uint2 rc) {
        const uint32_t x0 = rc.x & 0xFFFFu, x1 = rc.x >> 16, y0 = rc.y & 0xFFFFu, y1 = rc.y >> 16;
        const uint32_t tiles = (have && x0 <= x1) ? (x1 - x0 + 1u) * (y1 - y0 + 1u) : 0u;
        const bool in0 = rank / (uint32_t)kEmitChunk == chunk0;
        a += in0 ? tiles : 0u;
        b += in0 ? 0u : tiles;
    }
    __device__ __forceinline__ void flush(unsigned long long *__restrict__ chunk_sums) {   // all 64 lanes
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += (uint32_t)__shfl_xor((int)a, o);
            b += (uint32_t)__shfl_xor((int)b, o);
        }
        if ((threadIdx.x & 63) == 0) {
            if (a) atomicAdd(&chunk_sums[chunk0], (unsigned long long)a);
            if (b) atomicAdd(&chunk_sums[chunk0 + 1u], (unsigned long long)b);
        }
    }
};

// vout[first + i] = val, rrect[first + i] = rect[val] for the items in registers.  All the rectangle loads
// are issued before the first store: with load and store of one item back to back the compiler waits for
// every gather in turn (15 us of a 43 us kernel at 1M keys).
template <int R>
__device__ __forceinline__ void store_ranked(const uint32_t (&val)[R], const bool (&ok)[R], uint32_t L, uint32_t first,
                                             uint32_t *__restrict__ vout, const TileRect *__restrict__ rect,
                                             TileRect *__restrict__ rrect, unsigned long long *__restrict__ chunk_sums) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    TileRect rc[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (ok[r]) rc[r] = rect[val[r]];
    ChunkTally tally(first + (uint32_t)w * L);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)w * L + (uint32_t)r * 64 + lane;
        if (ok[r]) {
            vout[first + i] = val[r];
            rrect[first + i] = rc[r];
            tally.add(first + i, true, make_uint2((uint32_t)rc[r].x0 | ((uint32_t)rc[r].x1 << 16),
                                                  (uint32_t)rc[r].y0 | ((uint32_t)rc[r].y1 << 16)));
        }
    }
    if (chunk_sums) tally.flush(chunk_sums);
}

// Minimum and maximum of the valid keys over the workgroup (s_min / s_max preset to ~0 / 0 before a barrier).
template <int R>
__device__ __forceinline__ void key_span(const uint32_t (&key)[R], const bool (&ok)[R], uint32_t *s_min, uint32_t *s_max) {
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mn = ok[r] ? min(mn, key[r]) : mn;
        mx = ok[r] ? max(mx, key[r]) : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    }
    if ((threadIdx.x & 63) == 0 && mn <= mx) {
        atomicMin(s_min, mn);
        atomicMax(s_max, mx);
    }
    __syncthreads();
}

// SortHints.next_splitters: the key of depth rank g is the next frame's splitter j for every j with floor(j M / 256) == g
// (none, one, or several when fewer than 256 keys are kept) -- M = number of kept keys.  Every j in 1 .. 255 names exactly one
// rank, every rank lies in exactly one non-empty bucket, and both bucket paths call this for their ranks [start, start + size):
// all 255 are written, from one sorted sequence -- a monotone set by construction.  ONE thread of the workgroup calls it:
// a bucket holds one quantile or two (key_at(i) = the key of the bucket's i-th rank).
template <typename KeyAt>
__device__ __forceinline__ void leave_splitters(uint32_t start, uint32_t size, uint32_t M, KeyAt key_at, uint32_t *__restrict__ spl) {
    uint32_t j = (uint32_t)(((unsigned long long)start * (uint32_t)kBins + M - 1u) / M);
    for (; j < (uint32_t)kBins; ++j) {
        const uint32_t g = (uint32_t)(((unsigned long long)j * M) / (uint32_t)kBins);
        if (g >= start + size) break;
        if (j) spl[j] = key_at(g - start);
    }
}

// A bucket that fits the LDS: load, sort over the bits of (key - smallest key), stream out.  (All 1024 threads.)
// The offset from the smallest key, not the bits in which keys differ: a bucket holds 1/256 of the keys, a span
// of 2^15 .. 2^18 float-depth codes at 1M Gaussians = two 9-bit passes, where one bucket that straddles a
// power of two differs in 20+ bits (and the slowest bucket is the kernel's duration).
//
// What is sorted is (key, position in the bucket).  The partition pass left the bucket's Gaussian indices in
// vin[start ..) and their rectangles in rrect[start ..) (partition order); both are staged in LDS with
// coalesced loads, permuted there, and written back -- the rectangles in place.  (A gather rect[index] from
// here, 1M random 8-byte reads of an 8 MB table that no XCD's L2 holds, was 11 of the kernel's 38 us.)
// `items`: the 2 x kBucketCap words of LDS item buffer.
template <int R>
__device__ __forceinline__ void bucket_in_lds(uint32_t start, uint32_t size, uint32_t L, const uint32_t *kin,
                                              const uint32_t *vin, uint32_t *vout, TileRect *rrect, RankShared &sh,
                                              uint32_t *items, uint32_t *s_min, uint32_t *s_max,
                                              unsigned long long *chunk_sums, uint32_t *next_spl, uint32_t kept) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t key[R], val[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)w * L + (uint32_t)r * 64 + lane;
        ok[r] = (uint32_t)r * 64 < L && i < size;
        key[r] = ok[r] ? kin[start + i] : 0u;
        val[r] = i;
    }
    // what is staged after the sort is requested now (up to 8 rounds: 24 registers), so that its trip to memory
    // runs under the passes
    constexpr bool kEarly = R <= 8;
    uint2 *lrect = reinterpret_cast<uint2 *>(items);
    const uint2 *grect = reinterpret_cast<const uint2 *>(rrect + start);
    uint2 t[R];
    uint32_t v[R];
    if (kEarly) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = (uint32_t)r * kBigThreads + threadIdx.x;
            if (j < size) {
                t[r] = grect[j];
                v[r] = vin[start + j];
            }
        }
    }
    key_span<R>(key, ok, s_min, s_max);
    const uint32_t kmin = *s_min, span = *s_max - kmin;
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] -= kmin;
    lds_passes<R, false>(key, val, ok, L, 0, span ? 32 - __clz((int)span) : 0, sh, items, items + kBucketCap);
    // (the last pass left the sorted keys, minus kmin, in `items` -- unless no bit varies: then every key is kmin)
    if (next_spl && threadIdx.x == 0)
        leave_splitters(start, size, kept, [&](uint32_t i) -> uint32_t { return span ? items[i] + kmin : kmin; }, next_spl);
    if (next_spl) __syncthreads();        // (uniform; `items` is reused below)
    // val[r] = position in the bucket of the item of rank (w, r, lane).  Thread-strided staging: item j = r * 1024 + tid.
    const bool together = size * 3u <= 2u * (uint32_t)kBucketCap;   // 12 bytes per item fit the buffer at once
    uint32_t *lval = together ? items + 2 * size : items;
    if (!kEarly) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = (uint32_t)r * kBigThreads + threadIdx.x;
            if (j < size) {
                t[r] = grect[j];
                v[r] = vin[start + j];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = (uint32_t)r * kBigThreads + threadIdx.x;
        if (j < size) {
            lrect[j] = t[r];
            if (together) lval[j] = v[r];
        }
    }
    __syncthreads();
    ChunkTally tally(start + (uint32_t)w * L);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)w * L + (uint32_t)r * 64 + lane;
        if (ok[r]) {
            const uint2 rc = lrect[val[r]];
            reinterpret_cast<uint2 *>(rrect + start)[i] = rc;
            if (together) vout[start + i] = lval[val[r]];
            tally.add(start + i, true, rc);
        }
    }
    if (chunk_sums) tally.flush(chunk_sums);
    if (R == kBucketRounds && !together) {   // a bucket beyond 2/3 of the buffer: the indices take a second trip
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t j = (uint32_t)r * kBigThreads + threadIdx.x;
            if (j < size) lval[j] = v[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = (uint32_t)w * L + (uint32_t)r * 64 + lane;
            if (ok[r]) vout[start + i] = lval[val[r]];
        }
    }
}

// (the ranking itself: gsx_sample_device.h -- shared with the projection launch, whose spare workgroups rank a sample of
// keys they compute themselves on a frame that has no splitters yet)
template <int NB>
__global__ void __launch_bounds__(kThreads)
    sample_rank_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t ns, uint32_t *__restrict__ splitters,
                       unsigned long long *__restrict__ chunk_sums, uint32_t nsums) {
    __shared__ uint32_t sm[kSamplesMax];
    __shared__ uint32_t s_wave[kThreads / 64];
    sample_rank_body<NB>([&](uint32_t i) { return keys[i]; }, n, ns, splitters, chunk_sums, nsums, blockIdx.x, gridDim.x, sm, s_wave);
}

// One workgroup per bucket of the partition pass.  in: (kin, vin) partitioned by bucket, bucket sizes = the
// row totals of that pass.  out: vout[rank] = Gaussian index, rrect[rank] = rect[index] for the bucket's
// ranks.  kalt: the other key buffer (scratch of the through-memory path).
template <int NB>
__global__ void __launch_bounds__(kBigThreads)
    bucket_sort_kernel(const uint32_t *__restrict__ totals, const uint32_t *__restrict__ table_cm, int nblocks_cm,
                       uint32_t *kin, uint32_t *vin, uint32_t *kalt, uint32_t *vout,
                       const TileRect *__restrict__ rect, TileRect *rrect, uint32_t lds_cap,
                       unsigned long long *__restrict__ chunk_sums, uint32_t *next_spl, uint32_t *next_hdr,
                       const uint32_t *__restrict__ m_dev) {
    __shared__ RankShared sh;
    __shared__ __attribute__((aligned(16))) uint32_t sitems[2 * kBucketCap];
    uint32_t *skey = sitems;
    __shared__ uint32_t s_tot[NB];
    __shared__ uint32_t s_or, s_start, s_min, s_max;
    static_assert(NB <= kBigThreads && NB >= kBins, "one thread per bucket size below");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < NB) {
        // bucket sizes: the row totals of the partition pass, or (small inputs: chunk-major table, no row scan)
        // the column sums of the raw counts
        uint32_t t = 0;
        if (NB == kBins && nblocks_cm > 0) {
            int b = 0;
            for (; b + 8 <= nblocks_cm; b += 8) {     // 8 independent loads in flight
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = table_cm[(size_t)(b + u) * kBins + threadIdx.x];
#pragma unroll
                for (int u = 0; u < 8; ++u) t += v[u];
            }
            for (; b < nblocks_cm; ++b) t += table_cm[(size_t)b * kBins + threadIdx.x];
        } else {
            t = totals[threadIdx.x];
        }
        s_tot[threadIdx.x] = t;
    }
    if (threadIdx.x == 0) {
        s_or = 0;
        s_start = 0;
        s_min = 0xFFFFFFFFu;
        s_max = 0;
    }
    __syncthreads();
    if (threadIdx.x < NB) {   // first rank of this bucket = sizes of the buckets before it
        uint32_t c = (uint32_t)threadIdx.x < blockIdx.x ? s_tot[threadIdx.x] : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
        if (lane == 0 && c) atomicAdd(&s_start, c);
    }
    __syncthreads();
    const uint32_t start = s_start, size = s_tot[blockIdx.x];
    const uint32_t kept = next_spl ? *m_dev : 0u;      // (written by the partition pass: every kept key, all buckets)
    if (next_spl && blockIdx.x == 0 && threadIdx.x == 0) {
        next_spl[0] = 0u;
        next_hdr[kHintSplitters] = kept ? (uint32_t)kBins : 0u;     // nothing kept: the next frame takes its stand-in splitters
    }
    if (size == 0 || (next_spl && kept == 0u)) return;
    if (size <= lds_cap) {
        // ---- in LDS: items wave-striped, L per wave, as few rounds of 64 per wave as hold them
        const uint32_t L = (((size + kBigWaves - 1) / kBigWaves) + 63u) & ~63u;
        if (L <= 2 * 64)
            bucket_in_lds<2>(start, size, L, kin, vin, vout, rrect, sh, sitems, &s_min, &s_max, chunk_sums, next_spl, kept);
        else if (L <= 4 * 64)
            bucket_in_lds<4>(start, size, L, kin, vin, vout, rrect, sh, sitems, &s_min, &s_max, chunk_sums, next_spl, kept);
        else if (L <= 8 * 64)
            bucket_in_lds<8>(start, size, L, kin, vin, vout, rrect, sh, sitems, &s_min, &s_max, chunk_sums, next_spl, kept);
        else
            bucket_in_lds<kBucketRounds>(start, size, L, kin, vin, vout, rrect, sh, sitems, &s_min, &s_max, chunk_sums, next_spl, kept);
        return;
    }
    // ---- through global memory (a bucket that does not fit): LSD passes over the varying bytes, tile by tile
    const uint32_t key0 = kin[start];
    uint32_t diff = 0;
    for (uint32_t i = threadIdx.x; i < size; i += kBigThreads) diff |= kin[start + i] ^ key0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff |= (uint32_t)__shfl_xor((int)diff, o);
    if (lane == 0 && diff) atomicOr(&s_or, diff);
    __syncthreads();
    const uint32_t varying = s_or;
    uint32_t *sk = kin + start, *sv = vin + start, *dk = kalt + start, *dv = vout + start;
    uint32_t *gbase = s_tot;     // 256 running digit bases (skey is rank_items' scratch)
    for (int shift = 0; shift < 32; shift += 8) {
        if (((varying >> shift) & 255u) == 0) continue;
        if (threadIdx.x < kBins) s_tot[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < size; i += kBigThreads) atomicAdd(&s_tot[(sk[i] >> shift) & 255u], 1u);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int d = 0; d < kBins; ++d) {   // in place: counts -> first destinations
                const uint32_t c = s_tot[d];
                gbase[d] = run;
                run += c;
            }
        }
        __syncthreads();
        for (uint32_t t0 = 0; t0 < size; t0 += lds_cap) {
            const uint32_t tsize = min(lds_cap, size - t0);
            const uint32_t L = (((tsize + kBigWaves - 1) / kBigWaves) + 63u) & ~63u;
            uint32_t key[kBucketRounds], val[kBucketRounds], pos[kBucketRounds];
            bool ok[kBucketRounds];
#pragma unroll
            for (int r = 0; r < kBucketRounds; ++r) {
                const uint32_t i = (uint32_t)w * L + (uint32_t)r * 64 + lane;
                ok[r] = (uint32_t)r * 64 < L && i < tsize;
                key[r] = ok[r] ? sk[t0 + i] : 0u;
                val[r] = ok[r] ? sv[t0 + i] : 0u;
            }
            rank_items<kBucketRounds, 8>(key, ok, shift, sh, reinterpret_cast<unsigned long long *>(skey), pos);
#pragma unroll
            for (int r = 0; r < kBucketRounds; ++r)
                if (ok[r]) {
                    const uint32_t d = (key[r] >> shift) & 255u;
                    const uint32_t dst = gbase[d] + (pos[r] - sh.lstart[d]);
                    dk[dst] = key[r];
                    dv[dst] = val[r];
                }
            __syncthreads();
            if (threadIdx.x < kBins) {
                const uint32_t d = threadIdx.x;
                const uint32_t next = d + 1 < kBins ? sh.lstart[d + 1] : sh.lstart[kBins];
                gbase[d] += next - sh.lstart[d];
            }
            __syncthreads();
        }
        __threadfence_block();   // the next pass of this workgroup reads what this one wrote
        __syncthreads();
        uint32_t *tk = sk; sk = dk; dk = tk;
        uint32_t *tv = sv; sv = dv; dv = tv;
    }
    // the sorted values now sit in sv: they belong in vout, with their rectangles beside them
    for (uint32_t i0 = 0; i0 < size; i0 += kBigThreads) {     // (each trip: 64 consecutive ranks per wave)
        const uint32_t i = i0 + threadIdx.x;
        ChunkTally tally(start + i0 + (uint32_t)w * 64u);
        if (i < size) {
            const uint32_t v = sv[i];
            if (sv != vout + start) vout[start + i] = v;
            const uint2 rc = reinterpret_cast<const uint2 *>(rect)[v];
            reinterpret_cast<uint2 *>(rrect)[start + i] = rc;
            tally.add(start + i, true, rc);
        }
        if (chunk_sums) tally.flush(chunk_sums);
    }
    if (next_spl && threadIdx.x == 0)       // (sk: the sorted keys of the last pass, or the bucket as it came when no byte varies)
        leave_splitters(start, size, kept, [&](uint32_t i) -> uint32_t { return sk[i]; }, next_spl);
}

// Up to kBucketCap keys (the reference's own scenes: 2 000 .. 52 000 Gaussians fit or nearly fit) ONE workgroup
// does the whole depth sort in LDS: drop what reaches no tile, count the culled, LSD passes over the bits of
// (key - smallest key), rank-ordered rectangle gather.  One launch instead of eight.
template <int R>
__device__ __forceinline__ void small_sort_in_lds(const uint32_t *__restrict__ keys, uint32_t n, uint32_t L,
                                                  uint32_t *__restrict__ vout, const TileRect *__restrict__ rect,
                                                  TileRect *__restrict__ rrect, uint32_t *__restrict__ m_out,
                                                  uint32_t *__restrict__ culled_out, RankShared &sh, uint32_t *skey,
                                                  uint32_t *sval, uint32_t *s_min, uint32_t *s_max, uint32_t *s_culled,
                                                  unsigned long long *__restrict__ chunk_sums, const uint32_t *__restrict__ row_of) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t key[R], val[R];
    bool ok[R];
    uint32_t culled = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)w * L + (uint32_t)r * 64 + lane;
        const bool in = (uint32_t)r * 64 < L && i < n;
        key[r] = in ? keys[i] : 0u;
        val[r] = (row_of && in) ? row_of[i] : i;        // (GsxParams.row_of_index: the row the original index i names)
        ok[r] = in && key[r] < kEmptyKey;
        culled += in && key[r] == kCulledKey;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) culled += (uint32_t)__shfl_xor((int)culled, o);
    if (lane == 0 && culled) atomicAdd(s_culled, culled);
    key_span<R>(key, ok, s_min, s_max);
    const uint32_t kmin = *s_min, span = *s_max - kmin;
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] -= kmin;
    // one pass at least: the first pass is also what compacts
    const uint32_t m = lds_passes<R, true>(key, val, ok, L, 0, span ? 32 - __clz((int)span) : 1, sh, skey, sval);
    store_ranked<R>(val, ok, L, 0u, vout, rect, rrect, chunk_sums);
    if (threadIdx.x == 0) {
        *m_out = m;
        *culled_out = *s_culled;
    }
}

__global__ void __launch_bounds__(kBigThreads)
    small_depth_sort_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ vout,
                            const TileRect *__restrict__ rect, TileRect *__restrict__ rrect, uint32_t *__restrict__ m_out,
                            uint32_t *__restrict__ culled_out, unsigned long long *__restrict__ chunk_sums,
                            const uint32_t *__restrict__ row_of) {
    __shared__ RankShared sh;
    __shared__ uint32_t skey[kBucketCap];
    __shared__ uint32_t sval[kBucketCap];
    __shared__ uint32_t s_min, s_max, s_culled;
    if (threadIdx.x == 0) {
        s_min = 0xFFFFFFFFu;
        s_max = 0;
        s_culled = 0;
    }
    // the chunk sums this kernel adds to at the end start from zero (n <= 16 384: at most 17 of them)
    if (chunk_sums && threadIdx.x <= (n + (uint32_t)kEmitChunk - 1u) / (uint32_t)kEmitChunk) chunk_sums[threadIdx.x] = 0ull;
    __syncthreads();
    const uint32_t L = (((n + kBigWaves - 1) / kBigWaves) + 63u) & ~63u;
    if (L <= 2 * 64)
        small_sort_in_lds<2>(keys, n, L, vout, rect, rrect, m_out, culled_out, sh, skey, sval, &s_min, &s_max, &s_culled, chunk_sums, row_of);
    else if (L <= 4 * 64)
        small_sort_in_lds<4>(keys, n, L, vout, rect, rrect, m_out, culled_out, sh, skey, sval, &s_min, &s_max, &s_culled, chunk_sums, row_of);
    else if (L <= 8 * 64)
        small_sort_in_lds<8>(keys, n, L, vout, rect, rrect, m_out, culled_out, sh, skey, sval, &s_min, &s_max, &s_culled, chunk_sums, row_of);
    else
        small_sort_in_lds<kBucketRounds>(keys, n, L, vout, rect, rrect, m_out, culled_out, sh, skey, sval, &s_min, &s_max,
                                         &s_culled, chunk_sums, row_of);
}

struct PassPlan {
    int nblocks, nbp, nquads;
    bool self_scan;           // scan == kScanSelf
    int scan;                 // kScanSelf / kScanQuads / kScanRows, by size
    uint32_t *table, *totals, *quad_totals;
};

PassPlan plan_for(void *temp, int64_t bound, int bins = kBins) {
    PassPlan p;
    p.nblocks = (int)((bound + kItems - 1) / kItems);
    p.nquads = (p.nblocks + kQuad - 1) / kQuad;
    p.self_scan = bins == kBins && p.nblocks <= kSelfScanBlocks;
    p.scan = p.self_scan ? kScanSelf : ((bins == kBins && p.nquads <= kQuadScanQuads) ? kScanQuads : kScanRows);
    if (p.scan == kScanRows) p.nquads = (p.nquads + 7) & ~7;    // whole groups of 8 workgroups: one per XCD (count_kernel, scatter_kernel)
    p.nbp = p.self_scan ? p.nblocks : p.nquads * kQuad;
    p.table = (uint32_t *)temp;
    p.totals = p.table + (size_t)bins * p.nquads * kQuad;
    p.quad_totals = p.totals + 2 * (size_t)bins;      // behind the row totals and the splitters
    return p;
}

// Bits per pass: the key bits spread evenly over ceil(key_bits / 8) passes (13 tile-id bits -> 7 + 6:
// half the digit rows of 8 + 5), at least 6.
inline int pass_bits(int key_bits) {
    const int passes = (key_bits + 7) / 8;
    const int w = (key_bits + passes - 1) / passes;
    return w < 6 ? 6 : w;
}

template <typename Key, int MODE, int BITS>
void launch_pass(const PassPlan &p, const Key *kc, const uint32_t *vc, Key *ka, uint32_t *va, const uint32_t *n_dev,
                 int64_t bound, int shift, uint32_t *m_out, uint32_t *culled, const TileRect *rect, TileRect *rrect,
                 hipStream_t s, uint32_t *samples_out = nullptr, uint32_t sample_step = 0, const uint32_t *cin = nullptr,
                 uint32_t *cout = nullptr, bool carry = false) {
    constexpr uint32_t mask = (1u << BITS) - 1u;
    constexpr bool first = (MODE & kModeFirst) != 0;
    if (p.self_scan) {
        count_kernel<Key, true, first><<<p.nblocks, kThreads, 0, s>>>(kc, n_dev, (uint32_t)bound, shift, mask, p.table,
                                                                       p.nbp, culled, nullptr, nullptr, nullptr, samples_out,
                                                                       sample_step);
        scatter_kernel<Key, kScanSelf, MODE, BITS><<<(p.nblocks + 7) & ~7, kThreads, 0, s>>>(kc, vc, ka, va, n_dev, (uint32_t)bound,
                                                                                   shift, p.table, p.totals, p.nbp, m_out, rect,
                                                                                   rrect);
    } else if (p.scan == kScanQuads) {
        count_kernel<Key, false, first><<<p.nquads, kQuad * kThreads, 0, s>>>(kc, n_dev, (uint32_t)bound, shift, mask,
                                                                               p.table, p.nbp, culled, nullptr, p.quad_totals,
                                                                               nullptr, samples_out, sample_step);
        scatter_kernel<Key, kScanQuads, MODE, BITS><<<(p.nblocks + 7) & ~7, kThreads, 0, s>>>(
            kc, vc, ka, va, n_dev, (uint32_t)bound, shift, p.table, p.totals, p.nbp, m_out, rect, rrect, nullptr, p.quad_totals);
    } else {
        count_kernel<Key, false, first><<<p.nquads, kQuad * kThreads, 0, s>>>(kc, n_dev, (uint32_t)bound, shift, mask,
                                                                               p.table, p.nbp, culled, nullptr, nullptr, nullptr,
                                                                               samples_out, sample_step);
        row_scan_kernel<<<1u << BITS, kThreads, 0, s>>>(p.table, p.nbp, p.totals);
        const unsigned grid = (unsigned)((p.nblocks + 7) & ~7);     // (whole groups of 8: see chunk_id)
        if constexpr (sizeof(Key) == 4 && BITS == 8) {
            if (carry) {
                scatter_kernel<Key, kScanRows, MODE, BITS, 0, true><<<grid, kThreads, 0, s>>>(
                    kc, vc, ka, va, n_dev, (uint32_t)bound, shift, p.table, p.totals, p.nbp, m_out, rect, rrect, nullptr, nullptr,
                    nullptr, cin, cout);
                return;
            }
        }
        scatter_kernel<Key, kScanRows, MODE, BITS><<<grid, kThreads, 0, s>>>(kc, vc, ka, va, n_dev, (uint32_t)bound, shift,
                                                                             p.table, p.totals, p.nbp, m_out, rect, rrect);
    }
}

template <typename Key>
hipError_t sort_impl(void *temp, Key *&kc, Key *&ka, uint32_t *&vc, uint32_t *&va, const uint32_t *n_dev,
                     int64_t bound, int key_bits, hipStream_t s) {
    if (bound <= 0) return hipSuccess;
    const PassPlan p = plan_for(temp, bound);
    const int bits = pass_bits(key_bits);
    for (int shift = 0; shift < key_bits; shift += bits) {
        if (bits == 6)
            launch_pass<Key, kModePlain, 6>(p, kc, vc, ka, va, n_dev, bound, shift, nullptr, nullptr, nullptr, nullptr, s);
        else if (bits == 7)
            launch_pass<Key, kModePlain, 7>(p, kc, vc, ka, va, n_dev, bound, shift, nullptr, nullptr, nullptr, nullptr, s);
        else
            launch_pass<Key, kModePlain, 8>(p, kc, vc, ka, va, n_dev, bound, shift, nullptr, nullptr, nullptr, nullptr, s);
        Key *tk = kc; kc = ka; ka = tk;
        uint32_t *tv = vc; vc = va; va = tv;
    }
    return hipGetLastError();
}

}  // namespace

hipError_t radix_sort_pairs_u32(void *temp, uint32_t *&keys_cur, uint32_t *&keys_alt, uint32_t *&vals_cur,
                                uint32_t *&vals_alt, const uint32_t *n_dev, int64_t bound, int key_bits,
                                hipStream_t s) {
    return sort_impl<uint32_t>(temp, keys_cur, keys_alt, vals_cur, vals_alt, n_dev, bound, key_bits, s);
}

hipError_t radix_sort_pairs_u16(void *temp, uint16_t *&keys_cur, uint16_t *&keys_alt, uint32_t *&vals_cur,
                                uint32_t *&vals_alt, const uint32_t *n_dev, int64_t bound, int key_bits,
                                hipStream_t s) {
    return sort_impl<uint16_t>(temp, keys_cur, keys_alt, vals_cur, vals_alt, n_dev, bound, key_bits, s);
}


// Which route sorts the depth keys of a frame (gsx_internal.h: DepthRoute).  What decides is how many keys are
// KEPT -- the sample-partitioned route drops the others in its one pass over the n keys and sorts the rest in
// LDS buckets -- and that number is known to the caller from an earlier frame of the view (GsxParams.kept_hint;
// without it: n).  A bucket is sorted in LDS up to kBucketCap = 16 384 keys; the route keeps the MEAN bucket below
// ~6 000 (with 8 samples per bucket a bucket of 2.8x the mean has probability ~1e-4), and a bucket that does
// not fit is still sorted correctly, through global memory: a wrong hint costs time, never the order.
//   n <= 16 384                    one workgroup does everything in LDS
//   kept <= 1.5M                   256 buckets   (1M Gaussians at 1080p; one rank's strip of 5M at 4K)
//   beyond                         four compacting LSD passes of 8 bits
// Measured, depth sort alone (tools/attic/sort_probe.py under rocprofv3, round 3):  1.5M keys: 256 buckets 83 us (72 with
// the previous frame's splitters), LSD 107;  2.2M: LSD 137, 1024 buckets 179;  5M: LSD 260, 1024 buckets 284;  5M
// of which 1/8 kept (a rank's strip): 256 buckets 94 (70 hinted).  The 1024-bucket variant (8192 samples, 1023
// splitters, hybrid LDS-word / ballot matching) was built for 1.5M .. 6M kept keys and LOSES to the LSD passes at
// every size: a 2048-key chunk holds two keys per bucket, so its scatter writes 8-byte runs (125 us at 5M against
// 31 us for a plain 8-bit pass), and ranking 8192 samples costs 40 us.  It stays selectable for the tests (the
// partition machinery is the same templates), no frame takes it.
// Re-measured after the XCD-contiguous chunk mapping (which makes short runs much cheaper: scatter_kernel), kept =
// 0.95 n: 5M: LSD 220 (102 of them the last pass' rectangle gather), 1024 buckets 238 (scatter 80, bucket sort 77,
// count 35, sample 39), 512 buckets 293 (buckets of 9 300 keys overflow the LDS too often: bucket sort 166);  3M: LSD
// 148, 512 buckets 161;  2.2M: LSD 124, 512 buckets 135, 1024 buckets 158.  Without their 40 us sample kernel --
// i.e. with splitters handed over by the previous frame, which today exists for 256 buckets only -- 512 buckets would
// win by ~25 us between 1.5M and 3.5M kept keys and 1024 buckets by ~20 us at 5M: not built.  (And the LSD passes have
// since stopped gathering the rectangles -- scatter_kernel CARRY: 5M 222 -> 186 us, 2.2M 123 -> 106 us --, which takes
// most of that margin away.)
constexpr int64_t kSampledMin = 8 * kSamples, kKeptMax256 = 1536 * 1024;

DepthRoute depth_sort_route(int64_t n, int64_t kept_hint) {
    const int force = knob("GSX_DEPTH_SORT", -1);   // test library only: a DepthRoute
    if (n <= kBucketCap) return kDepthOneWorkgroup;
    if (force == kDepthLsd || n < kSampledMin) return kDepthLsd;
    if (force == kDepth256 || force == kDepth1024) return (DepthRoute)force;
    const int64_t kept = kept_hint > 0 && kept_hint < n ? kept_hint : n;
    return kept <= kKeptMax256 ? kDepth256 : kDepthLsd;
}

// The partition pass of the sampled routes: keys0 -> (keys1, vals_alt, rrect in partition order), values generated.
template <int NB>
static void launch_partition(const PassPlan &p, uint32_t *keys0, uint32_t *keys1, uint32_t *vals_cur, uint32_t *vals_alt,
                             int64_t n, uint32_t *m_dev, uint32_t *culled_dev, const TileRect *rect, TileRect *rrect,
                             const uint32_t *splitters, const uint32_t *hdr, uint32_t *samples_out, uint32_t step,
                             unsigned long long *zero_sums, uint32_t nsums, hipStream_t s, const uint32_t *row_of = nullptr) {
    (void)vals_cur;     // (FIRST mode generates the values: `vin` of its scatter is GsxParams.row_of_index, or null)
    if (NB == kBins && p.self_scan) {
        count_kernel<uint32_t, true, true, kBins><<<p.nblocks, kThreads, 0, s>>>(keys0, nullptr, (uint32_t)n, 0, 255u, p.table,
                                                                                 p.nbp, culled_dev, splitters, nullptr, hdr,
                                                                                 samples_out, step, zero_sums, nsums);
        scatter_kernel<uint32_t, kScanSelf, kModeFirst, 8, kBins><<<(p.nblocks + 7) & ~7, kThreads, 0, s>>>(
            keys0, row_of, keys1, vals_alt, nullptr, (uint32_t)n, 0, p.table, p.totals, p.nbp, m_dev, rect, rrect, splitters,
            nullptr, hdr);
    } else if (NB == kBins && p.scan == kScanQuads) {
        count_kernel<uint32_t, false, true, kBins><<<p.nquads, kQuad * kThreads, 0, s>>>(
            keys0, nullptr, (uint32_t)n, 0, 255u, p.table, p.nbp, culled_dev, splitters, p.quad_totals, hdr, samples_out, step,
            zero_sums, nsums);
        scatter_kernel<uint32_t, kScanQuads, kModeFirst, 8, kBins><<<(p.nblocks + 7) & ~7, kThreads, 0, s>>>(
            keys0, row_of, keys1, vals_alt, nullptr, (uint32_t)n, 0, p.table, p.totals, p.nbp, m_dev, rect, rrect, splitters,
            p.quad_totals, hdr);
    } else {
        count_kernel<uint32_t, false, true, NB><<<p.nquads, kQuad * kThreads, 0, s>>>(
            keys0, nullptr, (uint32_t)n, 0, 255u, p.table, p.nbp, culled_dev, splitters, nullptr, NB == kBins ? hdr : nullptr,
            NB == kBins ? samples_out : nullptr, step, zero_sums, nsums);
        row_scan_kernel<<<NB, kThreads, 0, s>>>(p.table, p.nbp, p.totals);
        scatter_kernel<uint32_t, kScanRows, kModeFirst, 8, NB><<<(p.nblocks + 7) & ~7, kThreads, 0, s>>>(
            keys0, row_of, keys1, vals_alt, nullptr, (uint32_t)n, 0, p.table, p.totals, p.nbp, m_dev, rect, rrect, splitters,
            nullptr, NB == kBins ? hdr : nullptr);
    }
}

// What the projection launch needs to rank the sample itself (SampleHint): where sort_depth_sampled(route, temp, .., n, ..)
// will look for its splitters and which chunk sums it adds to.  Only the 256-bucket route of a frame of >= 8 192 Gaussians.
SampleHint depth_presample(DepthRoute route, void *temp, int64_t n, uint64_t *chunk_sums, const uint32_t *row_of) {
    SampleHint h;
    if (route != kDepth256 || n < kSamplesMax || n >= ((int64_t)1 << 32)) return h;
    const PassPlan p = plan_for(temp, n, kBins);
    h.splitters = p.totals + kBins;
    h.chunk_sums = reinterpret_cast<unsigned long long *>(chunk_sums);
    h.nsums = (uint32_t)((n + kEmitChunk - 1) / kEmitChunk) + 1u;
    h.ns = (uint32_t)kSamples;
    h.row_of = row_of;
    return h;
}

// Same contract as sort_depth_compact.  keys0 / keys1 / vals: n words each; on return vals_cur[0 .. *m_dev)
// = Gaussian index of each depth rank, rrect[rank] = rect[index].  route: kDepthOneWorkgroup / kDepth256 /
// kDepth1024.  kept_hint: see depth_sort_route (here it only sizes the sample).  lds_cap: bucket size above which
// the through-memory path is taken (tests shrink it).  chunk_sums (or nullptr): ceil(n / 1024) + 1 words that receive
// the tile count of every 1024 consecutive ranks -- what chunk_sums_kernel would compute from rrect.
hipError_t sort_depth_sampled(DepthRoute route, void *temp, uint32_t *keys0, uint32_t *keys1, uint32_t *&vals_cur,
                              uint32_t *&vals_alt, int64_t n, int64_t kept_hint, uint32_t *m_dev, uint32_t *culled_dev,
                              const TileRect *rect, TileRect *rrect, uint32_t lds_cap, uint64_t *chunk_sums,
                              const SortHints &hints, hipStream_t s, const uint32_t *row_of, bool presampled) {
    if (n <= 0) return hipSuccess;
    unsigned long long *cs = reinterpret_cast<unsigned long long *>(chunk_sums);
    if (route == kDepthOneWorkgroup && n <= kBucketCap && lds_cap == 0) {   // everything fits one workgroup's LDS: one launch
        small_depth_sort_kernel<<<1, kBigThreads, 0, s>>>(keys0, (uint32_t)n, vals_cur, rect, rrect, m_dev, culled_dev, cs, row_of);
        return hipGetLastError();
    }
    const int nb = route == kDepth1024 ? kSortBinsMax : kBins;
    const PassPlan p = plan_for(temp, n, nb);
    uint32_t *splitters = p.totals + nb;               // behind the row totals
    // 8 samples per bucket -- of the keys that are kept: a window that keeps less than half of the Gaussians (a
    // rank's strip) takes the larger sample as well
    const int64_t kept = kept_hint > 0 && kept_hint < n ? kept_hint : n;
    const uint32_t ns = (nb > kBins || 2 * kept < n) && n >= kSamplesMax ? (uint32_t)kSamplesMax : (uint32_t)kSamples;
    const uint32_t nsums = (uint32_t)((n + kEmitChunk - 1) / kEmitChunk) + 1u;
    if (lds_cap == 0 || lds_cap > (uint32_t)kBucketCap) lds_cap = kBucketCap;
    // GsxParams.hints (256-bucket route): the count kernel leaves a sample of the kept keys for the next frame's
    // splitters, and with GSX_FLAG_HINTS_VALID the splitters of THIS frame are the ones the previous frame left --
    // no sample kernel on the frame's critical path
    const bool hinted = nb == kBins && hints.header != nullptr;
    // (the bucket kernel leaves the next frame's splitters itself: no sample for the compositing launch to rank)
    uint32_t *samples_out = hinted && !hints.next_splitters ? hints.samples : nullptr;
    const uint32_t step = hinted && n >= kSamples ? (uint32_t)(n / kSamples) : 0u;
    if (nb > kBins) {
#ifndef GSX_TEST_HOOKS
        // the 1 024-bucket route measured slower than the LSD passes everywhere and nothing in the
        // shipping library selects it (depth_sort_route): its kernels are compiled into libgsx_test.so only, where
        // gsx_debug_depth_sort and the GSX_DEPTH_SORT knob keep it tested
        return hipErrorNotSupported;
#else
        sample_rank_kernel<kSortBinsMax><<<ns / kRankPerGroup, kThreads, 0, s>>>(keys0, (uint32_t)n, ns, splitters, cs, nsums);
        launch_partition<kSortBinsMax>(p, keys0, keys1, vals_cur, vals_alt, n, m_dev, culled_dev, rect, rrect, splitters, nullptr,
                                       nullptr, 0u, nullptr, 0u, s, row_of);
        bucket_sort_kernel<kSortBinsMax><<<kSortBinsMax, kBigThreads, 0, s>>>(p.totals, p.table, 0, keys1, vals_alt, keys0, vals_cur,
                                                                              rect, rrect, lds_cap, cs, nullptr, nullptr, m_dev);
#endif
    } else {
        const bool use = hinted && hints.use;
        // (presampled: spare workgroups of the projection launch have ranked a sample into `splitters` and zeroed the chunk
        // sums already: depth_presample)
        if (!use && !presampled) sample_rank_kernel<kBins><<<ns / kRankPerGroup, kThreads, 0, s>>>(keys0, (uint32_t)n, ns, splitters, cs, nsums);
        launch_partition<kBins>(p, keys0, keys1, vals_cur, vals_alt, n, m_dev, culled_dev, rect, rrect,
                                use ? hints.splitters : splitters, use ? hints.header : nullptr, samples_out, step,
                                use ? cs : nullptr, nsums, s, row_of);
        bucket_sort_kernel<kBins><<<kBins, kBigThreads, 0, s>>>(p.totals, p.table, p.self_scan ? p.nblocks : 0, keys1, vals_alt,
                                                                keys0, vals_cur, rect, rrect, lds_cap, cs,
                                                                hinted ? hints.next_splitters : nullptr, hints.header, m_dev);
    }
    return hipGetLastError();
}

// The depth sort of the whole-path entry point: 4 passes over the IEEE bits of z_view.  Pass 0 keeps only
// the keys < kEmptyKey (*m_dev = how many, *culled_dev += how many were == kCulledKey; culled_dev must be
// zero when the first kernel runs), the last pass writes rrect[rank] = rect[index] instead of the keys.
// On return vals_cur[0 .. *m_dev) = Gaussian index of each depth rank.
hipError_t sort_depth_compact(void *temp, uint32_t *keys0, uint32_t *keys1, uint32_t *&vals_cur, uint32_t *&vals_alt,
                              int64_t n, uint32_t *m_dev, uint32_t *culled_dev, const TileRect *rect, TileRect *rrect,
                              hipStream_t s, uint32_t *samples_out, uint32_t *carry0, uint32_t *carry1, const uint32_t *row_of) {
    if (n <= 0) return hipSuccess;
    const PassPlan p = plan_for(temp, n);
    uint32_t *kc = keys0, *ka = keys1;
    // carry0 / carry1 (n words each, or null): the packed rectangles travel with the items (scatter_kernel, CARRY) --
    // the caller offers them when the tile grid fits 8-bit coordinates; worth it from ~1M keys (a row-scan pass)
    const bool carry = carry0 && carry1 && p.scan == kScanRows && n >= (1 << 20);
    uint32_t *cc = carry0, *ca = carry1;
    auto flip = [&]() {
        uint32_t *tk = kc; kc = ka; ka = tk;
        uint32_t *tv = vals_cur; vals_cur = vals_alt; vals_alt = tv;
        uint32_t *tc = cc; cc = ca; ca = tc;
    };
    // (samples_out: GsxParams.hints -- pass 0's count kernel leaves the sample of kept keys a later frame's splitters
    // are ranked from, should that frame take the 256-bucket route: a rank's strip does, once its kept count is known)
    launch_pass<uint32_t, kModeFirst, 8>(p, kc, row_of, ka, vals_alt, nullptr, n, 0, m_dev, culled_dev, carry ? rect : nullptr,
                                         nullptr, s, samples_out, samples_out && n >= kSamples ? (uint32_t)(n / kSamples) : 0u,
                                         nullptr, ca, carry);
    flip();
    for (int shift = 8; shift < 24; shift += 8) {
        launch_pass<uint32_t, kModePlain, 8>(p, kc, vals_cur, ka, vals_alt, m_dev, n, shift, nullptr, nullptr, nullptr, nullptr, s,
                                             nullptr, 0u, cc, ca, carry);
        flip();
    }
    launch_pass<uint32_t, kModeFinal, 8>(p, kc, vals_cur, ka, vals_alt, m_dev, n, 24, nullptr, nullptr, rect, rrect, s, nullptr, 0u,
                                         cc, nullptr, carry);
    flip();
    return hipGetLastError();
}

}  // namespace gsx
