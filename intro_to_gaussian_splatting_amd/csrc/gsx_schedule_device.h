// The compositing schedule from the tiles' costs in the PREVIOUS frame (GsxParams.hints: records staged until the tile
// was done, gsx_plan.h; a frame's first schedule, tile_schedule_kernel, goes by list length), put together by eight
// spare workgroups of the projection launch (the frame's first kernel and, of those in front of the compositing
// launch, its longest: 25 us at 1M Gaussians, 116 us at 5M) -- one per XCD:
//   * the window's tiles are dealt to the XCDs in chunks of ~32 consecutive tile ids (ids are column-major: half a
//     tile column at 1080p; chunk c goes to XCD c % 8, see sched_cut): vertical neighbours, which share most of their
//     Gaussians, stay in one 4 MiB L2, and every XCD gets its share of a dense region;
//   * inside an XCD the tiles are ranked by cost (256 classes between its cheapest and its most expensive tile) and
//     the compositing launch hands them to the XCD's 128 SIMDs round by round, alternately forwards and backwards
//     (blend_tile16_kernel), so that every SIMD gets one tile of every length class.
// Round 2 ranked the tiles over the whole frame (tile_schedule_kernel: one workgroup, 10 us at 1080p, 22 us at 4K, on
// the frame's critical path; 3.3x the algorithmic record traffic because neighbours no longer shared an L2); its own
// measurements had the two-column variant at 270 vs 262 us on the uniform scene and 454 vs 488 us on the heavy-tailed
// one, with a quarter less traffic -- with the schedule off the critical path the choice is this one.  (One workgroup
// ranking the whole frame behind the projection was tried first in round 3: 256 threads need 29 us for 8 000 tiles and
// 190 us for 32 000 -- longer than the kernel that was to hide it; an XCD's share is an eighth of that.)
//
// Layout: sched[x * cap + k] = the tile with the k-th longest list of XCD x, k < header[kHintXcdTiles + x]; cap =
// sched_cap(nt, nwy).  header[kHintSched] = nt when the schedule is there, 0 when the lengths on file are for another
// window (the compositing launch then takes the tiles in index order).
#pragma once

#include "gsx_internal.h"

namespace gsx {

struct SchedJob {
    const uint32_t *lens;       // cost of every tile of the window, left by the previous frame's compositing launch (gsx_plan.h)
    uint32_t *sched, *header;
    uint32_t nt, nwy, cap;
};


// (SchedCut, sched_cut, sched_cap: gsx_plan.h -- the hints buffer is sized from them)
// chunks of XCD x that hold s + 1 tiles (they come first among its chunks x, x + 8, ...)
__host__ __device__ inline uint32_t sched_big_chunks(const SchedCut &c, uint32_t x) {
    return c.rem > x ? (c.rem - x + kSchedXcds - 1u) / kSchedXcds : 0u;
}
__host__ __device__ inline uint32_t sched_count(uint32_t nt, uint32_t /*nwy*/, uint32_t x) {
    const SchedCut c = sched_cut(nt);
    const uint32_t n1 = sched_big_chunks(c, x);
    return n1 * (c.s + 1u) + (c.k - n1) * c.s;
}
// the j-th tile of XCD x
__host__ __device__ inline uint32_t sched_tile_of(uint32_t j, uint32_t x, const SchedCut &c) {
    const uint32_t n1 = sched_big_chunks(c, x), big = n1 * (c.s + 1u);
    uint32_t q, r;
    if (j < big) {
        q = j / (c.s + 1u);
        r = j % (c.s + 1u);
    } else {
        q = n1 + (j - big) / c.s;     // (c.s > 0 here: j >= big means a small chunk holds tile j)
        r = (j - big) % c.s;
    }
    const uint32_t ch = x + kSchedXcds * q;
    return ch * c.s + (ch < c.rem ? ch : c.rem) + r;
}

// All 256 threads of workgroup x (0 .. 7) call this.  raw_cost(tile): the tile's cost, bit 31 set when it runs as a long tile
// on helper workgroups (to the hand-out it is empty, to the XCD's total it counts).
template <typename RawCost>
__device__ __forceinline__ void schedule_xcd(const SchedJob &job, uint32_t x, RawCost raw_cost) {
    constexpr uint32_t kClasses = 256;
    __shared__ uint32_t hist[kClasses];
    __shared__ uint32_t s_lo, s_hi, s_total, s_wsum[4];
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, w = tid >> 6;
    const uint32_t nt = job.nt;
    const SchedCut cut = sched_cut(nt);
    const uint32_t mine_n = sched_count(nt, job.nwy, x);
    auto cost_of = [&](uint32_t j) -> uint32_t {
        const uint32_t l = raw_cost(sched_tile_of(j, x, cut));
        return (l >> 31) ? 0u : l;
    };
    uint32_t mn = 0xFFFFFFFFu, mx = 0u, total = 0u;
    for (uint32_t j = tid; j < mine_n; j += nthreads) {
        const uint32_t raw = raw_cost(sched_tile_of(j, x, cut)), l = (raw >> 31) ? 0u : raw;
        mn = min(mn, l);
        mx = max(mx, l);
        total += min(raw & 0x7FFFFFFFu, 1u << 20);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        total += (uint32_t)__shfl_xor((int)total, o);
    }
    if (tid < kClasses) hist[tid] = 0;
    if (tid == 0) {
        s_lo = 0xFFFFFFFFu;
        s_hi = 0u;
        s_total = 0u;
    }
    __syncthreads();
    if (lane == 0 && mn <= mx) {
        atomicMin(&s_lo, mn);
        atomicMax(&s_hi, mx);
        atomicAdd(&s_total, total);
    }
    __syncthreads();
    const uint32_t shortest = s_lo;
    const float per_entry = (float)(kClasses - 1) / (float)max(s_hi - shortest, 1u);
    auto cls = [&](uint32_t l) -> uint32_t {   // class 0 = the longest lists
        return (kClasses - 1u) - min((uint32_t)((float)(l - shortest) * per_entry), kClasses - 1u);
    };
    for (uint32_t j = tid; j < mine_n; j += nthreads) atomicAdd(&hist[cls(cost_of(j))], 1u);
    __syncthreads();
    const uint32_t mine = tid < kClasses ? hist[tid] : 0u;     // (256 threads: one class each)
    uint32_t v = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)v, o);
        if (lane >= o) v += y;
    }
    if (lane == 63) s_wsum[w] = v;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < w; ++k) before += s_wsum[k];
    if (tid < kClasses) hist[tid] = before + v - mine;   // first slot of this class
    __syncthreads();
    uint32_t *out = job.sched + (size_t)x * job.cap;
    for (uint32_t j = tid; j < mine_n; j += nthreads) {
        const uint32_t t = sched_tile_of(j, x, cut);
        out[atomicAdd(&hist[cls(cost_of(j))], 1u)] = t;
    }
    if (tid == 0) {
        job.header[kHintXcdTiles + x] = mine_n;
        job.header[kHintXcdCost + x] = s_total;
        if (x == 0) job.header[kHintSched] = nt;
    }
}

// The projection launch's spare workgroups: the schedule from the tiles' COSTS in the previous frame (GsxParams.hints:
// records staged until the tile was done, gsx_plan.h).
__device__ __forceinline__ void schedule_from_lengths(const SchedJob &job, uint32_t x) {
    if (job.header[kHintLens] != job.nt) {      // nothing usable on file: the compositing launch falls back to index order
        if (x == 0 && threadIdx.x == 0) job.header[kHintSched] = 0u;
        return;
    }
    schedule_xcd(job, x, [&](uint32_t t) -> uint32_t { return job.lens[t]; });
}

}  // namespace gsx
