// The NB - 1 splitters of the depth sort's partition pass from a SAMPLE of the keys, ranked by many workgroups at once.
// Two callers: sample_rank_kernel (gsx_sort.hip: the keys exist) and the spare workgroups of the projection launch
// (gsx_project.hip) on a frame that has no splitters from an earlier frame of its view -- they compute the sampled
// Gaussians' keys themselves (cull plane and view depth: 2 048 dot products), so the ranking runs BESIDE the projection
// instead of behind it: 13.8 us off a view's first frame at 1M Gaussians (round 6).
#pragma once

#include "gsx_internal.h"

namespace gsx {

constexpr int kSampleThreads = 256;
// The NB - 1 splitters of the partition pass: regular quantiles of the VALID ones among `ns` regularly spaced keys
// (key at index i n / ns; a key >= kEmptyKey is dropped by the sort and says nothing about where the kept keys
// lie -- on a rank that owns 1/8 of the frame 7 of 8 samples are of that kind, which is why such a frame takes
// 8192 samples instead of 2048): splitters[j] = the valid sample of rank floor(j V / NB), V = number of valid
// samples, splitters[0] = 0.  A single workgroup sorting the samples took 41 us (one CU doing 4 LDS radix
// passes); here every workgroup packs the valid samples into LDS in index order (a thread owns ns / 256
// consecutive samples, all loads in flight at once) and every sample's rank is counted directly --
// #{j : s[j] < s[i]} + #{j < i : s[j] == s[i]} -- by 16 lanes that share the V comparisons, 16 samples per
// workgroup; the sample that finds itself on a quantile writes the splitter(s) it is.
constexpr int kRankLanes = 16, kRankPerGroup = kSampleThreads / kRankLanes;   // 16 samples per 256-thread workgroup
constexpr int kSamplesMax = kSortSamplesMax, kSamplesPerThreadMax = kSamplesMax / kSampleThreads;
inline uint32_t sample_rank_workgroups(uint32_t ns) { return ns / (uint32_t)kRankPerGroup; }

// key_at(i): the depth key of index i (kCulledKey / kEmptyKey for a Gaussian the sort will drop).  All 256 threads of
// workgroup `wg` of `nwg` call this.  sm: ns words of LDS, s_wave: 4 (the caller's: the projection kernels lend the
// spare workgroups the LDS their other workgroups stage spherical harmonics in).
template <int NB, typename KeyAt>
__device__ __forceinline__ void sample_rank_body(KeyAt key_at, uint32_t n, uint32_t ns, uint32_t *__restrict__ splitters,
                                                 unsigned long long *__restrict__ chunk_sums, uint32_t nsums, uint32_t wg,
                                                 uint32_t nwg, uint32_t *sm, uint32_t *s_wave) {
    // the chunk sums the bucket kernel adds to start from zero (the first kernel of the sort has threads to spare)
    for (uint32_t k = wg * (uint32_t)kSampleThreads + threadIdx.x; chunk_sums && k < nsums; k += nwg * (uint32_t)kSampleThreads)
        chunk_sums[k] = 0ull;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t per = ns / (uint32_t)kSampleThreads;           // 8 or 32 (ns = 2048 / 8192)
    uint32_t v[kSamplesPerThreadMax];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kSamplesPerThreadMax; ++k) {
        v[k] = kCulledKey;
        if ((uint32_t)k < per) v[k] = key_at((uint32_t)(((uint64_t)(threadIdx.x * per + (uint32_t)k) * n) / ns));
        mine += v[k] < kEmptyKey;
    }
    uint32_t x = mine;   // inclusive scan over the wave, then over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) s_wave[w] = x;
    __syncthreads();
    uint32_t at = x - mine, valid = 0;
#pragma unroll
    for (int k = 0; k < kSampleThreads / 64; ++k) {
        at += k < w ? s_wave[k] : 0u;
        valid += s_wave[k];
    }
#pragma unroll
    for (int k = 0; k < kSamplesPerThreadMax; ++k)
        if (v[k] < kEmptyKey) sm[at++] = v[k];
    __syncthreads();
    if (valid == 0) {       // nothing reaches a tile: every key goes to bucket 0 (and is dropped there)
        if (wg == 0)
            for (int k = threadIdx.x; k < NB; k += kSampleThreads) splitters[k] = 0u;
        return;
    }
    const uint32_t i = wg * (uint32_t)kRankPerGroup + (threadIdx.x / kRankLanes);
    const uint32_t part = threadIdx.x % kRankLanes;
    if (wg == 0 && threadIdx.x == 0) splitters[0] = 0u;
    if (wg * (uint32_t)kRankPerGroup >= valid) return;
    const uint32_t mykey = i < valid ? sm[i] : 0u;
    uint32_t c = 0;
    // lane `part` compares against samples part, part + 16, part + 32, ... (conflict-free across the 16 lanes)
    for (uint32_t j = part; j < valid; j += kRankLanes) {
        const uint32_t y = sm[j];
        c += (y < mykey) | ((y == mykey) & (j < i));
    }
#pragma unroll
    for (int o = kRankLanes / 2; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if (part == 0 && i < valid) {
        // the quantiles j with floor(j valid / NB) == c (none, one, or several when valid < NB)
        for (uint32_t j = (c * (uint32_t)NB + valid - 1u) / valid; j < (uint32_t)NB && (j * valid) / NB == c; ++j)
            if (j) splitters[j] = mykey;
    }
}

}  // namespace gsx
