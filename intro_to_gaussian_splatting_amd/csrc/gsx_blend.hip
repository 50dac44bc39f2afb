// Stage 2 on gfx950: front-to-back alpha compositing of each tile's depth-ordered list.
//
// Reference behaviour restated (paths relative to the reference repository):
//   splat/gaussian_scene.py:146-171  render_pixel: T=1, C=0; alpha = w * sigmoid(opacity);
//                                    test = T(1-alpha); if test < 1e-6 return C (before adding);
//                                    C += T alpha c; T = test
//   splat/utils.py:357-365           w = exp(-1/2 d Q d^T), d = mean - pixel, full 2x2 Q
//   splat/gaussian_scene.py:173-198  render_tile: every listed Gaussian at every pixel of the tile,
//                                    pixel centres at integer coordinates
//   splat/gaussian_scene.py:200-238  render_image: image[x][y][c]
//
// CDNA4 mapping.  Under these semantics nothing is culled per pixel, so a tile costs
// |list| x 256 weight evaluations (~11 VALU ops + one v_exp_f32 each) against 48 B of record
// traffic per list entry: the kernel is VALU-bound, not HBM-bound.  Hence:
//   - one 64-lane wavefront per 16x16 tile, 4 pixels per lane along the memory-contiguous image
//     axis, so each record fetched from LDS is amortised over 4 evaluations, the terms of the
//     quadratic form that depend only on the shared coordinate are computed once per lane, and
//     the per-lane output is 48 contiguous bytes;
//   - (round 4) the tile's four 8x8 BLOCKS -- 16 lanes each -- walk lists of their own: a record whose alpha
//     stays below 2^-26 on a whole block is not on that block's list (it would change nothing there), the
//     wave's instructions read four different records from LDS, and a tile costs its LONGEST block list:
//     64 % of the tile's list on the benchmark scene (see Staged, stage_records);
//   - a single-wave workgroup needs no cross-wave barrier and no LDS flag: "every pixel of the
//     tile is saturated" is one __ballot over the wave;
//   - each lane gathers one 48-B record per batch of 64 (3 x dwordx4) into LDS; the k-loop then
//     reads record k with uniform-address (broadcast) ds_reads;
//   - the saturation test of the reference (stop when T(1-alpha) < 1e-6, before accumulating)
//     is kept exact but off the common path: one v_min3 + v_min + v_cmp per record decides, for
//     the whole wave, whether any of its 256 pixels stops at this record; only then are the
//     per-pixel selects executed;
//   - blockIdx -> tile: large scenes hand the tiles out by list length, so that every SIMD gets the same
//     share of the work (scheduled_tile); otherwise each XCD gets a contiguous stripe of tiles, so
//     neighbouring tiles -- which share most of their Gaussians -- hit the same 4 MiB L2.
//
// Arithmetic.  This translation unit is compiled with -ffp-contract=off and every fused
// operation is an explicit fmaf, so the result does not depend on how the compiler would have
// contracted each template instance: both output layouts give bit-identical pixels.  With
// Q'' = -1/2 log2(e) Q (packed by the projection kernels), e = mean - pixel:
//     alpha = exp2(e0^2 Q''00 + e0 e1 (Q''01 + Q''10) + e1^2 Q''11 + log2 op)    (one v_exp_f32)
// which is the reference's exp(-1/2 e Q e^T) * op.  The three products are NOT summed as they stand: for a
// long, thin, rotated footprint seen from 100 px along its ridge they are ~1e4 each and cancel to O(1), and
// float32 then loses up to 1e-3 of alpha (found on the heavy-tailed workload: 7.7e-4 on a pixel).  With
// M = -Q'' (symmetrised like the reference's e Q e^T does) the square is completed in y:
//     e M e^T = D1 e0^2 + (r11 e1 + h e0)^2,   r11 = sqrt(M11), h = M01 / r11, D1 = M00 - h^2 = det M / M11
//     exponent = fma(-w, w, s0),  w = fma(r11, e1, h e0),  s0 = fma(-(D1 e0), e0, log2 op)
// D1, h, r11 come from the projection kernels (float64 from the float32 Q'' entries, rounded once).  Along
// the ridge w is a small difference of terms of size sqrt(1e4), so the absolute error of the exponent drops
// from eps * 1e4 to eps * 2 |w| * 1e2; for round footprints nothing changes.  Cost: the same 4 operations per
// lane and record for the x-only part and the same 2 FMAs per pixel as the monomial form.  A record whose
// M11 is not positive and finite (only possible for caller-given inverse covariances on the stage-2 entry)
// keeps the monomial coefficients and is flagged; a batch holding one takes the unpacked loop.
// ILL-CONDITIONED footprints (round 5).  The completed square is the exact value to 1e-5; the REFERENCE is not: where its
// four float32 products cancel (needles from ~20:1) its own rounding moves alpha by 1e-4 .. 1e-3, and since the
// restatement this kernel is held against executes the reference's operations in the reference's order, that rounding
// is part of the result.  pack_record flags such records; on the tiles where the rounding can reach 2e-5 (stage_records:
// a needle's ridge) they are evaluated by alpha_ref -- the reference's operations one for one -- where they turn up
// (blend_tile16_ref_kernel, the REF instances of the loops below).
// T(1 - alpha) is evaluated as T - T alpha (1 ulp).
// Coordinates are TILE-RELATIVE (round 3): the lane that stages a record subtracts the tile's origin from the mean
// once (x' = x - x0, y' = y - y0; pixel offsets px', py' = 0 .. tile-1 are exact) and forms c0 = fma(r11, y', h x'),
// the value of w at the tile's origin.  Per lane then e0 = x' - px', c = fma(-h, px', c0) and per pixel
// w = fma(-r11, py', c): the subtraction e1 = y' - py' and the product h e0 disappear from the loop (two packed and
// one plain instruction per record and lane of the tile-16 kernel), and the chain of dependent operations in front
// of v_exp_f32 is one shorter than before -- what a lone wave at the end of a heavy-tailed frame waits for.  The
// cancellation inside w stays bounded by the splat's extent around the tile (hoisting against ABSOLUTE pixel
// coordinates, r11 x 1000, would give back what completing the square won).
#include <type_traits>

#include "gsx_internal.h"
#include "gsx_schedule_device.h"

#ifdef GSX_TEST_HOOKS
// Test library only (gsx_debug.h: gsx_debug_set_blend_probe): when set, every workgroup of blend_tile16_kernel leaves
// (cycles, tile, list length, records staged | flags << 24) there, indexed by blockIdx.x -- who is the frame waiting for?
__device__ uint4 *g_blend_probe = nullptr;
constexpr uint32_t kProbeSecond = 1u << 17;     // a second record per workgroup starts here, the REF instance's per-tile records at twice this (the buffer holds 3 x 2^17)
#endif

namespace gsx {
namespace {

constexpr float kStopRefCpu = 0.000001f;  // gaussian_scene.py:153
constexpr uint32_t kBatchCost = 5;        // staging a batch of 64 costs about as much as compositing five records (a tile's cost, gsx_plan.h)
// A staged batch holds up to 64 records followed by kPad NULL records (log2 op = -inf: alpha = exp2(-inf) = 0 at every
// pixel, colour 0), so that the compositing loops always take whole trips of 4 or 8 records with no per-record branch
// (a branch between the records of a trip keeps the compiler from interleaving their dependent chains, and a wave
// that has its SIMD to itself -- the long, saturated tiles of a heavy-tailed frame -- then pays every chain in full).
// A null record changes nothing under either form of the rule: T alpha = 0, T - 0 = T, fma(0, 0, C) = C.
constexpr int kPad = 8, kSlots = 64 + kPad;
// Round 4: a 16x16 tile is composited as FOUR 8x8 BLOCKS, each with its own list.  The reference lists a Gaussian for a
// tile when its bounding box touches the tile (plus a tile of slack), and under its rules every listed Gaussian is
// evaluated at every pixel -- but a record whose alpha stays below 2^-26 on a whole 8x8 block changes nothing there
// (see stage_records), and only 58 % of the (record, block) combinations of the benchmark scene are above that
// (tools/attic/analyze_skip.py; 91 % of the (record, tile) pairs).  The lane -> pixel assignment puts a block on 16 lanes
// (4 pixels each), every 16-lane group walks ITS block's list -- four different records per wave instruction, read
// from LDS with four addresses -- and a tile costs the LONGEST of its four block lists (64 % of its list on the
// benchmark scene) instead of all of it, at ~10 % more per trip (tools/microbench_trip.hip, V5).
// Block g covers x offsets [8 (g & 1), +8) and y offsets [8 (g >> 1), +8) of the tile.
constexpr int kBlocks = 4;
struct __attribute__((aligned(16))) Staged {
    float4 rec[3][kSlots];            // the staged records, by slot (see read_splat)
    uint16_t list[kBlocks][kSlots];   // per block: the records it keeps, in list order, as BYTE OFFSETS of their slots in rec[0]
                                      // (16 x slot: one shift less per record and lane), padded with a null record's
};

// Contiguous-chunk remap: hardware places block b on XCD b % 8; give XCD x the x-th eighth of
// the tile list.  Bijective for every n_tiles.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    uint32_t xcd = b & 7u, i = b >> 3, q = n >> 3, r = n & 7u;
    uint32_t start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return start + i;
}

// Which tile does tile workgroup b composite?  All the tile workgroups of a frame of up to ~8 000 tiles are
// resident at once (one wave each, 8 per SIMD), so a SIMD is busy for as long as the lists of ITS tiles take: with
// a schedule (sched[k] = tile with the k-th longest list, tile_schedule_kernel) the tiles are handed out by list
// length.  Workgroups b and b + 1024 land on the same SIMD (measured: XCD = b % 8, then round-robin over the XCD's
// 128 SIMDs; workgroups ahead of the tiles in the grid come in multiples of 1024), so round r of 1024 workgroups
// takes the r-th 1024 tiles of the falling-length order, alternately forwards and backwards: every SIMD gets one
// tile of every length class and the sums even out (busiest SIMD / mean 1.11 -> 1.01 list entries at 1M Gaussians).
// The price: neighbouring tiles no longer share an XCD's L2, the record gather misses more (610 MB instead of
// ~200 MB per frame at 1M Gaussians, 2.3 TB/s) -- the kernel is VALU-bound and 33 us faster all the same.  (Tiles
// dealt to the XCDs in two-column chunks and ranked inside each XCD: 460 MB, but 270 us instead of 262, and a
// 16 us schedule kernel; better only on the clustered scene.)  Without a schedule: index order, a contiguous
// eighth of the tiles per XCD.
__device__ __forceinline__ uint32_t scheduled_tile(uint32_t b, uint32_t nt, const uint32_t *__restrict__ sched) {
    if (!sched) return xcd_remap(b, nt);
    const uint32_t round = b >> 10, slot = b & 1023u;
    const uint32_t in_round = min(1024u, nt - (round << 10));
    return sched[(round << 10) + ((round & 1u) ? in_round - 1u - slot : slot)];
}

// The same with the schedule handed over through GsxParams.hints (gsx_schedule_device.h): XCD x = b % 8 composites ITS
// tiles -- the chunks of two tile columns dealt to it -- by falling list length, round by round over its 128 SIMDs
// (workgroups b and b + 1024 share a SIMD), alternately forwards and backwards.  The launch carries 8 x cap tile
// workgroups; one beyond its XCD's tile count has nothing to do (returns nt).  A schedule for another window (header
// word) is ignored: tiles in index order, workgroups beyond nt idle.
__device__ __forceinline__ uint32_t xcd_scheduled_tile(uint32_t b, uint32_t nt, uint32_t cap, const uint32_t *__restrict__ sched,
                                                       const uint32_t *__restrict__ header) {
    if (header[kHintSched] != nt) return b < nt ? xcd_remap(b, nt) : nt;
    const uint32_t x = b & 7u, i = b >> 3, mine = header[kHintXcdTiles + x];
    if (i >= mine) return nt;
    const uint32_t round = i >> 7, slot = i & 127u;
    const uint32_t in_round = min(128u, mine - (round << 7));
    return sched[(size_t)x * cap + (round << 7) + ((round & 1u) ? in_round - 1u - slot : slot)];
}

// Pixels of the output buffer that no tile of the window covers are zeroed by extra workgroups of the
// compositing launch itself (blockIdx >= number of tiles): no memset nodes on the frame path -- they cost
// a 5 us dispatch each, and a hipGraph memset node replayed on another stream than the one it was
// captured on left the border untouched on ROCm 7.2 (tools/attic/debug_border.py).

__device__ __forceinline__ void clear_block(uint32_t cb, const ClearPlan &cp, float *__restrict__ base) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) i = (k < cp.n && cb >= (uint32_t)cp.first[k]) ? k : i;
    const uint32_t per_row = (uint32_t)cp.fw[i] * 3u;
    const uint32_t total = (uint32_t)cp.rows[i] * per_row;
    const uint32_t e0 = (cb - (uint32_t)cp.first[i]) * (uint32_t)kClearFloats;
    const uint32_t e1 = min(total, e0 + (uint32_t)kClearFloats);
    float *origin = base + (int64_t)cp.s0[i] * cp.pitch + (int64_t)cp.f0[i] * 3;
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += 64) {
        const uint32_t row = e / per_row, col = e - row * per_row;
        origin[(int64_t)row * cp.pitch + col] = 0.0f;
    }
}

__global__ void __launch_bounds__(64) clear_kernel(ClearPlan cp, float *__restrict__ base) {
    clear_block(blockIdx.x, cp, base);
}

// Record kinds (Record.c.z, set by pack_record in gsx_project.hip): the completed square; the monomial fallback (a
// caller-given inverse covariance without a finite factorisation); REFERENCE ORDER -- an ill-conditioned footprint, on
// which the reference's float32 evaluation of d Q d^T (splat/utils.py:363-364) is executed operation for operation
// (alpha_ref below): its rounding there is part of the reference's result (1e-4 .. 1e-3 of alpha on 100:1 needles).
constexpr float kRefOrderFlag = 2.0f;      // (the monomial fallback: 1.0f)
enum { kKindSquare = 0, kKindMono = 1, kKindRefOrder = 2 };

struct Splat {  // one staged record, unpacked (wave-uniform values)
    // kKindSquare: as named.  kKindMono: monomial coefficients in (d1, h, r11).
    // kKindRefOrder: (mx, c0) = the mean in FRAME coordinates, (d1, h, r11, my) = (Q00, Q01, Q10, Q11), lop = the opacity factor itself
    float mx, c0, d1, h, r11, lop, cr, cg, cb, my;
    int kind;
    uint32_t blocks;        // bit g: block g of the tile keeps the record (all ones where blocks are not distinguished)
};

// Staged record (stage_records): a = (x', c0, D1, h)  b = (r11, log2 op, r, g)  c = (b, y', kind flag, block bits)
//                  kKindRefOrder: a = (x, y, Q00, Q01) b = (Q10, op, r, g)       c = (b, Q11, kRefOrderFlag, block bits)
__device__ __forceinline__ Splat read_splat(const Staged &sh, uint32_t k) {
    const float4 A = sh.rec[0][k], B = sh.rec[1][k], C = sh.rec[2][k];
    return Splat{A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, C.x, C.y,
                 C.z == kRefOrderFlag ? kKindRefOrder : (C.z != 0.0f ? kKindMono : kKindSquare), __float_as_uint(C.w)};
}

// The reference's alpha of one pixel, operation for operation (splat/utils.py:357-365 as torch executes it,
// oracle/probe_torch_order.py; splat/gaussian_scene.py:164): d = -1/2 (mean - pixel) -- exact --, (1,2) @ (2,2) one FMA
// per output, the final (1,2) @ (2,1) two rounded products and a sum (this file is compiled with -ffp-contract=off),
// exp, times the opacity factor.  px, py: the pixel in FRAME coordinates (integers: exact as floats).  The exponential
// is v_exp_f32 of power * log2(e): within 2 ulp of expf plus 6e-8 |power| log2(e), i.e. < 2e-6 of alpha wherever
// alpha >= 1e-7 -- the one step that is not the reference's bit for bit.
__device__ __forceinline__ float alpha_ref(float x, float y, float q00, float q01, float q10, float q11, float op,
                                           float px, float py) {
    const float e0 = x - px, e1 = y - py;
    const float d0 = -0.5f * e0, d1 = -0.5f * e1;
    const float t0 = __builtin_fmaf(d1, q10, d0 * q00);
    const float t1 = __builtin_fmaf(d1, q11, d0 * q01);
    const float power = t0 * e0 + t1 * e1;
    return __builtin_amdgcn_exp2f(power * 1.44269504088896340736f) * op;
}

// Exponent of one pixel.  px, p = the pixel's offsets inside the tile (px shared by the lane's pixels); g.mx, g.my =
// x', y' (tile-relative), g.c0 = r11 y' + h x'.  The association is fixed; every REF_CPU kernel evaluates exactly this.
__device__ __forceinline__ void exponent_x(const Splat &g, float px, float &s0, float &t0) {
    const float e_s = g.mx - px;
    if (g.kind == kKindMono) {   // monomial fallback: s0 = e_s^2 Q''00 + log2 op, t0 = e_s Qs
        s0 = __builtin_fmaf(e_s * e_s, g.d1, g.lop);
        t0 = e_s * g.h;
    } else {        // t0 = c = w at the row of the tile's origin
        s0 = __builtin_fmaf(-(g.d1 * e_s), e_s, g.lop);
        t0 = __builtin_fmaf(-g.h, px, g.c0);
    }
}
__device__ __forceinline__ float exponent_y(const Splat &g, float p, float s0, float t0) {
    if (g.kind == kKindMono) {
        const float e_p = g.my - p;
        return __builtin_fmaf(e_p, __builtin_fmaf(e_p, g.r11, t0), s0);
    }
    const float w = __builtin_fmaf(-g.r11, p, t0);
    return __builtin_fmaf(-w, w, s0);
}

// NPX pixels of one lane against one Gaussian.  The lane's pixels share x (px = their x offset inside the tile) and
// differ in y (e_p[j] = the pixel's y offset inside the tile).
// blk: the block of the tile the lane's pixels lie in -- a record the block does not keep (Splat.blocks) is not composited
// there: alpha = 0, which leaves T and the colour exactly as they are, like not visiting the record at all.
// ox, oy: the tile's origin in the frame (a kKindRefOrder record is evaluated in frame coordinates, like the reference).
// WITH_REF = false: the caller never meets a kKindRefOrder record (the first launch of the tile-16 kernel sends such
// tiles away before it gets here -- GSX_FLAG_PLAIN_FOOTPRINTS) and does not carry the code for one.
template <int NPX, bool WITH_REF = true>
__device__ __forceinline__ void composite(float px, const float (&e_p)[NPX], const Splat &g, float (&T)[NPX],
                                          float (&c0)[NPX], float (&c1)[NPX], float (&c2)[NPX], int blk, float ox, float oy) {
    float s0, t0;
    exponent_x(g, px, s0, t0);
    const bool kept = (g.blocks >> blk) & 1u;
    float ta[NPX], test[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        float alpha = (WITH_REF && g.kind == kKindRefOrder)
                          ? alpha_ref(g.mx, g.c0, g.d1, g.h, g.r11, g.my, g.lop, ox + px, oy + e_p[j])
                          : __builtin_amdgcn_exp2f(exponent_y(g, e_p[j], s0, t0));
        alpha = kept ? alpha : 0.0f;
        ta[j] = T[j] * alpha;
        test[j] = T[j] - ta[j];
    }
    float m = test[0], tmin = T[0];
#pragma unroll
    for (int j = 1; j < NPX; ++j) {
        m = fminf(m, test[j]);
        tmin = fminf(tmin, T[j]);   // fminf drops a NaN test: a stopped pixel is found through its T
    }
    if (__builtin_expect(__any((m < kStopRefCpu) | (tmin == 0.0f)), 0)) {
        // some pixel of the wave saturates here (or already has, T = 0): it must not receive
        // this Gaussian and stays at T = 0, like the reference's early return.  T == 0 is tested on
        // its own: a stopped pixel meeting a non-finite alpha would otherwise compute 0 * inf = NaN,
        // fail `NaN < threshold` and turn NaN, where the reference has already returned
        // (gaussian_scene.py:166-167; a live pixel never has T == 0, its T stays >= 1e-6).
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const bool stop = (T[j] == 0.0f) | (test[j] < kStopRefCpu);
            ta[j] = stop ? 0.0f : ta[j];
            test[j] = stop ? 0.0f : test[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        c0[j] = __builtin_fmaf(ta[j], g.cr, c0[j]);
        c1[j] = __builtin_fmaf(ta[j], g.cg, c1[j]);
        c2[j] = __builtin_fmaf(ta[j], g.cb, c2[j]);
        T[j] = test[j];
    }
}

// Gathers one record per lane into LDS, the mean made tile-relative; returns (wave-uniform) what the batch holds:
// kBatchRegular (every record has D1 >= 0, hence 0 <= alpha <= 1 and T never rises), kBatchWild (some D1 < 0: a 2D
// covariance whose float32 determinant came out negative and was floored, utils.py:383 -- the reference's exp can
// then exceed 1 -- thin footprints of a heavy-tailed scene have them in most batches) or kBatchMono (a record in
// the monomial fallback: only caller-given inverse covariances on the stage-2 entry).
//
// Records that cannot matter to ANY pixel of the tile are not staged at all.  The reference lists a Gaussian
// for every tile its bounding box touches, plus a tile of slack (`min <= x0 + T`, gaussian_scene.py:209-217), so
// ~8 % of the listed (tile, Gaussian) pairs lie more than 5.9 sigma from every pixel of their tile.  If
//     bound = log2 op - D1 min(e0^2) - min(w^2) < -26   over the tile's pixel rectangle (w = r11 e1 + h e0 is linear:
//                                                         its extremes are at the corners),
// then alpha < 2^-26 at every pixel, so T - T alpha == T bit for bit (T alpha is below half an ulp of T): skipping
// the record leaves every T identical and drops less than 2^bound of colour per channel.  How much a tile drops
// ALTOGETHER is accounted for, in scalar integer arithmetic: a skipped record is charged 2^-26, 2^-33 or 2^-40 --
// the class its bound falls into (three ballots per batch) -- to `skipped`, the tile's running total in units of
// 2^-40, and a batch skips only the classes that still fit kSkipBudget = 2^-17 = 7.6e-6 of colour per tile: all
// three, else the two farther ones, else the farthest, else none.  (Round 2 had the flat -26 alone: 6 711 skipped
// records of peak alpha 1.5e-8 each would have reached the 1e-4 pixel tolerance; tested with 16 384.  A threshold
// falling with the list length -- the first fix -- cost the heavy-tailed scene 0.11 ms although its 12 000-entry
// lists are mostly records 2^-100 away; summing the exact 2^bound over the wave -- the second -- cost the same in
// cross-lane latency, the lists being mostly skipped records there.)  The tile lists and D are untouched -- this is
// a decision of the compositing kernel, the same in every REF_CPU kernel (it depends only on the tile and its
// list, walked 64 records at a time from the start), so the kernel families stay bit-identical to each other.
// nb: in = records of the batch, out = records staged (wave-uniform).
//
// BLOCKS (round 4, see Staged): the test and the budget are per 8x8 block of a 16x16 tile.  Three ways to stage:
//   kStageWhole     the tile is one block (any tile size): a record is staged when the tile keeps it;
//   kStageBlocks    tile 16: a record is staged when ANY of the four blocks keeps it, the bits of the blocks that do ride
//                   in its c.w, and every block's list -- the slots of its records, in order -- is written to sh.list,
//                   padded with the slot of a null record (count[g] entries each; the four-pixels-per-lane kernel walks
//                   the lists, the one-pixel-per-lane kernels walk the slots and test the bits);
//   kStageOneBlock  tile 16: only what block `blk` keeps (a helper wave of a long tile composites ONE block).
// A block's decisions depend on the tile, the block and the list alone -- walked 64 entries at a time from the start --,
// so every kernel arrives at the same ones and the kernel families stay bit-identical to each other.
constexpr uint32_t kSkipBudget = 1u << 23;   // 2^-17 of colour per block and channel, in units of 2^-40 (colours are < 1)
// (kBatchRefOrder: kKindRefOrder records among regular ones; kBatchRefWild: and a D1 < 0 record as well; neither holds a monomial one)
enum { kBatchRegular = 0, kBatchWild = 1, kBatchMono = 2, kBatchRefOrder = 3, kBatchRefWild = 4, kBatchKindMask = 7, kBatchHasRef = 8 };
enum { kStageWhole = 0, kStageBlocks = 1, kStageOneBlock = 2 };

// Which of the three classes of far-away records does this batch still skip in one block?  bound: the lane's record's
// log2 of its largest alpha on the block (0 for lanes without a candidate).  Returns the threshold below which a record
// is skipped (-inf: none) and charges the skipped ones to the block's running total (wave-uniform, scalar).
__device__ __forceinline__ float skip_threshold(float bound, uint32_t &skipped, uint32_t budget) {
    const unsigned long long far1 = __ballot(bound < -26.0f), far2 = __ballot(bound < -33.0f), far3 = __ballot(bound < -40.0f);
    const uint32_t n3 = (uint32_t)__popcll(far3), n2 = (uint32_t)__popcll(far2) - n3, n1 = (uint32_t)__popcll(far1) - n2 - n3;
    const uint32_t cost3 = n3, cost23 = cost3 + (n2 << 7), cost123 = cost23 + (n1 << 14);
    if (skipped + cost123 <= budget) {
        skipped += cost123;
        return -26.0f;
    }
    if (skipped + cost23 <= budget) {
        skipped += cost23;
        return -33.0f;
    }
    if (skipped + cost3 <= budget) {
        skipped += cost3;
        return -40.0f;
    }
    return -__builtin_inff();
}

// The second half of staging: `have` lanes hold a record (a, b, c) of the batch in registers.  skipped: the running
// totals of the blocks (kStageWhole: [0], kStageOneBlock: [blk]).  count (kStageBlocks): entries of every block's list.
// WITH_REF: the caller composites reference-order records itself (qraw[gi] = the lane's raw conic, see Record -- fetched
// HERE, for the records that are still flagged after the tile's own test and kept: a heavy-tailed scene has a flagged
// Gaussian in nearly every batch, and fetched up front for all of them the side array was a dependent trip to memory per
// batch, ~3 us for a long tile's lone wave, 188 batches long): such a
// record is staged as (x, y, Q00, Q01) (Q10, op, r, g) (b, Q11, flag, bits) in FRAME coordinates; without it the record
// keeps its completed-square form -- all the caller needs is the batch's kind: it leaves the tile undone and counts it.
// Either way the skip bound is the completed square's, computed like any other record's.
template <int MODE, bool WITH_REF = true>
__device__ __forceinline__ int stage_records(float4 a, float4 b, float4 c, bool have, uint32_t &nb, Staged &sh, int lane,
                                             float tile_x0, float tile_y0, float tile_side, uint32_t (&skipped)[kBlocks],
                                             uint32_t budget, int blk, uint32_t (&count)[kBlocks], uint32_t dead = 0u,
                                             unsigned long long *ref_slots = nullptr, const float4 *__restrict__ qraw = nullptr,
                                             uint32_t gi = 0u) {
    bool irregular = false, mono = false, refo = false;
    float bound[kBlocks] = {0.0f, 0.0f, 0.0f, 0.0f};   // < -26: a candidate, alpha < 2^bound on the whole block
    float x_abs = 0.0f, y_abs = 0.0f;
    if (have) {
        refo = c.z == kRefOrderFlag;
        mono = !refo && c.z != 0.0f;
        if (WITH_REF) {
            x_abs = a.x;
            y_abs = a.y;
        }
        a.x -= tile_x0;                // tile-relative mean (see the head of this file)
        a.y -= tile_y0;
        const bool bounded = !mono && a.z >= 0.0f;
        // (a reference-order record: 0.02 covers the reference's own rounding of the exponent, < 1e-2)
        const float bx = a.x, by = a.y, bd1 = a.z, bh = a.w, br11 = b.x, blop = refo ? b.y + 0.02f : b.y;
        if (bounded) {               // (x', y', D1, h) (r11, log2 op, ..): completed square, D1 >= 0
            // log2 of the largest alpha over the pixel rectangle [x_lo, x_hi] x [y_lo, y_hi] (offsets inside the tile):
            // D1 min(e0^2) and min(w^2), w = r11 e1 + h e0 being linear in the pixel -- its extremes are at the corners
            auto over = [&](float x_lo, float x_hi, float y_lo, float y_hi) -> float {
                const float ex0 = bx - x_lo, ex1 = bx - x_hi;
                const float ey0 = by - y_lo, ey1 = by - y_hi;
                const float ex_min2 = ex0 * ex1 <= 0.0f ? 0.0f : fminf(ex0 * ex0, ex1 * ex1);
                const float w00 = __builtin_fmaf(br11, ey0, bh * ex0), w01 = __builtin_fmaf(br11, ey1, bh * ex0);
                const float w10 = __builtin_fmaf(br11, ey0, bh * ex1), w11 = __builtin_fmaf(br11, ey1, bh * ex1);
                const float wlo = fminf(fminf(w00, w01), fminf(w10, w11)), whi = fmaxf(fmaxf(w00, w01), fmaxf(w10, w11));
                const float wabs = fminf(fabsf(wlo), fabsf(whi));
                const float w_min2 = (wlo <= 0.0f && whi >= 0.0f) ? 0.0f : wabs * wabs;
                return blop - bd1 * ex_min2 - w_min2;      // NaN anywhere: every comparison below is false, the record stays
            };
            if (MODE == kStageWhole) {
                bound[0] = over(0.0f, tile_side - 1.0f, 0.0f, tile_side - 1.0f);
            } else if (MODE == kStageOneBlock) {
                bound[0] = over(8.0f * (float)(blk & 1), 8.0f * (float)(blk & 1) + 7.0f, 8.0f * (float)(blk >> 1), 8.0f * (float)(blk >> 1) + 7.0f);
            } else {
#pragma unroll
                for (int g = 0; g < kBlocks; ++g)
                    bound[g] = over(8.0f * (float)(g & 1), 8.0f * (float)(g & 1) + 7.0f, 8.0f * (float)(g >> 1), 8.0f * (float)(g >> 1) + 7.0f);
            }
            if (refo) {
                // A flagged footprint is ill-conditioned SOMEWHERE; whether the reference's rounding can show on THIS
                // tile is decided here, the same way by every kernel (tile rectangle, not block).  The reference's four
                // products sum to S = ln2 (M00 e0^2 + 2 |M01| |e0 e1| + M11 e1^2) in magnitude (M00 = D1 + h^2, M01 = h r11,
                // M11 = r11^2), its exponent is off by at most ~4 ulp(S) = 2.4e-7 S, its alpha by that times alpha: with S
                // and alpha bounded over the tile's pixels, a record that cannot move any of them by 2e-5 is composited by
                // the completed square like any other -- a needle matters this way only on the tiles along its ridge.
                const float hi = tile_side - 1.0f;
                const float ex = fmaxf(fabsf(bx), fabsf(bx - hi)), ey = fmaxf(fabsf(by), fabsf(by - hi));
                const float S = 0.69314718f * ((bd1 + bh * bh) * (ex * ex) + 2.0f * fabsf(bh) * br11 * (ex * ey) + (br11 * br11) * (ey * ey));
                const float amax = __builtin_amdgcn_exp2f(fminf(over(0.0f, hi, 0.0f, hi), 0.0f));
                // (delta: how far the reference's exponent may be off on this tile.  "alpha moves by delta alpha" holds
                // for a small delta only: beyond 0.1 -- a 3000:1 needle seen from 1000 px -- the reference's alpha is the
                // exact one times e^(+-delta), anything up to infinity, and the record stays flagged whatever the exact
                // alpha is: tools/fuzz.py extreme)
                const float delta = 2.4e-7f * S;
                if (delta < 0.1f && delta * amax < 2e-5f) {        // (NaN: stays flagged)
                    refo = false;
                    c.z = 0.0f;          // staged as the completed-square record it is on this tile
                } else {
                    // ... and a block may drop the record only where the REFERENCE's alpha is negligible: the exact
                    // bound raised by what the reference's rounding can add (in log2 units)
                    const float slack = 1.4427f * delta;
#pragma unroll
                    for (int g = 0; g < kBlocks; ++g) bound[g] += slack;       // (NaN / inf: the record is kept)
                }
            }
        }
        // irregular: the monomial fallback, a reference-order record, or D1 < 0: alpha may exceed 1 there, T is no
        // longer monotone, and the batch tests T after every record
        irregular = mono || refo || !(a.z >= 0.0f);
    }
    // who keeps the record: the classes every block still skips (skip_threshold) against the record's bound there
    uint32_t bits = 0;
    unsigned long long kept_by[kBlocks] = {0ull, 0ull, 0ull, 0ull};      // (lane order)
    if (MODE == kStageBlocks) {
#pragma unroll
        for (int g = 0; g < kBlocks; ++g) {
            if ((dead >> g) & 1u) continue;       // (wave-uniform) every pixel of the block has saturated: it keeps nothing more
            const float thr = skip_threshold(bound[g], skipped[g], budget);
            const bool k = have && !(bound[g] < thr);
            bits |= k ? (1u << g) : 0u;
            kept_by[g] = __ballot(k);
        }
    } else {
        const float thr = skip_threshold(bound[0], skipped[MODE == kStageOneBlock ? blk : 0], budget);
        bits = (have && !(bound[0] < thr)) ? 0xFu : 0u;
    }
    const bool keep = bits != 0u;
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (WITH_REF && refo && keep) q4 = qraw[gi];
    const unsigned long long mask = __ballot(keep);
    nb = (uint32_t)__popcll(mask);
    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));   // kept lanes below this one
    if (MODE == kStageBlocks) {
        // every block's list starts out as nothing but the first null record's slot (the loops take whole trips) ...
        const uint16_t null_slot = (uint16_t)(nb * 16u);
#pragma unroll
        for (int g = 0; g < kBlocks; ++g) {
            sh.list[g][lane] = null_slot;
            if (lane < kPad) sh.list[g][64 + lane] = null_slot;
        }
    }
    if (keep) {
        c.w = __uint_as_float(bits);
        if (WITH_REF && refo) {        // (x, y, Q00, Q01) (Q10, op, r, g) (b, Q11, flag, bits), frame coordinates
            a = make_float4(x_abs, y_abs, q4.x, q4.y);
            b.x = q4.z;
            b.y = c.y;
            c.y = q4.w;
        } else {
            c.y = a.y;                                             // y' (the depth is not needed here)
            if (!mono) a.y = __builtin_fmaf(b.x, a.y, a.w * a.x);  // c0 = w at the tile's origin
        }
        sh.rec[0][slot] = a;
        sh.rec[1][slot] = b;
        sh.rec[2][slot] = c;
        if (MODE == kStageBlocks) {
            // ... and receives the slots of the records the block keeps, in list order (the DS operations of one wave
            // execute in issue order: these land after the fill above)
#pragma unroll
            for (int g = 0; g < kBlocks; ++g)
                if ((bits >> g) & 1u)
                    sh.list[g][__builtin_amdgcn_mbcnt_hi((uint32_t)(kept_by[g] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kept_by[g], 0u))] = (uint16_t)(slot * 16u);
        }
    }
    if (MODE == kStageBlocks) {
#pragma unroll
        for (int g = 0; g < kBlocks; ++g) count[g] = (uint32_t)__popcll(kept_by[g]);
    }
    if (lane < kPad) {                  // the null records behind the batch (see kPad)
        sh.rec[0][nb + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        sh.rec[1][nb + lane] = make_float4(0.0f, -__builtin_inff(), 0.0f, 0.0f);
        sh.rec[2][nb + lane] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFu));
    }
    // which SLOTS hold a reference-order record (wave-uniform, scalar: a few iterations at most): the loops look at a
    // record's kind only in the trips that hold one
    unsigned long long ref_lanes = __ballot(refo && keep), in_slots = 0ull;
    while (ref_lanes) {
        const int l = __builtin_ctzll(ref_lanes);
        ref_lanes &= ref_lanes - 1ull;
        in_slots |= 1ull << __popcll(mask & ((1ull << l) - 1ull));
    }
    if (ref_slots) *ref_slots = in_slots;
    // (kBatchHasRef: the batch holds a reference-order record WHATEVER its kind -- one that also holds a monomial record is
    // kBatchMono --: what n_redo counts and what the plain instance must not composite.  A bit of the return value, not
    // a test of *ref_slots at the call site: that kept the 64-bit word live in the 64-VGPR instance and cost it 32 bytes
    // more scratch per lane, 925 -> 1 060 us on the 4K frame.)
    // (an instance that cannot evaluate reference-order records -- !WITH_REF -- is told about them FIRST: every such caller
    // leaves the tile undone on kBatchRefOrder, also when the batch holds a monomial record beside it)
    const int kind = (__any(mono && keep) && (WITH_REF || !in_slots))
                         ? kBatchMono
                         : (in_slots ? ((WITH_REF && __any(irregular && !refo && keep)) ? kBatchRefWild : kBatchRefOrder)
                                     : (__any(irregular && keep) ? kBatchWild : kBatchRegular));
    return kind | ((WITH_REF && in_slots) ? kBatchHasRef : 0);
}

// Gather + staging of one batch.  idx: this lane's entry of the tile's list (vals[base + lane]) when the caller has
// requested it ahead (a batch earlier: one of the two dependent trips to memory of a batch is then off its path).
template <int MODE, bool WITH_REF = true>
__device__ __forceinline__ int stage_batch(const Record *__restrict__ rec, const float4 *__restrict__ qraw,
                                           const uint32_t *__restrict__ vals,
                                           uint32_t base, uint32_t &nb, Staged &sh, int lane, float tile_x0,
                                           float tile_y0, float tile_side, uint32_t (&skipped)[kBlocks], uint32_t (&count)[kBlocks],
                                           uint32_t budget = kSkipBudget, const uint32_t *idx = nullptr, int blk = 0,
                                           uint32_t dead = 0u, unsigned long long *ref_slots = nullptr) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    const bool have = (uint32_t)lane < nb;
    uint32_t gi = 0u;
    if (have) {
        gi = idx ? *idx : vals[base + lane];
        const Record *q = rec + gi;
        a = q->a;
        b = q->b;
        c = q->c;
    }
    return stage_records<MODE, WITH_REF>(a, b, c, have, nb, sh, lane, tile_x0, tile_y0, tile_side, skipped, budget, blk, count, dead,
                                         ref_slots, qraw, gi);
}

// ---- packed-math form of the same arithmetic (two pixels per VGPR pair) ----------------------
// The lane's 4 pixels are kept as two float2 and every per-pixel operation except v_exp_f32 is issued
// as v_pk_{fma,mul,add}_f32.  On gfx950 a packed instruction takes the issue time of two plain ones
// (tools/microbench_valu.hip), so this saves no VALU cycles by itself; it halves the instruction
// count and, with one saturation test per two records, makes the loop 11 % faster than the scalar
// form.  Same operations, same order, same bits as composite<4>.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_exp2(v2f p) { return v2f{__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)}; }
__device__ __forceinline__ v2f splat2(float v) { return v2f{v, v}; }

// alpha of one (completed-square) record at the lane's 4 pixels (pairs a = (y0,y1), b = (y2,y3)); independent
// of T.  A = (x', c0, D1, h) as staged, r11, lop = log2(opacity factor); cx, cya, cyb = the pixels' offsets inside
// the tile.  Same operations as exponent_x / exponent_y.
__device__ __forceinline__ void alphas(float4 A, float r11, float lop, float cx, v2f cya, v2f cyb, v2f &al_a,
                                       v2f &al_b) {
    const float e_x = A.x - cx;
    const float s0 = __builtin_fmaf(-(A.z * e_x), e_x, lop);
    const float c = __builtin_fmaf(-A.w, cx, A.y);
    const v2f wa = pk_fma(splat2(-r11), cya, splat2(c)), wb = pk_fma(splat2(-r11), cyb, splat2(c));
    al_a = pk_exp2(pk_fma(-wa, wa, splat2(s0)));
    al_b = pk_exp2(pk_fma(-wb, wb, splat2(s0)));
}

// The same for a record in the monomial fallback (flag set): the operations of exponent_x / exponent_y's mono branch,
// packed.  A = (x', -, Q''00, Q''01 + Q''10), yp = y', q11 = Q''11.
__device__ __forceinline__ void alphas_mono(float4 A, float yp, float q11, float lop, float cx, v2f cya, v2f cyb, v2f &al_a,
                                            v2f &al_b) {
    const float e_x = A.x - cx;
    const float s0 = __builtin_fmaf(e_x * e_x, A.z, lop);
    const float t0 = e_x * A.w;
    const v2f ea = splat2(yp) - cya, eb = splat2(yp) - cyb;
    al_a = pk_exp2(pk_fma(ea, pk_fma(ea, splat2(q11), splat2(t0)), splat2(s0)));
    al_b = pk_exp2(pk_fma(eb, pk_fma(eb, splat2(q11), splat2(t0)), splat2(s0)));
}

// The same for a kKindRefOrder record: alpha_ref's operations, packed.  A = (x, y, Q00, Q01) in frame coordinates;
// fx = the lane's x in the frame, fya, fyb = its four y.
__device__ __forceinline__ void alphas_ref(float4 A, float q10, float q11, float op, float fx, v2f fya, v2f fyb, v2f &al_a,
                                           v2f &al_b) {
    const float e0 = A.x - fx, d0 = -0.5f * e0;
    const float p00 = d0 * A.z, p01 = d0 * A.w;
    const v2f e1a = splat2(A.y) - fya, e1b = splat2(A.y) - fyb;
    const v2f d1a = splat2(-0.5f) * e1a, d1b = splat2(-0.5f) * e1b;
    const v2f t0a = pk_fma(d1a, splat2(q10), splat2(p00)), t0b = pk_fma(d1b, splat2(q10), splat2(p00));
    const v2f t1a = pk_fma(d1a, splat2(q11), splat2(p01)), t1b = pk_fma(d1b, splat2(q11), splat2(p01));
    const v2f pa = t0a * splat2(e0) + t1a * e1a, pb = t0b * splat2(e0) + t1b * e1b;
    const float l2e = 1.44269504088896340736f;
    al_a = pk_exp2(pa * splat2(l2e)) * splat2(op);
    al_b = pk_exp2(pb * splat2(l2e)) * splat2(op);
}

#define GSX_ACCUMULATE(ta_a, ta_b, cr, cg, cb)          \
    do {                                                \
        c0a = pk_fma(ta_a, splat2(cr), c0a);            \
        c1a = pk_fma(ta_a, splat2(cg), c1a);            \
        c2a = pk_fma(ta_a, splat2(cb), c2a);            \
        c0b = pk_fma(ta_b, splat2(cr), c0b);            \
        c1b = pk_fma(ta_b, splat2(cg), c1b);            \
        c2b = pk_fma(ta_b, splat2(cb), c2b);            \
    } while (0)

// Exact saturation rule for one pixel pair (the rare path): a pixel whose T(1-alpha) drops below
// the threshold does not receive this Gaussian and stays at T = 0; a pixel that has stopped (T == 0)
// stays stopped whatever alpha is (0 * inf = NaN must not revive it, see composite<>).
__device__ __forceinline__ void checked_pair(v2f alpha, v2f &T, v2f &ta) {
    ta = T * alpha;
    v2f t = T - ta;
    const bool s0 = (T.x == 0.0f) | (t.x < kStopRefCpu), s1 = (T.y == 0.0f) | (t.y < kStopRefCpu);
    ta.x = s0 ? 0.0f : ta.x;
    ta.y = s1 ? 0.0f : ta.y;
    t.x = s0 ? 0.0f : t.x;
    t.y = s1 ? 0.0f : t.y;
    T = t;
}

__device__ __forceinline__ float min4(v2f a, v2f b) { return fminf(fminf(a.x, a.y), fminf(b.x, b.y)); }

// ---- spare workgroups of the compositing launch: the next frame's depth-sort splitters (GsxParams.hints) ----------
// kSortSamples regularly spaced kept depth keys of this frame (left by the partition's count kernel; invalid entries
// >= kEmptyKey) -> splitters[j] = the valid sample of rank floor(j V / 256), j = 1 .. 255.  kRankGroups single-wave
// workgroups; every sample's rank is counted directly -- #{s_j < s_i} + #{j < i : s_j == s_i} -- from the 8 KB table
// in the cache: no LDS, ~2 000 VALU instructions per lane, lost inside the compositing launch.  That launch is the
// frame's last: every reader of the current splitters has finished (stream order), so they are overwritten in place.
constexpr uint32_t kRankLanes = 4, kRankPerGroup = 64 / kRankLanes, kRankGroups = kSortSamples / kRankPerGroup;
__device__ __forceinline__ void rank_samples(uint32_t group, int lane, const BlendHints &h) {
    // 128 single-wave workgroups, 16 samples each, 4 lanes per sample that share the 2048 comparisons (a wave per 64
    // samples, 2048 comparisons per lane, took ~70 us -- longer than the compositing of a 100 000-Gaussian frame)
    const uint32_t *__restrict__ sm = h.samples;
    const uint32_t i = group * kRankPerGroup + ((uint32_t)lane / kRankLanes), part = (uint32_t)lane % kRankLanes;
    const uint32_t v = sm[i];
    uint32_t c = 0, valid = 0;
    for (uint32_t j0 = 0; j0 < (uint32_t)kSortSamples; j0 += 8 * kRankLanes) {
        uint32_t x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = sm[j0 + (uint32_t)e * kRankLanes + part];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t j = j0 + (uint32_t)e * kRankLanes + part;
            const bool ok = x[e] < kEmptyKey;
            valid += ok ? 1u : 0u;
            c += (ok && ((x[e] < v) | ((x[e] == v) & (j < i)))) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = kRankLanes / 2; o > 0; o >>= 1) {
        c += (uint32_t)__shfl_xor((int)c, o);
        valid += (uint32_t)__shfl_xor((int)valid, o);
    }
    if (group == 0 && lane == 0) {
        h.splitters[0] = 0u;
        h.header[kHintSplitters] = valid ? (uint32_t)kSortBins : 0u;   // nothing kept: the next frame takes its stand-in splitters
    }
    if (part == 0 && v < kEmptyKey && valid) {
        // the quantiles j with floor(j valid / 256) == c (none, one, or several when valid < 256)
        for (uint32_t j = (c * (uint32_t)kSortBins + valid - 1u) / valid; j < (uint32_t)kSortBins && (j * valid) / kSortBins == c; ++j)
            if (j) h.splitters[j] = v;
    }
}

// ---- two instances of every tile-16 loop.  REF = true (blend_tile16_ref_kernel, what a frame runs): reference-order
// records are evaluated where they turn up; 128 VGPRs, 4 waves per SIMD, nothing spilled.  REF = false
// (blend_tile16_kernel, GSX_FLAG_PLAIN_FOOTPRINTS: the caller knows from an earlier frame of the view that no tile holds
// such a record): the completed square only, 64 VGPRs, 8 waves per SIMD -- 6 % faster on a 4K frame of 5M Gaussians, no
// faster at 1080p --; a tile or long tile's quarter that does meet a flagged record stops there, writes nothing and adds
// one to *redo (GsxFrameStats.n_redo: the caller renders the frame again without the flag).  REF = true counts the tiles
// and quarters that held one there, so that a caller learns when the flag is safe.
// (Round 5 first ran REF = false for every frame with a SECOND launch behind it for the tiles it had left, fed by a list:
// one launch waited for the other's longest tile, a needle-ridden frame took 0.77 ms where this takes 0.53.)
// The barrier between staging a batch and compositing it: every workgroup is one wave, all that is needed is that the
// wave's LDS reads follow its own LDS writes -- which the hardware guarantees (the DS operations of one wave execute in
// issue order) once the compiler keeps them in program order (WAVE; the REF = false instances keep their s_barrier: a
// change there moves the register allocation of a kernel that has none to spare).
template <bool WAVE>
__device__ __forceinline__ void tile_sync() {
    if (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// ---- long tiles: a quarter of the tile -- one of its four 8x8 blocks -- per wave, one pixel per lane, eight records per trip
// Same per-(pixel, record) arithmetic as the kernels above (bit-identical frames, tested): what changes is
// the shape of the loop.  A lone wave spends ~430 cycles per record in the two-records-per-trip loop (LDS
// wait, the dependent exponent -> exp2 -> T chain); with the eight alphas of a trip computed independently
// and only the T / colour chains sequential, and four such waves per tile, a 20 000-entry tile takes about as
// long as 700 entries did.  The four workgroups of a tile are placed on one XCD (block ids congruent mod 8),
// so the records they all gather are fetched into that XCD's L2 once.
// REF (see above): false -- the quarter does not composite reference-order records itself: the first batch that holds
// one ends the workgroup, counted in *redo; true -- it does.
template <bool REF>
__device__ __forceinline__ void blend_long_tile_quarter(const Record *__restrict__ rec, const float4 *__restrict__ qraw,
                                                        const uint32_t *__restrict__ vals,
                                                        const uint2 *__restrict__ ranges, const TileGrid &g,
                                                        const OutDesc &out, uint32_t t, int quarter,
                                                        Staged &sh, uint32_t budget, uint32_t *cost_out, uint32_t *redo) {
    const int lane = REF ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;      // (REF: four independent waves per workgroup)
#ifdef GSX_TEST_HOOKS
    const unsigned long long probe_t0 = __builtin_readcyclecounter();
    const uint32_t probe_w0h = (uint32_t)wall_clock64();
    uint32_t probe_staged = 0, probe_checked_at = 0xFFFFFFu, probe_batch = 0;
    uint32_t probe_qc[3] = {0u, 0u, 0u}, probe_qtrips = 0, probe_qref = 0;      // cycles: staging, plain trips, trips with a ref-order record
#endif
    const int tx = g.wx0 + (int)(t / (uint32_t)g.nwy()), ty = g.wy0 + (int)(t % (uint32_t)g.nwy());
    const bool y_contig = out.stride_y < out.stride_x;
    // a quarter = block `quarter` of the tile (see Staged): 8 x 8 pixels; WH3: lanes run along y (96 contiguous bytes per
    // column), HW3: along x
    const int lx = 8 * (quarter & 1) + (y_contig ? (lane >> 3) : (lane & 7));      // the pixel's offset inside the tile
    const int ly = 8 * (quarter >> 1) + (y_contig ? (lane & 7) : (lane >> 3));
    const int px = tx * 16 + lx, py = ty * 16 + ly;
    const float cx = (float)lx, cy = (float)ly;
    float T = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    bool checked = false;   // wave-uniform: some pixel of this quarter has saturated
    bool saw_ref = false;   // (REF) wave-uniform: a batch held a reference-order record
    uint2 rg = ranges[t];
    rg.y &= ~kLongFlag;
    uint32_t skipped[kBlocks] = {0u, 0u, 0u, 0u}, unused[kBlocks];   // colour this block has left out so far (stage_records)
    uint32_t cost = 0;               // what this quarter walked (GsxParams.hints: see the tile kernel)
    constexpr int kTrip = 8;
    // The gather runs AHEAD of the compositing: while batch i is composited, the records of batch i + 1 and the
    // list entries of batch i + 2 are in flight.  A long tile's wave is nearly alone on its SIMD at the end of the
    // frame, nothing hides its two dependent trips to memory per batch, and with most records of a heavy-tailed
    // list not staged at all those trips ARE the tile's duration (188 batches x ~2.5 us on the clustered scene; the
    // compiler overlapped them in one build and not in the next, depending on its register allocation -- round 3).
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb4 = na, nc = na;      // records of the next batch
    uint32_t idx2 = 0, idx1 = 0;                                            // list entry of the batch after it, of the next batch
    if (rg.x + (uint32_t)lane < rg.y) {
        idx1 = vals[rg.x + lane];
        const Record *q = rec + idx1;
        na = q->a; nb4 = q->b; nc = q->c;
    }
    if (rg.x + 64u + (uint32_t)lane < rg.y) idx2 = vals[rg.x + 64u + lane];
    for (uint32_t base = rg.x; base < rg.y; base += 64) {
        uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, rg.y - base));
#ifdef GSX_TEST_HOOKS
        const unsigned long long probe_q0 = __builtin_readcyclecounter();
#endif
        const float4 ra = na, rb = nb4, rc = nc;
        const uint32_t idx0 = idx1;       // (this batch's list entry: the Gaussian's slot of the side array)
        if (base + 64u + (uint32_t)lane < rg.y) {
            const Record *q = rec + idx2;
            na = q->a; nb4 = q->b; nc = q->c;
            idx1 = idx2;
        }
        if (base + 128u + (uint32_t)lane < rg.y) idx2 = vals[base + 128u + lane];
        unsigned long long ref_slots = 0ull;
        const int kind_all = stage_records<kStageOneBlock, REF>(ra, rb, rc, (uint32_t)lane < nb, nb, sh, lane, (float)(tx * 16),
                                                                (float)(ty * 16), 16.0f, skipped, budget, quarter, unused, 0u, &ref_slots, qraw, idx0);
        const int kind = kind_all & kBatchKindMask;
        cost += nb + kBatchCost;
#ifdef GSX_TEST_HOOKS
        probe_staged += nb;
        if (checked && probe_checked_at == 0xFFFFFFu) probe_checked_at = probe_batch;
        ++probe_batch;
#endif
        // (kBatchHasRef, not the batch's kind: a batch that also holds a monomial record is kBatchMono whatever else it holds)
        if (REF && (kind_all & kBatchHasRef)) saw_ref = true;
        // (the plain instance also leaves a batch with a MONOMIAL record -- stage-2 entry, degenerate conics -- to the other one:
        // such a batch may hold a reference-order record as well, and its kind does not say)
        if (!REF && kind >= kBatchRefOrder) {       // (wave-uniform) not here: the quarter stays undone (GSX_FLAG_PLAIN_FOOTPRINTS)
            if (lane == 0) atomicAdd(redo, 1u);
            return;
        }
        tile_sync<REF>();
#ifdef GSX_TEST_HOOKS
        unsigned long long probe_q1 = __builtin_readcyclecounter();
        probe_qc[0] += (uint32_t)(probe_q1 - probe_q0);
#endif
        // Whole trips of eight records (the batch is padded with null records, see kPad).  The alphas of a trip
        // are computed independently of each other and of T -- no branch between them --; then either the plain
        // chain T -> T - T alpha with ONE wave-level saturation test per trip, or -- from the first trip in which any
        // pixel of the quarter saturates, which is then done again -- the reference's exact rule per pixel and record
        // with selects (a stopping pixel does not receive the record and stays at T = 0; T == 0 is tested on its
        // own, see composite<>).  Where nothing stops the selects change nothing: same bits.  (Round 2 sent what was
        // left of a batch, and every record of a saturated tile, through a one-record-at-a-time loop: on a
        // heavy-tailed scene, where half the records of a long list are not staged and the dense tiles saturate early,
        // that loop -- ~420 cycles per record for a wave alone on its SIMD -- was the frame's duration.)
        if (kind == kBatchMono) {       // a record in the monomial fallback (stage-2 entry only): one at a time, exact rule
            for (uint32_t k = 0; k < nb; ++k) {
                const Splat s = read_splat(sh, k);
                const float e_p[1] = {cy};
                float T1[1] = {T}, a0[1] = {c0}, a1[1] = {c1}, a2[1] = {c2};
                composite<1, REF>(cx, e_p, s, T1, a0, a1, a2, 0, (float)(tx * 16), (float)(ty * 16));
                T = T1[0]; c0 = a0[0]; c1 = a1[0]; c2 = a2[0];
            }
        } else {
            for (uint32_t k = 0; k < nb; k += kTrip) {
                float alpha[kTrip];
#pragma unroll
                for (int u = 0; u < kTrip; ++u) {
                    const float4 A = sh.rec[0][k + u];
                    const float2 Bq = *reinterpret_cast<const float2 *>(&sh.rec[1][k + u]);   // (r11, log2 op)
                    // (wave-uniform, a scalar branch few trips take) a reference-order record: (x, y, Q00, Q01) (Q10, op) (.., Q11),
                    // the reference's operations at the pixel's FRAME coordinates.  Its alpha may exceed 1 by a rounding:
                    // the chain below tests T after every record.  (Round 5 sent such a trip through the scalar form one
                    // record at a time: 4 300 cycles against 1 400.)
                    if (REF && ((uint32_t)(ref_slots >> k) >> u) & 1u) {
                        alpha[u] = alpha_ref(A.x, A.y, A.z, A.w, Bq.x, sh.rec[2][k + u].y, Bq.y, (float)(tx * 16) + cx, (float)(ty * 16) + cy);
                        continue;
                    }
                    const float e_x = A.x - cx;
                    const float s0 = __builtin_fmaf(-(A.z * e_x), e_x, Bq.y);
                    const float w = __builtin_fmaf(-Bq.x, cy, __builtin_fmaf(-A.w, cx, A.y));
                    alpha[u] = __builtin_amdgcn_exp2f(__builtin_fmaf(-w, w, s0));
                }
                float ta[kTrip], Tt = T;
                if (!checked) {
                    float m = T;
#pragma unroll
                    for (int u = 0; u < kTrip; ++u) {
                        ta[u] = Tt * alpha[u];
                        Tt = Tt - ta[u];
                        m = fminf(m, Tt);       // (a wild batch -- alpha > 1 possible -- needs every T tested)
                    }
                    // (a NaN hides from fminf; `!(m >= ..)` sends such a trip to the exact rule as well)
                    if (__builtin_expect(__any(!(m >= kStopRefCpu)), 0)) {
                        checked = true;
                        Tt = T;
                    }
                }
                if (checked) {
#pragma unroll
                    for (int u = 0; u < kTrip; ++u) {
                        const float t_a = Tt * alpha[u], t = Tt - t_a;
                        const bool stop = (Tt == 0.0f) | (t < kStopRefCpu);
                        ta[u] = stop ? 0.0f : t_a;
                        Tt = stop ? 0.0f : t;
                    }
                }
#pragma unroll
                for (int u = 0; u < kTrip; ++u) {
                    const float2 rg_ = *reinterpret_cast<const float2 *>(&sh.rec[1][k + u].z);    // (r, g)
                    const float cb = sh.rec[2][k + u].x;
                    c0 = __builtin_fmaf(ta[u], rg_.x, c0);
                    c1 = __builtin_fmaf(ta[u], rg_.y, c1);
                    c2 = __builtin_fmaf(ta[u], cb, c2);
                }
                T = Tt;
#ifdef GSX_TEST_HOOKS
                {
                    const unsigned long long now = __builtin_readcyclecounter();
                    probe_qc[1] += (uint32_t)(now - probe_q1);
                    probe_q1 = now;
                    ++probe_qtrips;
                }
#endif
            }
        }
        tile_sync<REF>();
        if (__ballot(T > 0.0f) == 0ull) break;
    }
    // (the largest of the four quarters' costs = the records one wave would have walked until all 256 pixels are done)
    if (cost_out && lane == 0) atomicMax(cost_out, cost | 0x80000000u);
    if (REF && saw_ref && lane == 0 && redo) atomicAdd(redo, 1u);      // (GsxFrameStats.n_redo)
#ifdef GSX_TEST_HOOKS
    if (g_blend_probe && lane == 0)
        g_blend_probe[blockIdx.x] = make_uint4((uint32_t)(__builtin_readcyclecounter() - probe_t0), t | 0x40000000u, rg.y - rg.x,
                                               probe_staged | (checked ? 0x80000000u : 0u));
    if (g_blend_probe && lane == 2 && blockDim.x == 64u)
        g_blend_probe[3 * kProbeSecond + 65536u + blockIdx.x] = make_uint4(probe_qc[0], probe_qc[1], probe_qc[2], probe_qref | (probe_qtrips << 12));
    if (g_blend_probe && lane == 1)
        g_blend_probe[kProbeSecond + blockIdx.x] =
            make_uint4(probe_checked_at, (uint32_t)wall_clock64(),
                       (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16), probe_w0h);
#endif
    float *o = out.ptr + (int64_t)(px - out.x0) * out.stride_x + (int64_t)(py - out.y0) * out.stride_y;
    o[0] = c0;
    o[1] = c1;
    o[2] = c2;
}

// One 16x16 tile on one wave, 4 pixels per lane (the body of blend_tile16_kernel; see there).  REF as in
// blend_long_tile_quarter: false -- a batch that holds a reference-order record ends the tile, counted in *lt.redo.
template <int VARIANT, bool REF>
__device__ __forceinline__ void blend_tile16(const Record *__restrict__ rec, const float4 *__restrict__ qraw,
                                             const uint32_t *__restrict__ vals,
                                             const uint2 *__restrict__ ranges, const TileGrid &g, const OutDesc &out,
                                             const LongTiles &lt, uint32_t budget, const BlendHints &hints, const TileSpan &span,
                                             uint32_t t, Staged &sh) {
    const int lane = REF ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;      // (REF: four independent waves per workgroup)
#ifdef GSX_TEST_HOOKS
    const unsigned long long probe_t0 = __builtin_readcyclecounter();
    const uint32_t probe_w0 = (uint32_t)wall_clock64();
    uint32_t probe_staged = 0, probe_batches = 0, probe_after = 0;
    uint32_t probe_first_ref = 0xFFFu, probe_ref_batches = 0, probe_ref_records = 0;      // (REF: tools/attic/ref_probe.py)
    uint32_t probe_cyc[4] = {0u, 0u, 0u, 0u}, probe_ent[3] = {0u, 0u, 0u};      // cycles: staging, compositing regular / wild / ref-order batches; entries walked
#endif
    const int tx = g.wx0 + (int)(t / (uint32_t)g.nwy()), ty = g.wy0 + (int)(t % (uint32_t)g.nwy());
    if ((span.axis ? ty : tx) < span.lo || (span.axis ? ty : tx) >= span.hi) return;      // another part's tile
    // WH3: lanes 4q..4q+3 cover one x (contiguous 192 B); HW3: lanes 16q..16q+15 cover one y-quad
    const bool y_contig = out.stride_y < out.stride_x;
    const int lx = y_contig ? (lane >> 2) : (lane & 15);            // the lane's pixels: offsets inside the tile
    const int ly0 = 4 * (y_contig ? (lane & 3) : (lane >> 4));
    const int px = tx * 16 + lx, py0 = ty * 16 + ly0;
    const float cx = (float)lx;  // pixel offsets as floats (exact)
    float cy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cy[j] = (float)(ly0 + j);
    const v2f cya = v2f{cy[0], cy[1]}, cyb = v2f{cy[2], cy[3]};
    // the block of the tile the lane's four pixels lie in (see Staged): the lane walks THAT block's list
    const int blk = (lx >> 3) | ((ly0 >> 3) << 1);
    const uint16_t *my_list = sh.list[blk];
    const char *rec_a = reinterpret_cast<const char *>(&sh.rec[0][0]), *rec_b = reinterpret_cast<const char *>(&sh.rec[1][0]),
               *rec_c = reinterpret_cast<const char *>(&sh.rec[2][0]);

    float T[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    float c0[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
    v2f Ta = splat2(1.0f), Tb = splat2(1.0f);  // packed state: pairs (y0,y1) and (y2,y3)
    v2f c0a = splat2(0.0f), c1a = c0a, c2a = c0a, c0b = c0a, c1b = c0a, c2b = c0a;

    bool checked = false;  // wave-uniform: some pixel of this tile has saturated
    bool restart_scalar = false;     // wave-uniform: a monomial record turned up, the tile starts over on the scalar form
    uint2 rg = ranges[t];            // the same in every lane: kept in scalar registers
    rg.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)rg.x);
    rg.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)rg.y);
    // What the next frame's schedule and its choice of long tiles are made from (GsxParams.hints): the tile's COST --
    // records actually staged until the tile was done (a dense tile that saturates after a tenth of its list costs a
    // tenth; records that cannot matter are not staged) plus a few per batch for the staging itself -- stored when the
    // tile ends.  (A long tile's entry is the helpers': tile_ranges_kernel flagged it, they raise it.)
    if (hints.lens && lane == 0 && t == 0) {
        const bool by_cost = hints.header[kHintLens] == (uint32_t)g.count() && hints.header[kHintSched] == (uint32_t)g.count();
        hints.header[kHintLens] = (uint32_t)g.count();
        // How many tiles qualified as long this frame steers the threshold of the next (gsx_plan.h: kHintLongPct).  A
        // split tile is staged four times over, so splitting hundreds costs more than it saves (512 split tiles of the
        // heavy-tailed test scene: 0.60 ms of compositing against 0.38 for none and 0.31 for the 176 most expensive).
        if (lt.max && by_cost) {       // (a frame that chose by list length says nothing about the threshold)
            const uint32_t found = *lt.count, pct = max(hints.header[kHintLongPct], lt.cost_pct);
            hints.header[kHintLongPct] = found > 192u ? min(pct + pct / 4u, 1000u) : (found < 64u ? max(pct - pct / 8u, lt.cost_pct) : pct);
        }
    }
    if (rg.y & kLongFlag) return;   // a long tile: four helper workgroups composite it
    bool saw_ref = false;           // (REF) wave-uniform: a batch of this tile held a reference-order record
    uint32_t cost = 0;
    uint32_t skipped[kBlocks] = {0u, 0u, 0u, 0u};    // colour every block has left out so far (stage_records)
    // A block whose 64 pixels have all saturated keeps nothing more (stage_records): a dead pixel has T = 0, adds 0 and
    // stays 0, so the records it no longer sees change nothing -- the wave walks on for the blocks that are still alive
    // only (a tile over the edge of an opaque object: its covered half is done after a fraction of the list).
    unsigned long long lanes_of[kBlocks];
#pragma unroll
    for (int gb = 0; gb < kBlocks; ++gb) lanes_of[gb] = __ballot(blk == gb);
    uint32_t dead = 0u;              // wave-uniform: bit g = block g is done
    // the list entries of the next batch are requested while this one is composited (one register): one of the two
    // dependent trips to memory per batch leaves the path of a wave that has its SIMD to itself
    uint32_t idx = rg.x + (uint32_t)lane < rg.y ? vals[rg.x + lane] : 0u;
    for (uint32_t base = rg.x; base < rg.y; base += 64) {
        uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, rg.y - base));
        const uint32_t idx_now = idx;
        if (base + 64u + (uint32_t)lane < rg.y) idx = vals[base + 64u + lane];
        uint32_t count[kBlocks];
        unsigned long long ref_slots = 0ull;     // (wave-uniform) the slots of this batch that hold a reference-order record
#ifdef GSX_TEST_HOOKS
        const unsigned long long probe_b0 = __builtin_readcyclecounter();
#endif
        const int kind_all = stage_batch<kStageBlocks, (REF || VARIANT == 0)>(rec, qraw, vals, base, nb, sh, lane, (float)(tx * 16),
                                                                              (float)(ty * 16), 16.0f, skipped, count, budget, &idx_now, 0,
                                                                              dead, &ref_slots);
        const int kind = kind_all & kBatchKindMask;
        const bool wild = kind != kBatchRegular;     // wave-uniform
        // (kBatchHasRef, not the batch's kind: a batch that also holds a monomial record is kBatchMono whatever else it holds)
        if (REF && (kind_all & kBatchHasRef)) saw_ref = true;
#ifdef GSX_TEST_HOOKS
        if (REF && kind >= kBatchRefOrder) {
            if (probe_first_ref == 0xFFFu) probe_first_ref = probe_batches;
            ++probe_ref_batches;
            probe_ref_records += (uint32_t)__popcll(ref_slots);
        }
#endif
        // the wave walks as far as its LONGEST block list; the other blocks' lists are padded with a null record
        const uint32_t nl = max(max(count[0], count[1]), max(count[2], count[3]));
        cost += nl + kBatchCost;
#ifdef GSX_TEST_HOOKS
        probe_staged += nl;
        probe_batches = min(0xFFFu, probe_batches + 1u);
        if (checked) probe_after += nl;           // (entries walked under the exact rule, whole batches)
#endif
        tile_sync<REF>();
#ifdef GSX_TEST_HOOKS
        const unsigned long long probe_b1 = __builtin_readcyclecounter();
        probe_cyc[0] += (uint32_t)(probe_b1 - probe_b0);
#endif
        if (VARIANT == 0) {
            for (uint32_t k = 0; k < nb; ++k) {
                const Splat s = read_splat(sh, k);
                composite<4, (REF || VARIANT == 0)>(cx, cy, s, T, c0, c1, c2, blk, (float)(tx * 16), (float)(ty * 16));
            }
        } else if (!REF && kind >= kBatchRefOrder) {
            // (wave-uniform) an ill-conditioned footprint -- or a record in the monomial fallback (stage-2 entry, degenerate
            // conics), whose batch may hold one without its kind saying so: not here -- the tile stays undone
            // (GSX_FLAG_PLAIN_FOOTPRINTS; the instance that evaluates both composites it when the frame is rendered again)
            if (lane == 0) atomicAdd(lt.redo, 1u);
            return;
        } else if (kind == kBatchMono) {
            // A record in the monomial fallback (caller-given inverse covariances on the stage-2 entry; a degenerate
            // footprint): the packed loops cannot evaluate it, and written on the packed state that rare path cost
            // the kernel 14 registers.  The tile starts over on the scalar form instead (below): the packed state is
            // dead from here, so the two forms do not add up in the register file.
            restart_scalar = true;
            break;
        } else {
            // Common path: whole trips of four list entries (the lists are padded with a null record's slot, see
            // stage_records), ONE wave-level saturation test per trip, no per-pixel selects: twice the independent work
            // of two records between two tests hides more of the wave's own LDS reads and exponentials -- 264 -> 254 us
            // at 1M Gaussians in round 2, more where a SIMD holds fewer than 8 waves (a rank's strip).  From the first
            // trip in which any pixel of the tile saturates -- that trip is done again -- the reference's exact rule
            // per pixel and record (checked_pair) on trips of two; a saturated pixel has T = 0, adds 0 and stays 0.
            // Same operations per pixel and record in both forms, same order: same bits.
            // Entry k of the lane's list names the slot of ITS block's k-th record: the four blocks read four different
            // records per instruction (four LDS addresses; a 16-lane group shares one).
            constexpr int kTrip = VARIANT == 2 ? 6 : 4;
            uint32_t k = 0;
            // one trip of N list entries from k; false: some pixel saturates inside it (nothing is committed)
            auto trip = [&](auto n_tag) __attribute__((always_inline)) -> bool {
                constexpr int N = decltype(n_tag)::value;
                // geometry now, colour after the test: what stays live across the saturation test is T alpha of
                // the trip (16 registers) and the slots (N), not the records (36) -- the colours are read again from
                // LDS, which keeps the kernel at 64 VGPRs without spills
                v2f ta_a[N], ta_b[N], ta = Ta, tb = Tb;
                uint32_t slot[N];
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    slot[u] = my_list[k + u];                  // (byte offset of the record's slot)
                    const float4 A = *reinterpret_cast<const float4 *>(rec_a + slot[u]);
                    const float2 Bq = *reinterpret_cast<const float2 *>(rec_b + slot[u]);   // (r11, log2 op)
                    v2f aa, ab;
                    alphas(A, Bq.x, Bq.y, cx, cya, cyb, aa, ab);
                    ta_a[u] = ta * aa;
                    ta_b[u] = tb * ab;
                    ta = ta - ta_a[u];
                    tb = tb - ta_b[u];
                }
                // a regular batch has 0 <= alpha <= 1 (D1 >= 0, log2 op <= 0): T never rises inside the trip, its
                // last value is its smallest -- ONE test per trip (round 2 folded a min per record).  (A NaN hides
                // from fminf; `!(m >= ..)` sends such a trip to the exact rule as well.)
                if (__builtin_expect(__any(!(min4(ta, tb) >= kStopRefCpu)), 0)) return false;
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    const float2 rg_ = *reinterpret_cast<const float2 *>(rec_b + slot[u] + 8);    // (r, g)
                    const float cb = *reinterpret_cast<const float *>(rec_c + slot[u]);
                    GSX_ACCUMULATE(ta_a[u], ta_b[u], rg_.x, rg_.y, cb);
                }
                Ta = ta;
                Tb = tb;
                k += N;
                return true;
            };
            // the exact rule, two list entries from k (their alphas are independent)
            auto exact_pair = [&]() __attribute__((always_inline)) {
                const uint32_t s0_ = my_list[k], s1_ = my_list[k + 1];
                const float4 A0 = *reinterpret_cast<const float4 *>(rec_a + s0_), B0 = *reinterpret_cast<const float4 *>(rec_b + s0_);
                const float4 A1 = *reinterpret_cast<const float4 *>(rec_a + s1_), B1 = *reinterpret_cast<const float4 *>(rec_b + s1_);
                const float cb0 = *reinterpret_cast<const float *>(rec_c + s0_), cb1 = *reinterpret_cast<const float *>(rec_c + s1_);
                v2f aa0, ab0, aa1, ab1, ta0a, ta0b, ta1a, ta1b;
                alphas(A0, B0.x, B0.y, cx, cya, cyb, aa0, ab0);
                alphas(A1, B1.x, B1.y, cx, cya, cyb, aa1, ab1);
                checked_pair(aa0, Ta, ta0a);
                checked_pair(ab0, Tb, ta0b);
                GSX_ACCUMULATE(ta0a, ta0b, B0.z, B0.w, cb0);
                checked_pair(aa1, Ta, ta1a);
                checked_pair(ab1, Tb, ta1b);
                GSX_ACCUMULATE(ta1a, ta1b, B1.z, B1.w, cb1);
                k += 2;
            };
            // (a wild batch -- alpha > 1 possible, T may rise again inside a trip -- goes straight to the exact rule, for
            // this batch only: that rule is valid for any batch, it just costs a few selects per record.  Whole trips
            // that only a pixel saturating INSIDE them stops, tried again after the first saturation, lost: once the
            // first pixel of a tile is done the others follow record by record, nearly every trip met one and was
            // done twice -- trained-like scene 206 -> 231 us, round 4.)
            if (!wild) {
                while (!checked && k + kTrip <= nl) checked = !trip(std::integral_constant<int, kTrip>());
                while (!checked && k < nl) checked = !trip(std::integral_constant<int, 2>());   // what is left: trips of two (+ a null record)
            }
            if (!REF || kind < kBatchRefOrder) {
                while (k < nl) exact_pair();
            } else {
                // The batch holds an ill-conditioned footprint (kKindRefOrder; such a batch counts as wild).  A pair of
                // list entries none of which -- in any of the four blocks -- is such a record goes through exact_pair
                // like everywhere else; a pair that holds one evaluates it by the reference's operations (alphas_ref;
                // a 16-lane group at a time: the blocks read different records) and the other by the completed square:
                // every record's alpha is what it is in any other batch.
                const float fx_ = (float)(tx * 16) + cx;
                const v2f fya = splat2((float)(ty * 16)) + cya, fyb = splat2((float)(ty * 16)) + cyb;
                auto is_ref = [&](uint32_t s_) -> bool { return (s_ >> 4) < 64u && ((ref_slots >> (s_ >> 4)) & 1ull) != 0ull; };
                while (k < nl) {
                    // Nothing but regular records beside the flagged ones (kBatchRefOrder), no pixel saturated yet: four list
                    // entries none of which -- in any block -- is flagged are a trip like in any regular batch (one flagged
                    // record in 64 sent the whole batch through the exact rule pair by pair: 726 cycles per entry against
                    // 466, a third of the heavy-tailed scene's compositing)
                    if (kind == kBatchRefOrder && !checked && k + kTrip <= nl) {
                        bool flagged = false;
#pragma unroll
                        for (int u = 0; u < kTrip; ++u) flagged |= is_ref(my_list[k + u]);
                        if (!__any(flagged)) {
                            checked = !trip(std::integral_constant<int, kTrip>());
                            continue;
                        }
                    }
                    const uint32_t s0_ = my_list[k], s1_ = my_list[k + 1];
                    // (a list's padding names the null record's slot, which may be slot 64: no bit of the mask)
                    const bool r0 = is_ref(s0_), r1 = is_ref(s1_);
                    if (!__any(r0 | r1)) {
                        exact_pair();
                        continue;
                    }
                    // ONE entry on this path (then the pair test again from the next): what is live here comes on top of
                    // the packed state, and the kernel has no register to spare
                    const float4 A0 = *reinterpret_cast<const float4 *>(rec_a + s0_), B0 = *reinterpret_cast<const float4 *>(rec_b + s0_);
                    const float2 C0 = *reinterpret_cast<const float2 *>(rec_c + s0_);     // (b, Q11 | y')
                    v2f aa, ab, ta_a, ta_b;
                    if (r0)
                        alphas_ref(A0, B0.x, C0.y, B0.y, fx_, fya, fyb, aa, ab);
                    else
                        alphas(A0, B0.x, B0.y, cx, cya, cyb, aa, ab);
                    checked_pair(aa, Ta, ta_a);
                    checked_pair(ab, Tb, ta_b);
                    GSX_ACCUMULATE(ta_a, ta_b, B0.z, B0.w, C0.x);
                    k += 1;
                }
            }
        }
        tile_sync<REF>();
#ifdef GSX_TEST_HOOKS
        {
            const int pk = kind == kBatchRegular ? 0 : (kind >= kBatchRefOrder ? 2 : 1);
            probe_cyc[1 + pk] += (uint32_t)(__builtin_readcyclecounter() - probe_b1);
            probe_ent[pk] += nl;
        }
#endif
        bool live;
        if (VARIANT == 0)
            live = (T[0] > 0.0f) | (T[1] > 0.0f) | (T[2] > 0.0f) | (T[3] > 0.0f);
        else
            live = (Ta.x > 0.0f) | (Ta.y > 0.0f) | (Tb.x > 0.0f) | (Tb.y > 0.0f);
        const unsigned long long alive = __ballot(live);
        if (alive == 0ull) break;
        if (checked || VARIANT == 0) {
#pragma unroll
            for (int gb = 0; gb < kBlocks; ++gb) dead |= (alive & lanes_of[gb]) == 0ull ? (1u << gb) : 0u;
        }
    }
    if (VARIANT != 0 && restart_scalar) {
        // the whole tile again, one record at a time on the scalar form (composite<4>: same bits as the packed loops)
        tile_sync<REF>();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T[j] = 1.0f;
            c0[j] = c1[j] = c2[j] = 0.0f;
        }
#pragma unroll
        for (int gb = 0; gb < kBlocks; ++gb) skipped[gb] = 0u;
        for (uint32_t base = rg.x; base < rg.y; base += 64) {
            uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, rg.y - base));
            uint32_t count[kBlocks];
            const int later = stage_batch<kStageBlocks, (REF || VARIANT == 0)>(rec, qraw, vals, base, nb, sh, lane, (float)(tx * 16),
                                                                               (float)(ty * 16), 16.0f, skipped, count, budget);
            if (REF && (later & kBatchHasRef)) saw_ref = true;
            if (!REF && (later & kBatchKindMask) >= kBatchRefOrder) {   // (wave-uniform) a later batch holds one: the tile stays undone
                if (lane == 0) atomicAdd(lt.redo, 1u);
                return;
            }
            cost += 4u * nb;
            tile_sync<REF>();
            for (uint32_t k = 0; k < nb; ++k) {         // every staged record, in order; a lane's block takes what it keeps
                const Splat s = read_splat(sh, k);
                composite<4, (REF || VARIANT == 0)>(cx, cy, s, T, c0, c1, c2, blk, (float)(tx * 16), (float)(ty * 16));
            }
            tile_sync<REF>();
            if (__ballot((T[0] > 0.0f) | (T[1] > 0.0f) | (T[2] > 0.0f) | (T[3] > 0.0f)) == 0ull) break;
        }
    } else if (VARIANT != 0) {
        c0[0] = c0a.x; c0[1] = c0a.y; c0[2] = c0b.x; c0[3] = c0b.y;
        c1[0] = c1a.x; c1[1] = c1a.y; c1[2] = c1b.x; c1[3] = c1b.y;
        c2[0] = c2a.x; c2[1] = c2a.y; c2[2] = c2b.x; c2[3] = c2b.y;
    }

    if (hints.lens && lane == 0) hints.lens[t] = cost;
    if (REF && saw_ref && lane == 0 && lt.redo) atomicAdd(lt.redo, 1u);        // (GsxFrameStats.n_redo)
#ifdef GSX_TEST_HOOKS
    // REF (several waves per workgroup): by tile, behind the two records of the first launch's workgroups
    if (REF && g_blend_probe && lane == 0)
        g_blend_probe[2 * kProbeSecond + t] = make_uint4((uint32_t)(__builtin_readcyclecounter() - probe_t0),
                                                         probe_batches | (probe_first_ref << 12) | (checked ? 0x80000000u : 0u),
                                                         probe_ref_batches | (probe_ref_records << 12), rg.y - rg.x);
    if (REF && g_blend_probe && lane == 1) {
        g_blend_probe[2 * kProbeSecond + 65536u + t] = make_uint4(probe_cyc[0], probe_cyc[1], probe_cyc[2], probe_cyc[3]);
        g_blend_probe[3 * kProbeSecond + t] = make_uint4(probe_ent[0], probe_ent[1], probe_ent[2], 0u);
    }
    if ((!REF || blockDim.x == 64u) && g_blend_probe && lane == 0)
        g_blend_probe[blockIdx.x] = make_uint4((uint32_t)(__builtin_readcyclecounter() - probe_t0), t, rg.y - rg.x,
                                               probe_staged | (checked ? 0x80000000u : 0u));
    // (where and when: HW_ID = register 4, XCC_ID = register 20; wall_clock64 ticks at 100 MHz on every XCD alike)
    if ((!REF || blockDim.x == 64u) && g_blend_probe && lane == 1)
        g_blend_probe[kProbeSecond + blockIdx.x] =
            make_uint4(probe_batches | (probe_after << 12), (uint32_t)wall_clock64(),
                       (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16),
                       probe_w0);
#endif
    float *o = out.ptr + (int64_t)(px - out.x0) * out.stride_x + (int64_t)(py0 - out.y0) * out.stride_y;
    if (y_contig && (reinterpret_cast<uintptr_t>(o) & 15u) == 0) {
        float4 *o4 = reinterpret_cast<float4 *>(o);
        o4[0] = make_float4(c0[0], c1[0], c2[0], c0[1]);
        o4[1] = make_float4(c1[1], c2[1], c0[2], c1[2]);
        o4[2] = make_float4(c2[2], c0[3], c1[3], c2[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float *oj = o + (int64_t)j * out.stride_y;
            oj[0] = c0[j];
            oj[1] = c1[j];
            oj[2] = c2[j];
        }
    }
}

// Fast path, tile = 16: one wave per tile, 4 pixels per lane.  A lane owns pixels
// (x, y..y+3): x is the coordinate its pixels share, so the x-only terms of the exponent are
// computed once per record.  The assignment (and therefore every bit of the result) is the same
// for both output layouts; only the store addressing differs:
//   GSX_LAYOUT_WH3  out[x][y][c]: the lane's 4 pixels are 48 contiguous bytes (3 x dwordx4);
//   GSX_LAYOUT_HW3  out[y][x][c]: 4 stores of 12 B; the 16 lanes that share a y write 192
//                   contiguous bytes per store instruction.
// VARIANT 0: scalar-form composite<4>; VARIANT 1: packed form, four (then two) records per saturation test;
// VARIANT 2 (test library only): six per test.
template <int VARIANT, bool REF>
__device__ __forceinline__ void blend_tile16_grid(const Record *__restrict__ rec, const float4 *__restrict__ qraw,
                                                  const uint32_t *__restrict__ vals, const uint2 *__restrict__ ranges,
                                                  const TileGrid &g, const OutDesc &out, const ClearPlan &cp, const LongTiles &lt,
                                                  uint32_t nhelpers, const uint32_t *__restrict__ sched, uint32_t budget,
                                                  uint32_t quarters, const BlendHints &hints, uint32_t tile_blocks,
                                                  uint32_t sched_cap_, const TileSpan &span, Staged &sh) {
    // block order: [spare workgroups: the next frame's splitters] [helpers of long tiles (dispatched first: they have
    // the most to do)] [tiles] [clears]
    // (hints.rank_last -- gsx_api.hip: a window whose tiles just about fill the chip once -- : the spare workgroups come
    // last in the grid instead)
    const uint32_t nrank = hints.samples ? kRankGroups : 0u;
    const uint32_t rank0 = hints.rank_last ? gridDim.x - nrank : 0u;
    // span (GsxParams.n_substrips): this launch composites only the tiles whose column (axis 0) / row (axis 1) lies in
    // [lo, hi) -- one part of the window; the launch of the first part also runs what a frame does once (the next
    // frame's splitters, the zero fill of what no tile covers).  The whole grid is launched every time: a workgroup of
    // another part exits at once (~2 us per launch for the 8 000 of a 1080p frame).
    if (blockIdx.x - rank0 < nrank) {
        if (span.first) rank_samples(blockIdx.x - rank0, (int)threadIdx.x, hints);
        return;
    }
    const uint32_t block = hints.rank_last ? blockIdx.x : blockIdx.x - nrank;
    if (quarters) {
        // A window of few tiles (a rank's strip): EVERY tile on four waves, a quarter of its pixels each.  One wave per
        // tile would leave the SIMDs with one to four waves, and a wave with few neighbours needs up to 3.3x its own
        // issue time per trip (tools/attic/occupancy_probe.py); the quarter form -- one pixel per lane, eight independent
        // alphas per trip, the gather running ahead -- is built for exactly that situation.  Same arithmetic, same
        // pixels.  32 consecutive blocks serve 8 tiles; the 4 quarters of a tile share b % 8, i.e. an XCD, and XCD x
        // gets the x-th eighth of the window's tiles (neighbouring tiles share most of their Gaussians).
        const uint32_t b = block, nt = (uint32_t)g.count(), groups = (nt + 7u) >> 3;
        if (b >= groups * 32u) {
            clear_block(b - groups * 32u, cp, out.ptr);
            return;
        }
        const uint32_t u = ((b >> 5) << 3) | (b & 7u);          // (XCD = b & 7, index inside it = b >> 5)
        const uint32_t per = nt >> 3, extra = nt & 7u, xcd = b & 7u, i = b >> 5;
        if (i >= per + (xcd < extra ? 1u : 0u)) return;
        (void)u;
        blend_long_tile_quarter<REF>(rec, qraw, vals, ranges, g, out, xcd_remap((i << 3) | xcd, nt), (int)((b >> 3) & 3u), sh, budget,
                                     nullptr, lt.redo);
        return;
    }
    if (block < nhelpers) {
        // 32 consecutive blocks serve 8 long tiles; the 4 quarters of a tile share b % 8, i.e. an XCD
        const uint32_t b = block, slot = (b >> 5) * 8u + (b & 7u);
        const int quarter = (int)((b >> 3) & 3u);
        if (slot >= min(*lt.count, lt.max)) return;
        const uint32_t lt_tile = lt.list[slot];
        {
            const int lead = span.axis ? g.wy0 + (int)(lt_tile % (uint32_t)g.nwy()) : g.wx0 + (int)(lt_tile / (uint32_t)g.nwy());
            if (lead < span.lo || lead >= span.hi) return;
        }
        blend_long_tile_quarter<REF>(rec, qraw, vals, ranges, g, out, lt_tile, quarter, sh, budget,
                                     hints.lens ? hints.lens + lt_tile : nullptr, lt.redo);
        return;
    }
    const uint32_t bid = block - nhelpers;
    if (bid >= tile_blocks) {       // (tile_blocks = number of tiles, or 8 x cap with the per-XCD schedule)
        if (span.first) clear_block(bid - tile_blocks, cp, out.ptr);
        return;
    }
    const uint32_t t = hints.xcd_sched ? xcd_scheduled_tile(bid, (uint32_t)g.count(), sched_cap_, sched, hints.header)
                                       : scheduled_tile(bid, (uint32_t)g.count(), sched);
    if (t >= (uint32_t)g.count()) return;
    blend_tile16<VARIANT, REF>(rec, qraw, vals, ranges, g, out, lt, budget, hints, span, t, sh);
}

template <int VARIANT>
#ifndef GSX_PLAIN_WAVES
#define GSX_PLAIN_WAVES 8    // (build-time knob for A/B runs: tools/ab_bench.sh)
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSX_PLAIN_WAVES, 8)))   // 64 VGPRs: every lost wave costs
    blend_tile16_kernel(const Record *__restrict__ rec, const float4 *__restrict__ qraw, const uint32_t *__restrict__ vals,
                        const uint2 *__restrict__ ranges, TileGrid g, OutDesc out, ClearPlan cp, LongTiles lt,
                        uint32_t nhelpers, const uint32_t *__restrict__ sched, uint32_t budget, uint32_t quarters, BlendHints hints,
                        uint32_t tile_blocks, uint32_t sched_cap_, TileSpan span) {
    __shared__ Staged sh;
    blend_tile16_grid<VARIANT, false>(rec, qraw, vals, ranges, g, out, cp, lt, nhelpers, sched, budget, quarters, hints, tile_blocks,
                                      sched_cap_, span, sh);
}

// The same grid with reference-order records evaluated where they turn up (REF, see above): what a frame runs.  128 VGPRs,
// 4 waves per SIMD.  (5 and 6 waves -- 96 and 80 VGPRs, 130 and 206 of them spilled -- measured 1 .. 6 % slower on the
// uniform and the heavy-tailed 1M-Gaussian frames alike, 8 waves -- 314 spilled -- 15 .. 20 %.)
#ifndef GSX_REF_WAVES
#define GSX_REF_WAVES 4      // (build-time knobs for A/B runs: tools/ab_bench.sh)
#endif
#ifndef GSX_REF_VARIANT
#define GSX_REF_VARIANT 1    // (2: trips of six records)
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSX_REF_WAVES, 8)))
    blend_tile16_ref_kernel(const Record *__restrict__ rec, const float4 *__restrict__ qraw, const uint32_t *__restrict__ vals,
                            const uint2 *__restrict__ ranges, TileGrid g, OutDesc out, ClearPlan cp, LongTiles lt,
                            uint32_t nhelpers, const uint32_t *__restrict__ sched, uint32_t budget, uint32_t quarters,
                            BlendHints hints, uint32_t tile_blocks, uint32_t sched_cap_, TileSpan span) {
    __shared__ Staged sh;
    blend_tile16_grid<GSX_REF_VARIANT, true>(rec, qraw, vals, ranges, g, out, cp, lt, nhelpers, sched, budget, quarters, hints,
                                             tile_blocks, sched_cap_, span, sh);
}

// Any tile size: one wave per tile, one pixel per lane, tile*tile/64 sweeps over the list.
// Same arithmetic as the fast path; exists so that tile_size is a run-time argument as in the
// reference (its notebooks use 16 and 2).
__global__ void __launch_bounds__(64)
    blend_generic_kernel(const Record *__restrict__ rec, const float4 *__restrict__ qraw, const uint32_t *__restrict__ vals,
                         const uint2 *__restrict__ ranges, TileGrid g, OutDesc out, ClearPlan cp) {
    __shared__ Staged sh;
    if (blockIdx.x >= (uint32_t)g.count()) {
        clear_block(blockIdx.x - (uint32_t)g.count(), cp, out.ptr);
        return;
    }
    const int lane = threadIdx.x;
    const uint32_t t = xcd_remap(blockIdx.x, (uint32_t)g.count());
    const int tx = g.wx0 + (int)(t / (uint32_t)g.nwy()), ty = g.wy0 + (int)(t % (uint32_t)g.nwy());
    const int Ts = g.tile, npx = Ts * Ts;
    const uint2 rg = ranges[t];
    const bool fast_y = out.stride_y < out.stride_x;
    // tile 16: the four 8x8 blocks of the tile decide for themselves what they keep, like in blend_tile16_kernel (a lane
    // composites a staged record only if its pixel's block keeps it); any other tile size: the tile is one block
    const bool blocks = Ts == 16;
    for (int chunk = 0; chunk < npx; chunk += 64) {
        uint32_t skipped[kBlocks] = {0u, 0u, 0u, 0u}, count[kBlocks];   // colour left out (stage_records): every sweep over the list decides alike
        const int p = chunk + lane;
        const bool valid = p < npx;
        const int pf = p % Ts, ps = p / Ts;
        const int lx = fast_y ? ps : pf, ly = fast_y ? pf : ps;      // the pixel's offset inside the tile
        const int px = tx * Ts + lx, py = ty * Ts + ly;
        const float fx = (float)lx, fy = (float)ly;
        const int blk = blocks ? ((lx >> 3) | ((ly >> 3) << 1)) & 3 : 0;
        float T[1] = {1.0f}, c0[1] = {0.0f}, c1[1] = {0.0f}, c2[1] = {0.0f};
        for (uint32_t base = rg.x; base < rg.y; base += 64) {
            uint32_t nb = min(64u, rg.y - base);
            if (blocks)
                (void)stage_batch<kStageBlocks>(rec, qraw, vals, base, nb, sh, lane, (float)(tx * Ts), (float)(ty * Ts), (float)Ts, skipped, count);
            else
                (void)stage_batch<kStageWhole>(rec, qraw, vals, base, nb, sh, lane, (float)(tx * Ts), (float)(ty * Ts), (float)Ts, skipped, count);
            __syncthreads();
            for (uint32_t k = 0; k < nb; ++k) {
                const Splat s = read_splat(sh, k);
                const float e_p[1] = {fy};
                composite<1>(fx, e_p, s, T, c0, c1, c2, blk, (float)(tx * Ts), (float)(ty * Ts));
            }
            __syncthreads();
            if (__ballot(valid && T[0] > 0.0f) == 0ull) break;
        }
        if (valid) {
            float *o = out.ptr + (int64_t)(px - out.x0) * out.stride_x + (int64_t)(py - out.y0) * out.stride_y;
            o[0] = c0[0];
            o[1] = c1[0];
            o[2] = c2[0];
        }
    }
}

// GSX_SEM_STD_3DGS, one (pixel, Gaussian) pair.  `thr` is the pixel's alpha threshold: 1/255 while
// the pixel is live, +inf once it has stopped (or lies outside the frame), so "alpha < thr" skips
// everything for a dead pixel without a separate flag.  A skipped or stopping pair composites with
// weight 0 -- fma(0, c, C) = C and T - 0 = T exactly -- which keeps the loop free of control flow.
constexpr float kStdAlphaMin = 1.0f / 255.0f, kStdStop = 0.0001f;

__device__ __forceinline__ void std_composite(float pw, float alpha, float cr, float cg, float cb, float &thr, float &T,
                                              float &c0, float &c1, float &c2) {
    const bool use = !(pw > 0.0f) && !(alpha < thr);
    const float ta = T * alpha;
    const bool stop = use && (T - ta < kStdStop);
    const float w = (use && !stop) ? ta : 0.0f;
    thr = stop ? __builtin_inff() : thr;
    c0 = __builtin_fmaf(w, cr, c0);
    c1 = __builtin_fmaf(w, cg, c1);
    c2 = __builtin_fmaf(w, cb, c2);
    T = T - w;
}

// Per-pixel rule sets, any tile size, one pixel per lane, every tile of the frame including partial
// edge tiles.  Not the hot path of this build (the parity target is the CPU semantics); kept simple.
//   GSX_SEM_REF_CUDA (splat/c/render.cu:21-87): per-pixel inclusive bounding-box cull,
//     alpha = min(0.99, opacity * strength), stop (before accumulating) when T(1 - alpha) < 0.001.
//     The truncated means and the a, 2b, c conic are baked into the records by the packing kernels,
//     log2(opacity) rides in the exponent.
//   GSX_SEM_STD_3DGS (published 3DGS forward pass, include/gsx.h): skip when the exponent is > 0,
//     alpha = min(0.99, opacity * exp(exponent)), skip when alpha < 1/255, stop when
//     T(1 - alpha) < 1e-4, out = C + T * background.  The record holds the opacity itself.
template <int SEM>
__global__ void __launch_bounds__(64)
    blend_rules_kernel(const Record *__restrict__ rec, const float4 *__restrict__ bbox,
                       const uint32_t *__restrict__ vals, const uint2 *__restrict__ ranges, TileGrid g,
                       OutDesc out, float bg0, float bg1, float bg2, ClearPlan cp) {
    __shared__ float4 sh[4][64];
    if (blockIdx.x >= (uint32_t)g.count()) {
        clear_block(blockIdx.x - (uint32_t)g.count(), cp, out.ptr);
        return;
    }
    const int lane = threadIdx.x;
    const uint32_t t = xcd_remap(blockIdx.x, (uint32_t)g.count());
    const int tx = g.wx0 + (int)(t / (uint32_t)g.nwy()), ty = g.wy0 + (int)(t % (uint32_t)g.nwy());
    const int Ts = g.tile, npx = Ts * Ts;
    const uint2 rg = ranges[t];
    const bool fast_y = out.stride_y < out.stride_x;
    for (int chunk = 0; chunk < npx; chunk += 64) {
        const int p = chunk + lane;
        const int pf = p % Ts, ps = p / Ts;
        const int px = tx * Ts + (fast_y ? ps : pf), py = ty * Ts + (fast_y ? pf : ps);
        const bool valid = p < npx && px < g.width && py < g.height;   // render.cu:41-44
        const float fx = (float)px, fy = (float)py;
        float T = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        bool done = !valid;
        float thr = valid ? kStdAlphaMin : __builtin_inff();  // GSX_SEM_STD_3DGS: see std_composite
        for (uint32_t base = rg.x; base < rg.y; base += 64) {
            const uint32_t nb = min(64u, rg.y - base);
            if ((uint32_t)lane < nb) {
                const uint32_t gi = vals[base + lane];
                const Record *q = rec + gi;
                sh[0][lane] = q->a;
                sh[1][lane] = q->b;
                sh[2][lane] = q->c;
                if (SEM == GSX_SEM_REF_CUDA) sh[3][lane] = bbox[gi];
            }
            __syncthreads();
            for (uint32_t k = 0; k < nb; ++k) {
                const float4 A = sh[0][k], B = sh[1][k];
                const float cb = sh[2][k].x;
                const float e_x = A.x - fx, e_y = A.y - fy;
                bool use;
                float alpha;
                if (SEM == GSX_SEM_REF_CUDA) {
                    const float4 bb = sh[3][k];
                    use = fx >= bb.x && fx <= bb.y && fy >= bb.z && fy <= bb.w;
                    const float a0 = __builtin_fmaf(e_x * e_x, A.z, B.y);
                    const float pw = __builtin_fmaf(e_y, __builtin_fmaf(e_y, B.x, e_x * A.w), a0);
                    alpha = fminf(0.99f, __builtin_amdgcn_exp2f(pw));
                } else {
                    const float pw = __builtin_fmaf(e_y, __builtin_fmaf(e_y, B.x, e_x * A.w), (e_x * e_x) * A.z);
                    alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(pw));
                    std_composite(pw, alpha, B.z, B.w, cb, thr, T, c0, c1, c2);
                    continue;
                }
                const float ta = T * alpha, test = T - ta;
                if (use && !done) {
                    if (test < 0.001f) {
                        done = true;
                    } else {
                        c0 = __builtin_fmaf(ta, B.z, c0);
                        c1 = __builtin_fmaf(ta, B.w, c1);
                        c2 = __builtin_fmaf(ta, cb, c2);
                        T = test;
                    }
                }
            }
            __syncthreads();
            if (__ballot(SEM == GSX_SEM_STD_3DGS ? thr < 1.0f : !done) == 0ull) break;
        }
        if (valid) {
            float *o = out.ptr + (int64_t)(px - out.x0) * out.stride_x + (int64_t)(py - out.y0) * out.stride_y;
            if (SEM == GSX_SEM_STD_3DGS) {
                c0 = __builtin_fmaf(T, bg0, c0);
                c1 = __builtin_fmaf(T, bg1, c1);
                c2 = __builtin_fmaf(T, bg2, c2);
            }
            o[0] = c0;
            o[1] = c1;
            o[2] = c2;
        }
    }
}

// GSX_SEM_STD_3DGS, tile = 16: one wave per tile, 4 pixels per lane sharing x (the lane / pixel
// assignment of blend_tile16_kernel), so a record read from LDS and the x-only terms of the exponent
// serve 4 evaluations.  Same per-pixel arithmetic as blend_rules_kernel<GSX_SEM_STD_3DGS> -- the two
// give bit-identical frames (tested).  Pixels outside the frame (partial edge tiles) start dead.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8)))
    blend_std16_kernel(const Record *__restrict__ rec, const uint32_t *__restrict__ vals,
                       const uint2 *__restrict__ ranges, TileGrid g, OutDesc out, float bg0, float bg1, float bg2, ClearPlan cp,
                       const uint32_t *__restrict__ sched) {
    __shared__ float4 sh[3][64 + 1];
    __shared__ uint16_t lists[kBlocks][64];     // per 8x8 block of the tile: byte offsets of the records it keeps (see Staged)
    if (blockIdx.x >= (uint32_t)g.count()) {
        clear_block(blockIdx.x - (uint32_t)g.count(), cp, out.ptr);
        return;
    }
    const int lane = threadIdx.x;
    const uint32_t t = scheduled_tile(blockIdx.x, (uint32_t)g.count(), sched);
    const int tx = g.wx0 + (int)(t / (uint32_t)g.nwy()), ty = g.wy0 + (int)(t % (uint32_t)g.nwy());
    const bool y_contig = out.stride_y < out.stride_x;
    const int lx = y_contig ? (lane >> 2) : (lane & 15), ly0 = 4 * (y_contig ? (lane & 3) : (lane >> 4));
    const int px = tx * 16 + lx;
    const int py0 = ty * 16 + ly0;
    const int blk = (lx >> 3) | ((ly0 >> 3) << 1);            // the 8x8 block the lane's four pixels lie in
    const uint16_t *my_list = lists[blk];
    const char *rec_a = reinterpret_cast<const char *>(&sh[0][0]), *rec_b = reinterpret_cast<const char *>(&sh[1][0]),
               *rec_c = reinterpret_cast<const char *>(&sh[2][0]);
    const float cx = (float)px;
    float cy[4], T[4], c0[4], c1[4], c2[4], thr[4];
    bool valid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cy[j] = (float)(py0 + j);
        T[j] = 1.0f;
        c0[j] = c1[j] = c2[j] = 0.0f;
        valid[j] = px < g.width && py0 + j < g.height;
        thr[j] = valid[j] ? kStdAlphaMin : __builtin_inff();
    }
    const uint2 rg = ranges[t];
    const float tile_x0 = (float)(tx * 16), tile_y0 = (float)(ty * 16);
    // a block whose pixels are all done takes no more records (blend_tile16_kernel: `dead`)
    unsigned long long lanes_of[kBlocks];
#pragma unroll
    for (int gb = 0; gb < kBlocks; ++gb) lanes_of[gb] = __ballot(blk == gb);
    uint32_t dead = 0u;
    for (uint32_t base = rg.x; base < rg.y; base += 64) {
        uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, rg.y - base));
        // A record whose alpha stays below 1/255 at every pixel of a BLOCK of the tile (8 x 8 pixels, 16 lanes: round 4,
        // see Staged) is skipped at every pixel there by the published rule, so it need not be on that block's list: the
        // frame is the same bit for bit (the tile lists -- tight rectangles around the alpha = 1/255 ellipse -- keep the
        // corner tiles the ellipse does not reach, and most of a tile's blocks lie outside most of its ellipses).  The
        // bound is the one of stage_records: with M = -Q'' the exponent is -(D1 e0^2 + m11 w^2), w = e1 + (m01 / m11) e0
        // linear in the pixel, minimised over the block's pixel rectangle term by term -- and taken with 1 % + 0.01 of
        // slack, far above the rounding of either side, so that no record whose alpha could round to 1/255 anywhere is
        // dropped.  Every 16-lane group then walks ITS block's list; the wave goes as far as the longest of the four.
        uint32_t bits = 0;
        float4 a, b, c;
        if ((uint32_t)lane < nb) {
            const Record *q = rec + vals[base + lane];
            a = q->a;     // (x, y, Q''00, Q''01 + Q''10)
            b = q->b;     // (Q''11, opacity, r, g)
            c = q->c;
            bits = 0xFu;
            const float m00 = -a.z, m01 = -0.5f * a.w, m11 = -b.x;
            if (m11 > 0.0f && b.y > 0.0f) {
                const float k = m01 / m11, d1 = m00 - m01 * k;
                if (d1 >= 0.0f) {
                    const float lop = __builtin_amdgcn_logf(b.y);
                    bits = 0u;
#pragma unroll
                    for (int gb = 0; gb < kBlocks; ++gb) {
                        const float bx0 = tile_x0 + 8.0f * (float)(gb & 1), by0 = tile_y0 + 8.0f * (float)(gb >> 1);
                        const float ex0 = a.x - bx0, ex1 = a.x - (bx0 + 7.0f);
                        const float ey0 = a.y - by0, ey1 = a.y - (by0 + 7.0f);
                        const float ex_min2 = ex0 * ex1 <= 0.0f ? 0.0f : fminf(ex0 * ex0, ex1 * ex1);
                        const float w00 = __builtin_fmaf(k, ex0, ey0), w01 = __builtin_fmaf(k, ex0, ey1);
                        const float w10 = __builtin_fmaf(k, ex1, ey0), w11 = __builtin_fmaf(k, ex1, ey1);
                        const float wlo = fminf(fminf(w00, w01), fminf(w10, w11)), whi = fmaxf(fmaxf(w00, w01), fmaxf(w10, w11));
                        const float wabs = fminf(fabsf(wlo), fabsf(whi));
                        const float w_min2 = (wlo <= 0.0f && whi >= 0.0f) ? 0.0f : wabs * wabs;
                        const float least = 0.99f * (d1 * ex_min2 + m11 * w_min2);       // -exponent is at least this
                        if (!(lop - least < -7.994353f - 0.01f)) bits |= 1u << gb;      // log2(1/255); NaN keeps
                    }
                }
            }
        }
        bits &= ~dead;
        const unsigned long long kept = __ballot(bits != 0u);
        nb = (uint32_t)__popcll(kept);
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(kept >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kept, 0u));
        uint32_t nl = 0;
#pragma unroll
        for (int gb = 0; gb < kBlocks; ++gb) {
            const unsigned long long kb = __ballot((bits >> gb) & 1u);
            lists[gb][lane] = (uint16_t)(nb * 16u);                       // the null record behind the batch ...
            if ((bits >> gb) & 1u)                                        // ... then the block's own records, in order
                lists[gb][__builtin_amdgcn_mbcnt_hi((uint32_t)(kb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kb, 0u))] = (uint16_t)(slot * 16u);
            nl = max(nl, (uint32_t)__popcll(kb));
        }
        if (bits != 0u) {
            sh[0][slot] = a;
            sh[1][slot] = b;
            sh[2][slot] = c;
        }
        if (lane == 0) {        // a null record: opacity 0 -> alpha 0 < 1/255, skipped by the rule itself
            sh[0][nb] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            sh[1][nb] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            sh[2][nb] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        for (uint32_t k = 0; k < nl; ++k) {
            const uint32_t off = my_list[k];
            const float4 A = *reinterpret_cast<const float4 *>(rec_a + off), B = *reinterpret_cast<const float4 *>(rec_b + off);
            const float cb = *reinterpret_cast<const float *>(rec_c + off);
            const float e_x = A.x - cx;
            const float a0 = (e_x * e_x) * A.z, b0 = e_x * A.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e_y = A.y - cy[j];
                const float pw = __builtin_fmaf(e_y, __builtin_fmaf(e_y, B.x, b0), a0);
                const float alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(pw));
                std_composite(pw, alpha, B.z, B.w, cb, thr[j], T[j], c0[j], c1[j], c2[j]);
            }
        }
        __syncthreads();
        const unsigned long long alive = __ballot(fminf(fminf(thr[0], thr[1]), fminf(thr[2], thr[3])) < 1.0f);
        if (alive == 0ull) break;
#pragma unroll
        for (int gb = 0; gb < kBlocks; ++gb) dead |= (alive & lanes_of[gb]) == 0ull ? (1u << gb) : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c0[j] = __builtin_fmaf(T[j], bg0, c0[j]);
        c1[j] = __builtin_fmaf(T[j], bg1, c1[j]);
        c2[j] = __builtin_fmaf(T[j], bg2, c2[j]);
    }
    float *o = out.ptr + (int64_t)(px - out.x0) * out.stride_x + (int64_t)(py0 - out.y0) * out.stride_y;
    if (y_contig && valid[3] && (reinterpret_cast<uintptr_t>(o) & 15u) == 0) {
        float4 *o4 = reinterpret_cast<float4 *>(o);
        o4[0] = make_float4(c0[0], c1[0], c2[0], c0[1]);
        o4[1] = make_float4(c1[1], c2[1], c0[2], c1[2]);
        o4[2] = make_float4(c2[2], c0[3], c1[3], c2[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!valid[j]) continue;
            float *oj = o + (int64_t)j * out.stride_y;
            oj[0] = c0[j];
            oj[1] = c1[j];
            oj[2] = c2[j];
        }
    }
}

__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t *__restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}

}  // namespace

hipError_t launch_zero_words(uint32_t *p, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    zero_words_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p, n);
    return hipGetLastError();
}

hipError_t launch_clear(const ClearPlan &cp, float *base, hipStream_t s) {
    if (cp.n <= 0 || cp.first[cp.n] <= 0) return hipSuccess;
    clear_kernel<<<(unsigned)cp.first[cp.n], 64, 0, s>>>(cp, base);
    return hipGetLastError();
}

// Up to this many tiles in the window every tile is composited by four waves (blend_tile16_kernel, `quarters`).
// Measured (round 3, tools/attic/strip_probe.py with GSX_QUARTERS_BELOW): it does NOT pay -- a 1/8 strip of the 5M / 4K frame
// composites in 0.192 ms on one wave per tile and in 0.314 ms on four, 1M / 1080p: 0.073 vs 0.090 ms; the quarter form
// issues 14 operations per pixel and record where the 4-pixel form shares the x terms (11.25) and stages every
// record four times.  The mode stays in the kernel (default off) as the strongest test of the quarter path: a frame
// rendered with it must equal the normal frame bit for bit.
constexpr int kQuartersBelow = 0;

#ifdef GSX_TEST_HOOKS
hipError_t set_blend_probe(void *device_buffer) {
    uint4 *p = (uint4 *)device_buffer;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_blend_probe), &p, sizeof p);
}
#endif

bool blend_splits_long_tiles(const TileGrid &grid, int semantics, bool generic) {
    if (knob("GSX_LONG_SPLIT", 1) == 0) return false;     // test library only
    return semantics == GSX_SEM_REF_CPU && grid.tile == 16 && !generic;
}
// The schedule is one more kernel on the frame's critical path (5 .. 10 us): it pays when the compositing
// kernel runs for 100+ us (1M Gaussians at 1080p: -33 us of compositing), not for the small scenes (100 000
// Gaussians: -3 us).  D is not known on the host; the Gaussian count is.
bool blend_uses_schedule(const TileGrid &grid, int semantics, bool generic, int64_t n, int asked) {
    const int forced = knob("GSX_TILE_SCHEDULE", -1);   // test library only: 0 never, 1 always
    if ((semantics != GSX_SEM_REF_CPU && semantics != GSX_SEM_STD_3DGS) || grid.tile != 16 || generic) return false;
    // a window whose tiles all go on four waves (launch_blend) is not handed out by list length
    if (semantics == GSX_SEM_REF_CPU && grid.count() <= (int64_t)knob("GSX_QUARTERS_BELOW", kQuartersBelow)) return false;
    if (asked >= 0) return asked != 0;          // GSX_FLAG_TILE_SCHEDULE / GSX_FLAG_NO_TILE_SCHEDULE
    if (forced >= 0) return forced != 0;
    // a window of up to 2048 tiles (a rank's strip of a 1080p frame) puts at most two tiles on a SIMD: the
    // kernel lasts as long as its longest tile whatever the hand-out
    return n >= 300000 && grid.count() > 2048;
}

bool blend_in_parts(const TileGrid &grid, int semantics, bool generic) {
    return semantics == GSX_SEM_REF_CPU && grid.tile == 16 && !generic && grid.count() > (int64_t)knob("GSX_QUARTERS_BELOW", kQuartersBelow);
}

hipError_t launch_blend(const Record *rec, const float4 *bbox, const uint32_t *sorted_vals, const uint2 *ranges,
                        const TileGrid &grid, const OutDesc &out, int semantics, const float *background,
                        bool generic, const ClearPlan &cp, const LongTiles &lt, const uint32_t *sched,
                        const BlendHints &hints, hipStream_t s, const TileSpan *part) {
    const TileSpan span = part ? *part : TileSpan{0, -2147483647 - 1, 2147483647, 1u};
    const int64_t nt = grid.count();
    if (nt <= 0) return launch_clear(cp, out.ptr, s);
    const unsigned nb = (unsigned)nt + (unsigned)(cp.n > 0 ? cp.first[cp.n] : 0);
    if (semantics == GSX_SEM_REF_CUDA) {
        blend_rules_kernel<GSX_SEM_REF_CUDA><<<nb, 64, 0, s>>>(rec, bbox, sorted_vals, ranges, grid, out, 0.0f, 0.0f, 0.0f, cp);
        return hipGetLastError();
    }
    if (semantics == GSX_SEM_STD_3DGS) {
        if (grid.tile == 16 && !generic) {
            blend_std16_kernel<<<nb, 64, 0, s>>>(rec, sorted_vals, ranges, grid, out, background[0], background[1],
                                                 background[2], cp, sched);
            return hipGetLastError();
        }
        blend_rules_kernel<GSX_SEM_STD_3DGS><<<nb, 64, 0, s>>>(rec, bbox, sorted_vals, ranges, grid, out, background[0],
                                                              background[1], background[2], cp);
        return hipGetLastError();
    }
    if (semantics != GSX_SEM_REF_CPU) return hipErrorNotSupported;
    if (grid.tile == 16 && !generic) {
        const int variant = knob("GSX_BLEND_VARIANT", 1);   // test library only: A/B runs of the compositing loop
        // test library only: GSX_SKIP_BUDGET_LOG2 = -9 hardly ever refuses a skip (round 2's behaviour); default 2^-17
        const uint32_t budget = 1u << (40 + knob("GSX_SKIP_BUDGET_LOG2", -17));
        // a window of up to kQuartersBelow tiles puts every tile on four waves (see the kernel)
        const bool quarters = nt <= (int64_t)knob("GSX_QUARTERS_BELOW", kQuartersBelow);
        BlendHints bh = hints;
        if (quarters) bh.lens = nullptr;
        // GSX_FLAG_PLAIN_FOOTPRINTS: the instance that does not evaluate reference-order records (see tile_sync)
        const bool plain = bh.plain != 0u && lt.redo != nullptr;
        // tile workgroups: one per tile, or -- per-XCD schedule from GsxParams.hints -- 8 x cap (gsx_schedule_device.h)
        const uint32_t cap = bh.xcd_sched ? sched_cap((uint32_t)nt, (uint32_t)grid.nwy()) : 0u;
        const unsigned tile_blocks = bh.xcd_sched ? kSchedXcds * cap : (unsigned)nt;
        const unsigned clear_blocks = (unsigned)(cp.n > 0 ? cp.first[cp.n] : 0);
        unsigned nh = lt.max ? 4u * lt.max : 0u, grid_blocks = tile_blocks + clear_blocks + nh;
        if (quarters) {
            nh = 0;
            grid_blocks = (unsigned)((nt + 7) / 8) * 32u + clear_blocks;
        }
        grid_blocks += bh.samples ? kRankGroups : 0u;
        const uint32_t q = quarters ? 1u : 0u;
        if (variant == 1 && !plain)
            blend_tile16_ref_kernel<<<grid_blocks, 64, 0, s>>>(rec, bbox, sorted_vals, ranges, grid, out, cp, lt, nh, sched, budget, q, bh, tile_blocks, cap, span);
        else if (variant == 0)
            blend_tile16_kernel<0><<<grid_blocks, 64, 0, s>>>(rec, bbox, sorted_vals, ranges, grid, out, cp, lt, nh, sched, budget, q, bh, tile_blocks, cap, span);
        else if (variant == 2)
            blend_tile16_kernel<2><<<grid_blocks, 64, 0, s>>>(rec, bbox, sorted_vals, ranges, grid, out, cp, lt, nh, sched, budget, q, bh, tile_blocks, cap, span);
        else if (variant == 3)
            blend_tile16_kernel<3><<<grid_blocks, 64, 0, s>>>(rec, bbox, sorted_vals, ranges, grid, out, cp, lt, nh, sched, budget, q, bh, tile_blocks, cap, span);
        else
            blend_tile16_kernel<1><<<grid_blocks, 64, 0, s>>>(rec, bbox, sorted_vals, ranges, grid, out, cp, lt, nh, sched, budget, q, bh, tile_blocks, cap, span);
    } else {
        blend_generic_kernel<<<nb, 64, 0, s>>>(rec, bbox, sorted_vals, ranges, grid, out, cp);
    }
    return hipGetLastError();
}

}  // namespace gsx
