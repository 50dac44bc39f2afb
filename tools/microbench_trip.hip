// Testbed for the compositing loop's trip (four records x the lane's four pixels): the loop body of blend_tile16_kernel
// alone -- records staged in LDS once, then kIters passes over the 64 staged records, 8 single-wave workgroups per SIMD
// like the real launch -- in several formulations of alpha.  Prints cycles per trip and SIMD (wall time x clock / trips
// issued on one SIMD), i.e. what the formulation costs in issue time when nothing else (gather, staging, tail) is in
// the way.  Round 4: decides between the instruction-cost models.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o microbench_trip tools/microbench_trip.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kIters = 512, kRec = 64;

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float v) { return v2f{v, v}; }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float min4(v2f a, v2f b) { return fminf(fminf(a.x, a.y), fminf(b.x, b.y)); }

#define ACC(ta_a, ta_b, cr, cg, cb)            \
    do {                                       \
        c0a = pk_fma(ta_a, splat2(cr), c0a);   \
        c1a = pk_fma(ta_a, splat2(cg), c1a);   \
        c2a = pk_fma(ta_a, splat2(cb), c2a);   \
        c0b = pk_fma(ta_b, splat2(cr), c0b);   \
        c1b = pk_fma(ta_b, splat2(cg), c1b);   \
        c2b = pk_fma(ta_b, splat2(cb), c2b);   \
    } while (0)

// LDS images.  Per-record layout (V0 .. V2): a = (x', c0, D1, h), b = (r11, lop, 2 r11, -r11^2), hb = (hx, blue) / (hx, hx^3)
// pairs, rg = (red, green).  Pair layout (V3, V4): for records 2i, 2i+1: xa = (x'0, x'1, c0_0, c0_1), xb = (D1_0, D1_1, h_0, h_1),
// xc = (r11_0, r11_1, lop_0, lop_1), xd = (tr_0, tr_1, nr2_0, nr2_1), he = (hx_0, hx3_0, hx_1, hx3_1), col = rgb of both.
struct Lds {
    float4 a[kRec + 8], b[kRec + 8];
    float2 hb[kRec + 8], rg[kRec + 8], h3[kRec + 8];
    float4 xa[kRec / 2 + 4], xb[kRec / 2 + 4], xc[kRec / 2 + 4], xd[kRec / 2 + 4], he[kRec / 2 + 4];
    float bl[kRec + 8];
    uint8_t lists[4][kRec + 8];
};

__device__ __forceinline__ v4f lds4(const float4 *p) { return *reinterpret_cast<const v4f *>(p); }

template <int V>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) trip_kernel(float *out, const float *in) {
    __shared__ Lds sh;
    const int lane = threadIdx.x;
    {   // one fake record per lane: a small splat near the tile
        const float xr = in[lane] * 16.0f, yr = in[64 + lane] * 16.0f, r11 = 0.3f + in[128 + lane], h = in[192 + lane] - 0.5f;
        const float d1 = 0.2f + in[lane + 1], lop = -1.0f - in[lane + 2];
        const float tr = r11 + r11, nr2 = -(r11 * r11), hx = ex2(nr2 + nr2);
        sh.a[lane] = make_float4(xr, __builtin_fmaf(r11, yr, h * xr), d1, h);
        sh.b[lane] = make_float4(r11, lop, tr, nr2);
        sh.hb[lane] = make_float2(hx, 0.5f);
        sh.h3[lane] = make_float2(hx, hx * hx * hx);
        sh.rg[lane] = make_float2(0.25f, 0.75f);
        sh.bl[lane] = 0.5f;
        for (int g4 = 0; g4 < 4; ++g4) sh.lists[g4][lane] = (uint8_t)((lane * 5 + 16 * g4 + 3 * (lane >> 2)) & 63);   // four different walks
        float *xa = reinterpret_cast<float *>(sh.xa), *xb = reinterpret_cast<float *>(sh.xb), *xc = reinterpret_cast<float *>(sh.xc);
        float *xd = reinterpret_cast<float *>(sh.xd), *he = reinterpret_cast<float *>(sh.he);
        const int p = lane >> 1, q = lane & 1;
        xa[4 * p + q] = xr;  xa[4 * p + 2 + q] = __builtin_fmaf(r11, yr, h * xr);
        xb[4 * p + q] = d1;  xb[4 * p + 2 + q] = h;
        xc[4 * p + q] = r11; xc[4 * p + 2 + q] = lop;
        xd[4 * p + q] = tr;  xd[4 * p + 2 + q] = nr2;
        he[4 * p + 2 * q] = hx; he[4 * p + 2 * q + 1] = hx * hx * hx;
    }
    __syncthreads();
    const unsigned long long cyc0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
    const float cx = (float)(lane >> 2), cy0 = (float)(4 * (lane & 3));
    const v2f cya = v2f{cy0, cy0 + 1.0f}, cyb = v2f{cy0 + 2.0f, cy0 + 3.0f};
    v2f Ta = splat2(1.0f), Tb = splat2(1.0f);
    v2f c0a = splat2(0.0f), c1a = c0a, c2a = c0a, c0b = c0a, c1b = c0a, c2b = c0a;
    int stops = 0;
    for (int it = 0; it < kIters; ++it) {
        for (uint32_t k = 0; k < kRec; k += 4) {
            v2f aa[4], ab[4];
            float cb[4];
            uint32_t slots[4] = {k, k + 1, k + 2, k + 3};
            if (V == 0) {           // round 3: direct, four v_exp_f32 per record
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const v4f A = lds4(&sh.a[k + u]);
                    const v2f Bq = *reinterpret_cast<const v2f *>(&sh.b[k + u]);
                    cb[u] = sh.bl[k + u];
                    const float e_x = A.x - cx;
                    const float s0 = __builtin_fmaf(-(A.z * e_x), e_x, Bq.y);
                    const float c = __builtin_fmaf(-A.w, cx, A.y);
                    const v2f wa = pk_fma(splat2(-Bq.x), cya, splat2(c)), wb = pk_fma(splat2(-Bq.x), cyb, splat2(c));
                    const v2f ea = pk_fma(-wa, wa, splat2(s0)), eb = pk_fma(-wb, wb, splat2(s0));
                    aa[u] = v2f{ex2(ea.x), ex2(ea.y)};
                    ab[u] = v2f{ex2(eb.x), ex2(eb.y)};
                }
            } else if (V == 5) {    // round 3's arithmetic, but every 16-lane group (an 8x8 block of the tile) walks its OWN list
                const int grp = lane >> 4;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t slot = sh.lists[grp][k + u];            // 4 distinct addresses per wave instruction
                    const v4f A = lds4(&sh.a[slot]);
                    const v2f Bq = *reinterpret_cast<const v2f *>(&sh.b[slot]);
                    cb[u] = sh.bl[slot];
                    slots[u] = slot;
                    const float e_x = A.x - cx;
                    const float s0 = __builtin_fmaf(-(A.z * e_x), e_x, Bq.y);
                    const float c = __builtin_fmaf(-A.w, cx, A.y);
                    const v2f wa = pk_fma(splat2(-Bq.x), cya, splat2(c)), wb = pk_fma(splat2(-Bq.x), cyb, splat2(c));
                    const v2f ea = pk_fma(-wa, wa, splat2(s0)), eb = pk_fma(-wb, wb, splat2(s0));
                    aa[u] = v2f{ex2(ea.x), ex2(ea.y)};
                    ab[u] = v2f{ex2(eb.x), ex2(eb.y)};
                }
            } else if (V == 1 || V == 2) {   // recurrence per record: scalar chain (1) / pixel-pair products (2)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const v4f HB = V == 1 ? lds4(reinterpret_cast<const float4 *>(&sh.hb[k + u])) : lds4(reinterpret_cast<const float4 *>(&sh.h3[k + u]));
                    cb[u] = V == 1 ? HB.y : 0.5f;
                    cb[u + 1] = V == 1 ? HB.w : 0.5f;
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const v4f A = lds4(&sh.a[k + u + v]), B = lds4(&sh.b[k + u + v]);
                        const float hx = v ? HB.z : HB.x, hx3 = v ? HB.w : HB.y;
                        const float e_x = A.x - cx;
                        const float s0 = __builtin_fmaf(-(A.z * e_x), e_x, B.y);
                        const float c = __builtin_fmaf(-A.w, cx, A.y);
                        const float w0 = __builtin_fmaf(-B.x, cy0, c);
                        const float a0 = ex2(__builtin_fmaf(-w0, w0, s0));
                        const float g0 = ex2(__builtin_fmaf(B.z, w0, B.w));
                        if (V == 1) {
                            const float a1 = a0 * g0, g1 = g0 * hx, a2 = a1 * g1, g2 = g1 * hx, a3 = a2 * g2;
                            aa[u + v] = v2f{a0, a1};
                            ab[u + v] = v2f{a2, a3};
                        } else {
                            const v2f al = v2f{a0, a0 * g0};
                            const v2f qq = (splat2(g0) * splat2(g0)) * v2f{hx, hx3};
                            aa[u + v] = al;
                            ab[u + v] = al * qq;
                        }
                    }
                }
            } else {                // record pairs packed: x terms, w0, exponents as v_pk (3: recurrence, 4: direct)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const uint32_t p = (k + u) >> 1;
                    const v4f XA = lds4(&sh.xa[p]), XB = lds4(&sh.xb[p]), XC = lds4(&sh.xc[p]);
                    const v2f X = v2f{XA.x, XA.y}, C0 = v2f{XA.z, XA.w}, D = v2f{XB.x, XB.y}, H = v2f{XB.z, XB.w};
                    const v2f R = v2f{XC.x, XC.y}, L = v2f{XC.z, XC.w};
                    cb[u] = 0.5f;
                    cb[u + 1] = 0.5f;
                    const v2f E = X - splat2(cx);
                    const v2f S0 = pk_fma(-(D * E), E, L);
                    const v2f C = pk_fma(-H, splat2(cx), C0);
                    if (V == 3) {
                        const v4f XD = lds4(&sh.xd[p]), HE = lds4(&sh.he[p]);
                        const v2f TR = v2f{XD.x, XD.y}, NR = v2f{XD.z, XD.w};
                        const v2f W0 = pk_fma(-R, splat2(cy0), C);
                        const v2f E0 = pk_fma(-W0, W0, S0), G0 = pk_fma(TR, W0, NR);
                        const v2f G = v2f{ex2(G0.x), ex2(G0.y)};
                        const v2f GG = G * G;
                        const float a00 = ex2(E0.x), a01 = ex2(E0.y);
                        const v2f al0 = v2f{a00, a00 * G.x}, al1 = v2f{a01, a01 * G.y};
                        aa[u] = al0;
                        ab[u] = al0 * (splat2(GG.x) * v2f{HE.x, HE.y});
                        aa[u + 1] = al1;
                        ab[u + 1] = al1 * (splat2(GG.y) * v2f{HE.z, HE.w});
                    } else {
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const float r11 = v ? R.y : R.x, c = v ? C.y : C.x, s0 = v ? S0.y : S0.x;
                            const v2f wa = pk_fma(splat2(-r11), cya, splat2(c)), wb = pk_fma(splat2(-r11), cyb, splat2(c));
                            const v2f ea = pk_fma(-wa, wa, splat2(s0)), eb = pk_fma(-wb, wb, splat2(s0));
                            aa[u + v] = v2f{ex2(ea.x), ex2(ea.y)};
                            ab[u + v] = v2f{ex2(eb.x), ex2(eb.y)};
                        }
                    }
                }
            }
            v2f ta_a[4], ta_b[4], ta = Ta, tb = Tb;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ta_a[u] = ta * aa[u];
                ta_b[u] = tb * ab[u];
                ta = ta - ta_a[u];
                tb = tb - ta_b[u];
            }
            if (__builtin_expect(__any(!(min4(ta, tb) >= 1e-6f)), 0)) {     // (restart the pixel: keeps the loop honest)
                ta = splat2(1.0f);
                tb = splat2(1.0f);
                ++stops;
            }
            if (V == 5) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const v2f RG = *reinterpret_cast<const v2f *>(&sh.rg[slots[u]]);
                    ACC(ta_a[u], ta_b[u], RG.x, RG.y, cb[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const v4f RG = lds4(reinterpret_cast<const float4 *>(&sh.rg[k + u]));
                    ACC(ta_a[u], ta_b[u], RG.x, RG.y, cb[u]);
                    ACC(ta_a[u + 1], ta_b[u + 1], RG.z, RG.w, cb[u + 1]);
                }
            }
            Ta = ta;
            Tb = tb;
        }
    }
    const float s = c0a.x + c0a.y + c1a.x + c1a.y + c2a.x + c2a.y + c0b.x + c0b.y + c1b.x + c1b.y + c2b.x + c2b.y + Ta.x + Tb.y;
    if (s == 12345.678f || stops == -1) out[0] = s;
    if (blockIdx.x == 0) out[1 + lane] = s;
    if (lane == 0 && blockIdx.x < 64) {     // shader cycles per 100 MHz tick: the clock this variant ran at
        out[128 + 2 * blockIdx.x] = (float)(__builtin_readcyclecounter() - cyc0);
        out[129 + 2 * blockIdx.x] = (float)(wall_clock64() - wall0);
    }
}

template <int V>
void run(const char *name, double mhz, float *buf) {
    const int grid = 256 * 4 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    trip_kernel<V><<<grid, 64>>>(buf, buf + 1024);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    trip_kernel<V><<<grid, 64>>>(buf, buf + 1024);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms, probe[8], clk[128];
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(probe, buf + 1, sizeof probe, hipMemcpyDeviceToHost);
    (void)hipMemcpy(clk, buf + 128, sizeof clk, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int i = 0; i < 64; ++i) ghz += clk[2 * i] / clk[2 * i + 1] * 0.1 / 64;
    const double trips = (double)kIters * (kRec / 4) * 8;  // per SIMD
    printf("%-58s %.3f ms  %.1f cycles/trip/SIMD at %.0f MHz  measured clock %.3f GHz -> %.1f real cycles/trip  [%g]\n", name, ms, ms * 1e-3 * mhz * 1e6 / trips, mhz,
           ghz, ms * 1e-3 * ghz * 1e9 / trips, probe[3]);
}

int main() {
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double mhz = khz / 1000.0;
    float *buf, host[512];
    for (int i = 0; i < 512; ++i) host[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f;
    (void)hipMalloc(&buf, 1 << 16);
    (void)hipMemset(buf, 0, 1 << 16);
    (void)hipMemcpy(buf + 1024, host, sizeof host, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("V0 direct: 4 exp + 4 pk_fma per record (round 3)", mhz, buf);
        run<1>("V1 recurrence, scalar chain (2 exp + 5 mul)", mhz, buf);
        run<2>("V2 recurrence, pixel-pair products (2 exp + 1 mul + 3 pk)", mhz, buf);
        run<3>("V3 = V2 with x terms / w0 / exponents packed over 2 records", mhz, buf);
        run<4>("V4 = V0 with x terms packed over 2 records", mhz, buf);
        run<5>("V5 = V0, every 16-lane group walks its own list", mhz, buf);
    }
    return 0;
}
