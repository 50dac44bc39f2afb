// Issue-rate microbenchmark for the VALU instructions the compositing loop is made of (gfx950).
// Every workgroup is one wave; the grid fills every SIMD with W waves.  Prints cycles per
// instruction per SIMD (wall time x clock / instructions issued on one SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o microbench_valu tools/microbench_valu.hip && ./microbench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kUnroll = 16;

template <int OP>
__global__ void __launch_bounds__(64) k(float *out, float seed) {
    float a[kUnroll];
    v2f p[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
        a[j] = seed + threadIdx.x * 1e-3f + j;
        p[j] = v2f{a[j], a[j] + 0.5f};
    }
    const float m = seed * 0.999f, c = seed * 1e-3f;
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            if (OP == 0) a[j] = __builtin_fmaf(a[j], m, c);
            if (OP == 1) p[j] = __builtin_elementwise_fma(p[j], v2f{m, m}, v2f{c, c});
            if (OP == 2) a[j] = __builtin_amdgcn_exp2f(a[j]);
            if (OP == 3) p[j] = p[j] * v2f{m, m};
            if (OP == 4) p[j] = p[j] - v2f{c, c};
            if (OP == 5) a[j] = fminf(a[j], m);
            if (OP == 6) {  // the loop's mix: 1 exp per 2 packed fma + 1 packed mul + 1 packed sub (per 2 px)
                p[j] = __builtin_elementwise_fma(p[j], v2f{m, m}, v2f{c, c});
                p[j] = __builtin_elementwise_fma(p[j], v2f{m, m}, v2f{c, c});
                p[j].x = __builtin_amdgcn_exp2f(p[j].x);
                p[j].y = __builtin_amdgcn_exp2f(p[j].y);
            }
            if (OP == 7) {  // exp interleaved with independent plain fma (co-issue test)
                a[j] = __builtin_amdgcn_exp2f(a[j]);
                p[j].x = __builtin_fmaf(p[j].x, m, c);
                p[j].y = __builtin_fmaf(p[j].y, m, c);
                p[j].x = __builtin_fmaf(p[j].x, m, c);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) s += a[j] + p[j].x + p[j].y;
    if (OP >= 8) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int i = 0; i < kIters; ++i) {
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                acc[j & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], m, acc[j & 3], 0, 0, 0);
                if (OP == 9) {  // + 4 independent plain fma per mfma
                    p[j].x = __builtin_fmaf(p[j].x, m, c);
                    p[j].y = __builtin_fmaf(p[j].y, m, c);
                    p[(j + 1) % kUnroll].x = __builtin_fmaf(p[(j + 1) % kUnroll].x, m, c);
                    p[(j + 1) % kUnroll].y = __builtin_fmaf(p[(j + 1) % kUnroll].y, m, c);
                }
                if (OP == 10) {  // + 2 packed fma per mfma
                    p[j] = __builtin_elementwise_fma(p[j], v2f{m, m}, v2f{c, c});
                    p[(j + 5) % kUnroll] = __builtin_elementwise_fma(p[(j + 5) % kUnroll], v2f{m, m}, v2f{c, c});
                }
            }
        }
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        for (int j = 0; j < kUnroll; ++j) s += p[j].x + p[j].y;
    }
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char *name, int instr_per_unroll, int waves_per_simd, double mhz) {
    float *out;
    hipMalloc(&out, 4);
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<grid, 64>>>(out, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<grid, 64>>>(out, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)kIters * kUnroll * instr_per_unroll * waves_per_simd;  // per SIMD
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cycles/instr/SIMD\n", name, waves_per_simd, ms,
           ms * 1e-3 * mhz * 1e6 / instr);
    hipFree(out);
}

int main() {
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double mhz = khz / 1000.0;
    printf("clock %.0f MHz\n", mhz);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 1, w, mhz);
        run<1>("v_pk_fma_f32", 1, w, mhz);
        run<2>("v_exp_f32", 1, w, mhz);
        run<3>("v_pk_mul_f32", 1, w, mhz);
        run<4>("v_pk_add_f32", 1, w, mhz);
        run<5>("v_min_f32", 1, w, mhz);
        run<6>("2 pk_fma + 2 exp", 4, w, mhz);
        run<7>("1 exp + 3 fma", 4, w, mhz);
        run<8>("mfma_4x4x1 (per mfma)", 1, w, mhz);
        run<9>("mfma + 4 fma (per group)", 1, w, mhz);
        run<10>("mfma + 2 pk_fma (per group)", 1, w, mhz);
    }
    return 0;
}
