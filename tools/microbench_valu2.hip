// Issue cost of VALU instructions whose operands are all VGPRs (round 4: the round-1 microbenchmark fed its plain
// v_fma_f32 two SGPR operands -- 2.5 cycles -- while the PMC counters of the compositing kernel average 4.3-4.6 busy
// cycles per VALU instruction whatever the share of packed ones).  One wave per workgroup, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o microbench_valu2 tools/microbench_valu2.hip && ./microbench_valu2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kUnroll = 8;

template <int OP>
__global__ void __launch_bounds__(64) k(float *out, const float *in) {
    float a[kUnroll], b[kUnroll], c[kUnroll];
    v2f p[kUnroll], q[kUnroll], r[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
        a[j] = in[threadIdx.x + 64 * j];
        b[j] = in[threadIdx.x + 64 * j + 1];
        c[j] = in[threadIdx.x + 64 * j + 2];
        p[j] = v2f{a[j], b[j]};
        q[j] = v2f{b[j], c[j]};
        r[j] = v2f{c[j], a[j]};
    }
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            if (OP == 0) a[j] = __builtin_fmaf(a[j], b[j], c[j]);                 // 3 VGPR sources
            if (OP == 1) a[j] = a[j] * b[j];                                      // 2 VGPR sources
            if (OP == 2) p[j] = __builtin_elementwise_fma(p[j], q[j], r[j]);      // 3 VGPR pairs
            if (OP == 3) p[j] = p[j] * q[j];
            if (OP == 4) a[j] = __builtin_fmaf(a[j], b[0], c[0]);                 // shared VGPR sources
            if (OP == 5) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p[j]) : "v"(p[j]), "v"(q[j]));
            if (OP == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[j]) : "v"(a[j]));
            if (OP == 7) asm volatile("v_mul_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "=v"(a[j]) : "v"(a[j]), "v"(b[j]));
            if (OP == 8) a[j] = __builtin_amdgcn_exp2f(a[j]);
            if (OP == 9) {      // a dependent chain of plain multiplications (the recurrence as first written)
                a[j] = a[j] * b[j];
                b[j] = b[j] * c[j];
            }
            if (OP == 10) a[j] = fminf(a[j], b[j]);
            if (OP == 11) a[j] = a[j] - b[j];
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) s += a[j] + b[j] + p[j].x + p[j].y;
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char *name, int instr_per_unroll, int waves_per_simd, double mhz, float *buf) {
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<grid, 64>>>(buf, buf + 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<grid, 64>>>(buf, buf + 16);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)kIters * kUnroll * instr_per_unroll * waves_per_simd;  // per SIMD
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cycles/instr/SIMD at %.0f MHz\n", name, waves_per_simd, ms, ms * 1e-3 * mhz * 1e6 / instr, mhz);
}

int main() {
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double mhz = khz / 1000.0;
    float *buf;
    hipMalloc(&buf, 1 << 16);
    hipMemset(buf, 0, 1 << 16);
    for (int w : {4, 8}) {
        run<0>("v_fma_f32 (3 VGPR)", 1, w, mhz, buf);
        run<1>("v_mul_f32 (2 VGPR)", 1, w, mhz, buf);
        run<4>("v_fma_f32 (1 own + 2 shared VGPR)", 1, w, mhz, buf);
        run<10>("v_min_f32 (2 VGPR)", 1, w, mhz, buf);
        run<11>("v_sub_f32 (2 VGPR)", 1, w, mhz, buf);
        run<2>("v_pk_fma_f32 (3 VGPR pairs)", 1, w, mhz, buf);
        run<3>("v_pk_mul_f32 (2 VGPR pairs)", 1, w, mhz, buf);
        run<5>("v_pk_mov_b32", 1, w, mhz, buf);
        run<6>("v_mov_b32_dpp quad_perm", 1, w, mhz, buf);
        run<7>("v_mul_f32_dpp quad_perm", 1, w, mhz, buf);
        run<8>("v_exp_f32", 1, w, mhz, buf);
        run<9>("2 dependent v_mul_f32", 2, w, mhz, buf);
    }
    return 0;
}
