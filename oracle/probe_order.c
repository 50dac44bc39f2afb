/*
 * Candidate float32 evaluation orders for the small matrix products of the reference's stage 1
 * and weight evaluation.  TEST INFRASTRUCTURE ONLY: built on the fly by oracle/probe_torch_order.py,
 * which runs torch (the arithmetic the reference executes) beside each candidate and counts bit
 * differences.  Compile with -ffp-contract=off so only the explicit fmaf calls fuse.
 */
#include <math.h>
#include <stdint.h>

enum { L2R = 0, FMA_SEQ = 1, FMA_REV = 2, PAIRWISE = 3, F64 = 4, L2R_REV = 5, FMA_PAIR = 6 };

static float dot(const float *a, int sa, const float *b, int sb, int K, int mode) {
    float acc;
    switch (mode) {
    case L2R:
        acc = a[0] * b[0];
        for (int k = 1; k < K; ++k) acc = acc + a[k * sa] * b[k * sb];
        return acc;
    case FMA_SEQ:
        acc = a[0] * b[0];
        for (int k = 1; k < K; ++k) acc = fmaf(a[k * sa], b[k * sb], acc);
        return acc;
    case FMA_REV:
        acc = a[(K - 1) * sa] * b[(K - 1) * sb];
        for (int k = K - 2; k >= 0; --k) acc = fmaf(a[k * sa], b[k * sb], acc);
        return acc;
    case L2R_REV:
        acc = a[(K - 1) * sa] * b[(K - 1) * sb];
        for (int k = K - 2; k >= 0; --k) acc = acc + a[k * sa] * b[k * sb];
        return acc;
    case PAIRWISE: {
        float lo = a[0] * b[0], hi = 0.0f;
        if (K > 1) lo = lo + a[sa] * b[sb];
        if (K > 2) hi = a[2 * sa] * b[2 * sb];
        if (K > 3) hi = hi + a[3 * sa] * b[3 * sb];
        return K > 2 ? lo + hi : lo;
    }
    case FMA_PAIR: {
        float lo = a[0] * b[0], hi = 0.0f;
        if (K > 1) lo = fmaf(a[sa], b[sb], lo);
        if (K > 2) hi = a[2 * sa] * b[2 * sb];
        if (K > 3) hi = fmaf(a[3 * sa], b[3 * sb], hi);
        return K > 2 ? lo + hi : lo;
    }
    default: {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += (double)a[k * sa] * (double)b[k * sb];
        return (float)s;
    }
    }
}

/* C[n,i,j] = sum_k A[n,i,k] B[n or 0,k,j]; bstride = K*J for batched B, 0 for one shared B */
void probe_bmm(const float *A, const float *B, float *C, int64_t N, int I, int K, int J, int64_t bstride, int mode) {
    for (int64_t n = 0; n < N; ++n)
        for (int i = 0; i < I; ++i)
            for (int j = 0; j < J; ++j)
                C[(n * I + i) * J + j] = dot(A + (n * I + i) * K, 1, B + n * bstride + j, J, K, mode);
}
