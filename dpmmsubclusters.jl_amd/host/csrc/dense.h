// dense.h -- the two dense kernels of the master's per-distribution maths, laid out for the vector units of the host CPU.
//
//   factorisation      P = L' L     (L lower triangular; U = L' is the "reverse" Cholesky factor P = U U' that the Wishart
//                                    draw and logdet(psi) need, niw.jl:35,59)
//   triangular solve   L Y = A      (A lower-triangular Bartlett factor; R = Y' = A' U^-1 is the upper-triangular factor of a
//                                    Wishart(nu, (nu psi)^-1) draw, see hostmath.h)
//
// Both are D^3/6 multiply-adds per distribution, and 3K (+ up to K(K-1)/2 merge pairs) of them sit on the critical path
// between two GPU phases of every sweep.  Everything is row-major LOWER triangular so that every inner loop runs over a
// contiguous row prefix: no column walks, no reductions, no divisions in inner loops.  The bulk of the work is organised as
// rank-8 updates on two rows at a time (16 FMAs per 8 + 2 vector loads + 2 stores), which the compiler turns into straight
// AVX-512 / AVX2 FMA code (function multi-versioning: the build machine and the GPU box have different CPUs).
#pragma once
#include <math.h>
#include <string.h>

namespace dpmmh {

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && !defined(DPMMH_NO_CLONES)
#define DPMMH_CLONES __attribute__((target_clones("avx512f", "default")))
#else
#define DPMMH_CLONES
#endif

constexpr int kBlk = 8;

// y[0..n) -= a0*x0[0..n) + ... + a7*x7[0..n)   and the same for a second row: the rank-8, two-row update both kernels share.
// The sums are written as TREES (pairs, then pairs of pairs): a left-to-right chain of eight fused multiply-adds per row leaves the
// two FMA pipes of the core idle three cycles out of four (latency 4, two chains); four independent partial sums per row fill them.
#define DPMMH_RANK8_2ROWS(ya, yb, A, B, X0, X1, X2, X3, X4, X5, X6, X7, n)                                                    \
    _Pragma("omp simd") for (int k_ = 0; k_ < (n); ++k_) {                                                                  \
        const double v0 = (X0)[k_], v1 = (X1)[k_], v2 = (X2)[k_], v3 = (X3)[k_], v4 = (X4)[k_], v5 = (X5)[k_], v6 = (X6)[k_], v7 = (X7)[k_]; \
        (ya)[k_] -= (((A)[0] * v0 + (A)[1] * v1) + ((A)[2] * v2 + (A)[3] * v3)) + (((A)[4] * v4 + (A)[5] * v5) + ((A)[6] * v6 + (A)[7] * v7));    \
        (yb)[k_] -= (((B)[0] * v0 + (B)[1] * v1) + ((B)[2] * v2 + (B)[3] * v3)) + (((B)[4] * v4 + (B)[5] * v5) + ((B)[6] * v6 + (B)[7] * v7));    \
    }

// the same with sixteen vectors: halves the traffic of the updated rows (at D = 256 they stream from L2); four partial sums per row
#define DPMMH_RANK16_2ROWS(ya, yb, A, B, X, ldx, n)                                                                          \
    _Pragma("omp simd") for (int k_ = 0; k_ < (n); ++k_) {                                                                  \
        double sa0 = 0.0, sa1 = 0.0, sa2 = 0.0, sa3 = 0.0, sb0 = 0.0, sb1 = 0.0, sb2 = 0.0, sb3 = 0.0;                        \
        _Pragma("GCC unroll 4") for (int c_ = 0; c_ < 4; ++c_) {                                                            \
            const double u0 = (X)[(size_t)c_ * (ldx) + k_], u1 = (X)[(size_t)(c_ + 4) * (ldx) + k_];                          \
            const double u2 = (X)[(size_t)(c_ + 8) * (ldx) + k_], u3 = (X)[(size_t)(c_ + 12) * (ldx) + k_];                   \
            sa0 += (A)[c_] * u0; sa1 += (A)[c_ + 4] * u1; sa2 += (A)[c_ + 8] * u2; sa3 += (A)[c_ + 12] * u3;                  \
            sb0 += (B)[c_] * u0; sb1 += (B)[c_ + 4] * u1; sb2 += (B)[c_ + 8] * u2; sb3 += (B)[c_ + 12] * u3;                  \
        }                                                                                                                   \
        (ya)[k_] -= (sa0 + sa1) + (sa2 + sa3);                                                                              \
        (yb)[k_] -= (sb0 + sb1) + (sb2 + sb3);                                                                              \
    }

// Register-blocked form for long rows (the bulk of the work at D >= 128): FOUR rows x SIXTEEN columns of the updated block live in
// eight vector accumulators while the sixteen X rows stream past: 8 FMAs per 2 vector loads + 4 broadcasts, eight independent
// chains -- FMA-bound instead of load-bound (the two-row form issues one load per FMA).  GCC vector extensions: AVX-512 code in the
// avx512f clone, pairs of AVX2 operations in the default clone.  n must be a multiple of 16.
typedef double v8d __attribute__((vector_size(64), aligned(8)));
#define ld8(p) (*reinterpret_cast<const v8d *>(p))
#define st8(p, v) (*reinterpret_cast<v8d *>(p) = (v))
#define DPMMH_RANK16_4ROWS(y0, y1, y2, y3, A4 /* [4][16] */, X, ldx, n)                                                      \
    for (int q_ = 0; q_ < (n); q_ += 16) {                                                                                  \
        v8d c00 = ld8((y0) + q_), c01 = ld8((y0) + q_ + 8), c10 = ld8((y1) + q_), c11 = ld8((y1) + q_ + 8);                   \
        v8d c20 = ld8((y2) + q_), c21 = ld8((y2) + q_ + 8), c30 = ld8((y3) + q_), c31 = ld8((y3) + q_ + 8);                   \
        _Pragma("GCC unroll 16") for (int c_ = 0; c_ < 16; ++c_) {                                                          \
            const v8d x0 = ld8((X) + (size_t)c_ * (ldx) + q_), x1 = ld8((X) + (size_t)c_ * (ldx) + q_ + 8);                  \
            const double a0 = (A4)[c_], a1 = (A4)[16 + c_], a2 = (A4)[32 + c_], a3 = (A4)[48 + c_];                           \
            c00 -= a0 * x0; c01 -= a0 * x1; c10 -= a1 * x0; c11 -= a1 * x1;                                                 \
            c20 -= a2 * x0; c21 -= a2 * x1; c30 -= a3 * x0; c31 -= a3 * x1;                                                 \
        }                                                                                                                   \
        st8((y0) + q_, c00); st8((y0) + q_ + 8, c01); st8((y1) + q_, c10); st8((y1) + q_ + 8, c11);                           \
        st8((y2) + q_, c20); st8((y2) + q_ + 8, c21); st8((y3) + q_, c30); st8((y3) + q_ + 8, c31);                           \
    }

// P (row-major D x D, LOWER triangle valid, destroyed) = L' L with L lower triangular.  L (row-major D x D) receives the factor
// (entries above the diagonal are set to zero) -- or L == nullptr: only the log-determinant is wanted and the factor stays in P's
// lower triangle.  Returns log det P = 2 sum log L_jj, NaN when P is not positive definite.
DPMMH_CLONES inline double chol_ltl(double *__restrict__ P, int D, double *__restrict__ Lout) {
    double ld = 0.0;
    bool ok = true;
    // rows j1-1 .. j0 of the factor, in place: pending updates of the block's own later rows, then scale by 1 / diagonal
    auto factor_rows = [&](int j0, int j1) {
        for (int j = j1 - 1; j >= j0 && ok; --j) {
            double *row = P + (size_t)j * D;
            for (int c = j + 1; c < j1; ++c) {
                const double *w = P + (size_t)c * D;
                const double a = w[j];
#pragma omp simd
                for (int k = 0; k <= j; ++k) row[k] -= a * w[k];
            }
            const double s = row[j];
            if (!(s > 0.0)) { ok = false; return; }
            const double d = sqrt(s), inv = 1.0 / d;
            ld += log(d);
#pragma omp simd
            for (int k = 0; k < j; ++k) row[k] *= inv;
            row[j] = d;
        }
    };
    // rows k in [klo, khi):  P[k][0..k] -= sum_{c < nw} W[c][k] * W[c][0..k]  with the nw (<= 8) factor rows W = P + w0 * D
    auto trail8 = [&](int w0r, int nw, int klo, int khi) {
        if (nw == kBlk) {
            const double *w0 = P + (size_t)w0r * D, *w1 = w0 + D, *w2 = w1 + D, *w3 = w2 + D, *w4 = w3 + D, *w5 = w4 + D, *w6 = w5 + D, *w7 = w6 + D;
            int k = klo;
            for (; k + 1 < khi; k += 2) {
                double *ya = P + (size_t)k * D, *yb = ya + D;
                const double a[kBlk] = {w0[k], w1[k], w2[k], w3[k], w4[k], w5[k], w6[k], w7[k]};
                const double b[kBlk] = {w0[k + 1], w1[k + 1], w2[k + 1], w3[k + 1], w4[k + 1], w5[k + 1], w6[k + 1], w7[k + 1]};
                DPMMH_RANK8_2ROWS(ya, yb, a, b, w0, w1, w2, w3, w4, w5, w6, w7, k + 2)     // row k's entry at column k+1 is scratch
            }
            if (k < khi) {
                double *ya = P + (size_t)k * D;
                const double a[kBlk] = {w0[k], w1[k], w2[k], w3[k], w4[k], w5[k], w6[k], w7[k]};
#pragma omp simd
                for (int q = 0; q <= k; ++q)
                    ya[q] -= a[0] * w0[q] + a[1] * w1[q] + a[2] * w2[q] + a[3] * w3[q] + a[4] * w4[q] + a[5] * w5[q] + a[6] * w6[q] + a[7] * w7[q];
            }
        } else {
            for (int c = 0; c < nw; ++c) {
                const double *w = P + (size_t)(w0r + c) * D;
                for (int k = klo; k < khi; ++k) {
                    double *y = P + (size_t)k * D;
                    const double a = w[k];
#pragma omp simd
                    for (int q = 0; q <= k; ++q) y[q] -= a * w[q];
                }
            }
        }
    };
    int j1 = D;
    while (j1 > 0 && ok) {
        if (j1 >= 2 * kBlk) {
            // sixteen rows: [j1-8, j1) first (its update reaches the other eight at once), then [j1-16, j1-8); all rows above get ONE
            // rank-16 update -- half the read-modify-write traffic of two rank-8 sweeps over the trailing triangle
            const int jm = j1 - kBlk, j0 = j1 - 2 * kBlk;
            factor_rows(jm, j1);
            if (!ok) break;
            trail8(jm, kBlk, j0, jm);
            factor_rows(j0, jm);
            if (!ok) break;
            const double *W = P + (size_t)j0 * D;
            int k = 0;
            if (D % 16 == 0) {
                // four rows at a time over columns 0 .. roundup(k + 4, 16): entries beyond a row's diagonal are scratch
                for (; k + 3 < j0; k += 4) {
                    double *ya = P + (size_t)k * D;
                    double a4[64];
                    for (int c = 0; c < 16; ++c) {
                        const double *w = W + (size_t)c * D + k;
                        a4[c] = w[0]; a4[16 + c] = w[1]; a4[32 + c] = w[2]; a4[48 + c] = w[3];
                    }
                    const int n16 = (k + 4 + 15) & ~15;
                    DPMMH_RANK16_4ROWS(ya, ya + D, ya + 2 * (size_t)D, ya + 3 * (size_t)D, a4, W, D, n16)
                }
            }
            for (; k + 1 < j0; k += 2) {
                double *ya = P + (size_t)k * D, *yb = ya + D;
                double a[16], b[16];
                for (int c = 0; c < 16; ++c) { a[c] = W[(size_t)c * D + k]; b[c] = W[(size_t)c * D + k + 1]; }
                DPMMH_RANK16_2ROWS(ya, yb, a, b, W, D, k + 2)
            }
            if (k < j0) {
                double *ya = P + (size_t)k * D;
                double a[16];
                for (int c = 0; c < 16; ++c) a[c] = W[(size_t)c * D + k];
#pragma omp simd
                for (int q = 0; q <= k; ++q) {
                    double sa = 0.0;
                    for (int c = 0; c < 16; ++c) sa += a[c] * W[(size_t)c * D + q];
                    ya[q] -= sa;
                }
            }
            j1 = j0;
        } else {
            const int j0 = j1 > kBlk ? j1 - kBlk : 0;
            factor_rows(j0, j1);
            if (!ok) break;
            if (j0 > 0) trail8(j0, j1 - j0, 0, j0);
            j1 = j0;
        }
    }
    if (!ok) return NAN;
    if (Lout) {
        for (int j = 0; j < D; ++j) {
            memcpy(Lout + (size_t)j * D, P + (size_t)j * D, sizeof(double) * (j + 1));
            memset(Lout + (size_t)j * D + j + 1, 0, sizeof(double) * (D - 1 - j));
        }
    }
    return 2.0 * ld;
}

// L Y = A for lower-triangular L (row-major D x D) and a lower-triangular right-hand side: on entry Y (row-major D x D) holds A
// with ZEROS above the diagonal, on exit the (lower-triangular) solution.  Row i needs rows k < i of Y: earlier blocks of eight
// rows enter as rank-8 updates on two rows at a time, the block's own rows as plain updates.
DPMMH_CLONES inline void solve_lower_left(double *__restrict__ Y, const double *__restrict__ L, int D) {
    for (int i0 = 0; i0 < D; i0 += kBlk) {
        const int i1 = i0 + kBlk < D ? i0 + kBlk : D;
        // contributions of all earlier row blocks [k0, k0 + 8), k0 < i0 (full blocks: i0 is a multiple of 8)
        int k0 = 0;
        for (; k0 + 2 * kBlk <= i0; k0 += 2 * kBlk) {      // two earlier blocks at a time: rank-16 (rows k0..k0+15 are zero beyond column k0+15)
            const double *X = Y + (size_t)k0 * D;
            const int len = k0 + 2 * kBlk;
            int i = i0;
            for (; i + 3 < i1; i += 4) {      // len is a multiple of 16
                double *ya = Y + (size_t)i * D;
                double a4[64];
                for (int r = 0; r < 4; ++r) memcpy(a4 + 16 * r, L + (size_t)(i + r) * D + k0, sizeof(double) * 16);
                DPMMH_RANK16_4ROWS(ya, ya + D, ya + 2 * (size_t)D, ya + 3 * (size_t)D, a4, X, D, len)
            }
            for (; i + 1 < i1; i += 2) {
                double *ya = Y + (size_t)i * D, *yb = ya + D;
                const double *a = L + (size_t)i * D + k0, *b = a + D;
                DPMMH_RANK16_2ROWS(ya, yb, a, b, X, D, len)
            }
            if (i < i1) {
                double *ya = Y + (size_t)i * D;
                const double *a = L + (size_t)i * D + k0;
#pragma omp simd
                for (int q = 0; q < len; ++q) {
                    double sa = 0.0;
                    for (int c = 0; c < 16; ++c) sa += a[c] * X[(size_t)c * D + q];
                    ya[q] -= sa;
                }
            }
        }
        for (; k0 < i0; k0 += kBlk) {
            const double *y0 = Y + (size_t)k0 * D, *y1 = y0 + D, *y2 = y1 + D, *y3 = y2 + D, *y4 = y3 + D, *y5 = y4 + D, *y6 = y5 + D, *y7 = y6 + D;
            const int len = k0 + kBlk;                    // rows k0..k0+7 of Y are zero beyond their diagonals
            int i = i0;
            for (; i + 1 < i1; i += 2) {
                double *ya = Y + (size_t)i * D, *yb = ya + D;
                const double *a = L + (size_t)i * D + k0, *b = a + D;
                DPMMH_RANK8_2ROWS(ya, yb, a, b, y0, y1, y2, y3, y4, y5, y6, y7, len)
            }
            if (i < i1) {
                double *ya = Y + (size_t)i * D;
                const double *a = L + (size_t)i * D + k0;
#pragma omp simd
                for (int q = 0; q < len; ++q)
                    ya[q] -= a[0] * y0[q] + a[1] * y1[q] + a[2] * y2[q] + a[3] * y3[q] + a[4] * y4[q] + a[5] * y5[q] + a[6] * y6[q] + a[7] * y7[q];
            }
        }
        // the block's own rows
        for (int i = i0; i < i1; ++i) {
            double *yi = Y + (size_t)i * D;
            const double *li = L + (size_t)i * D;
            for (int k = i0; k < i; ++k) {
                const double a = li[k];
                const double *yk = Y + (size_t)k * D;
#pragma omp simd
                for (int q = 0; q <= k; ++q) yi[q] -= a * yk[q];
            }
            const double inv = 1.0 / li[i];
#pragma omp simd
            for (int q = 0; q <= i; ++q) yi[q] *= inv;
        }
    }
}

}  // namespace dpmmh
