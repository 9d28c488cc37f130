// niw_master.hip -- the dense per-distribution maths of the master, on the device (NIW prior).
//
// Stands in for the master's calc_posterior (src/priors/niw.jl:20-31), the factorisation behind sample_distribution and
// log_marginal_likelihood (niw.jl:33-40, 53-62) and the Wishart / Normal draws of sample_cluster_params
// (src/shared_actions.jl:41-49) for all 3K distributions of a sweep.  At D = 256 the host needs 2-3 ms per sweep for them
// (96 factorisations and triangular solves of 256 x 256, DRAM-bound on 14 cores) plus 30 MB over the host link (rows up,
// parameters down); here the packed statistics rows never leave HBM and the parameters are born in the sweep kernels' layout.
//
// Everything is Float64, row-major, LOWER triangular, padded to DP = 16 * ceil(D / 16) (identity in the padding):
//   form    P = nu' psi' from the packed rows {N, sum x, lower triangle of sum x x'} and the prior (same formula, same order
//           of operations as hostmath.h niw_posterior_packed)
//   factor  P = L' L in place (L lower; U = L' is the "reverse" Cholesky factor P = U U' that the draw needs), blocked by 16:
//           diagonal block -> panel (one thread per column) -> rank-16 trailing update in 4 x 4 register tiles
//   draw    Bartlett factor A (chi on the diagonal, standard normals below) from the counter-based generator, L Y = A blocked
//           by 16 rows (4 x 4 register tiles for the products, one thread per column for the 16 x 16 triangular part),
//           R = Y' is the upper-triangular factor of a Wishart(nu', P^-1) draw; mu = m' + R^-1 xi / sqrt(kappa')
//   pack    R, mu, additive constants -> the fragment images of the sweep kernels (no host staging, no copy)
// One workgroup of 256 threads per distribution; plain FP64 vector arithmetic (5.6 Mflop per factorisation at D = 256:
// synchronisation, not arithmetic, is what a 256 x 256 problem costs).  Measured and dropped: 512 threads per workgroup (factorisation
// 5 % faster, draws 30 % slower), 1024 (half the register budget: spills), an explicit one-step prefetch of the Y rows in the solve's
// product loop (+7 %: the register hand-over waits for the loads it was meant to overlap).
#include <algorithm>
#include "dpmm_device.h"
#include "dpmm_kernels.h"

namespace dpmm {

enum : uint32_t { STREAM_M_NORMAL = 32, STREAM_M_CHI = 33, STREAM_M_XI = 34 };

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {          // (0, 1), 53 bits
    return ((double)((((uint64_t)a << 32) | b) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double normal_from(const Philox4 &r) {        // one standard normal from one generator block
    const double u1 = u53(r.v[0], r.v[1]), u2 = u53(r.v[2], r.v[3]);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
// Marsaglia-Tsang Gamma(a, 1), a >= 1; trial t of element `idx` uses generator blocks (idx, 2t) and (idx, 2t + 1)
__device__ double gamma_mt(double a, uint64_t seed, uint64_t idx, uint32_t epoch) {
    if (a < 1.0) {      // Gamma(a) = Gamma(a + 1) U^(1/a)
        const Philox4 ru = philox4x32_10(seed, idx * 64 + 63, epoch, STREAM_M_CHI);
        return gamma_mt(a + 1.0, seed, idx, epoch) * pow(u53(ru.v[0], ru.v[1]), 1.0 / a);
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0;; ++t) {
        const double x = normal_from(philox4x32_10(seed, idx * 64 + 2 * t, epoch, STREAM_M_CHI));
        const Philox4 ru = philox4x32_10(seed, idx * 64 + 2 * t + 1, epoch, STREAM_M_CHI);
        const double u = u53(ru.v[0], ru.v[1]);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (u < 1.0 - 0.0331 * x * x * x * x) return d * v;
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return d * v;
        if (t > 60) return d * v;      // unreachable in practice (acceptance > 95 %): bounded for safety
    }
}

// ------------------------------------------------------------------------------------------------------------------ form
// job j: cluster k = jobs[2j] (0-based), slot s = jobs[2j+1]; rows 3s + w, w = 0 (cluster = left + right), 1 (left), 2 (right).
// small[(3j + w) * 4 + {0,1,2}] = N, kappa', nu'   (entry 3: log det(nu' psi'), written by the factorisation)
__global__ __launch_bounds__(256) void niw_form_kernel(NiwMasterArgs A, const int32_t *__restrict__ jobs, const double *__restrict__ rows,
                                                       double *__restrict__ small) {
    const int j = blockIdx.x / 3, w = blockIdx.x % 3;
    const int k = jobs[2 * j], s = jobs[2 * j + 1];
    const int D = A.D, DP = A.DP;
    const int64_t stride = A.packed_stride;
    const double *l = rows + (int64_t)(2 * k) * stride, *r = l + stride;
    const double cl = (w != 2) ? 1.0 : 0.0, cr = (w != 1) ? 1.0 : 0.0;
    const int row = 3 * s + w;
    double *P = A.fac + (int64_t)row * DP * DP;
    double *m = A.mean + (int64_t)row * DP;
    __shared__ double sm[DPMM_MASTER_MAXD], sm0[DPMM_MASTER_MAXD];
    const double N = cl * l[0] + cr * r[0];
    const double k0 = A.kappa0, v0 = A.nu0, k1 = k0 + N, v1 = v0 + N;
    if (w == 1 && blockIdx.y == 0) {      // keep the statistics of the slot (what the host keeps in its packed rows)
        double *dst = A.rows_store + (int64_t)(2 * s) * stride;
        for (int64_t e = threadIdx.x; e < 2 * stride; e += blockDim.x) dst[e] = l[e];
    }
    for (int a = threadIdx.x; a < DP; a += blockDim.x) {
        const double m0 = a < D ? A.m0[a] : 0.0;
        double mv = m0;
        if (N != 0.0 && a < D) mv = (m0 * k0 + (cl * l[1 + a] + cr * r[1 + a])) / k1;
        sm[a] = mv; sm0[a] = m0;
        if (blockIdx.y == 0) m[a] = mv;
    }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        double *o = small + (int64_t)(3 * j + w) * 4;
        o[0] = N; o[1] = (N == 0.0) ? k0 : k1; o[2] = (N == 0.0) ? v0 : v1;
        A.kap[row] = o[1]; A.nu[row] = o[2];
    }
    const double *tl = l + 1 + D, *tr = r + 1 + D;
    // rows a = blockIdx.y, blockIdx.y + gridDim.y, ... of the scale matrix; one thread per column
    for (int a = blockIdx.y; a < DP; a += gridDim.y)
    for (int b = threadIdx.x; b < DP; b += blockDim.x) {
        const int64_t e = (int64_t)a * DP + b;
        double v = 0.0;
        if (b <= a) {
            if (a >= D) v = (a == b) ? 1.0 : 0.0;                   // padding: identity
            else {
                const int64_t t = (int64_t)a * (a + 1) / 2 + b;
                const double pab = A.psi_lo[t];
                if (N == 0.0) v = pab * v0;
                else {
                    const double sab = cl * tl[t] + cr * tr[t];
                    v = ((v0 * pab + (k0 * sm0[a]) * sm0[b] - (k1 * sm[a]) * sm[b] + sab) / v1) * v1;     // psi' then nu' psi' (niw.jl:29,35)
                }
            }
        }
        P[e] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------- factor
// P (row-major DP x DP, lower triangle) -> L with P = L' L, in place; small[.. + 3] = log det P = 2 sum log L_jj (NaN when a
// pivot is not positive).  One workgroup per job row (blockIdx.x = 3 j + w).
__global__ __launch_bounds__(256) void niw_chol_kernel(NiwMasterArgs A, const int32_t *__restrict__ jobs, double *__restrict__ small) {
    const int j = blockIdx.x / 3, w = blockIdx.x % 3;
    const int row = jobs ? 3 * jobs[2 * j + 1] + w : (int)blockIdx.x;      // jobs == nullptr: matrix blockIdx.x of A.fac (pair scratch)
    const int DP = A.DP, NB = DP / 16, tid = threadIdx.x;
    double *P = A.fac + (int64_t)row * DP * DP;
    extern __shared__ double lds[];
    double *Wp = lds;                    // [16][DP]  factor rows of the current block, columns 0 .. j0 + 15
    double *Dg = lds + 16 * DP;          // [16][17]  diagonal block
    __shared__ double s_ld;
    __shared__ int s_bad;
    if (tid == 0) { s_ld = 0.0; s_bad = 0; }
    __syncthreads();
    for (int jb = NB - 1; jb >= 0; --jb) {
        const int j0 = 16 * jb;
        // (1) diagonal block -> LDS, factorised from its last row up by the first 16 threads
        { const int a = tid >> 4, b = tid & 15; Dg[a * 17 + b] = (b <= a) ? P[(int64_t)(j0 + a) * DP + j0 + b] : 0.0; }
        __syncthreads();
        if (tid < 16) {
            const int kcol = tid;
            for (int jj = 15; jj >= 0; --jj) {
                double v = Dg[jj * 17 + kcol];
                for (int c = jj + 1; c < 16; ++c) v -= Dg[c * 17 + jj] * Dg[c * 17 + kcol];
                const double piv = __shfl(v, jj, 16);        // lane jj holds the pivot; everybody needs its square root
                const double inv = rsqrt(piv), d = piv * inv;      // (no division in the 16-step dependent chain)
                if (!(piv > 0.0) && kcol == 0) s_bad = 1;
                if (kcol < jj) Dg[jj * 17 + kcol] = v * inv;
                else if (kcol == jj) Dg[jj * 17 + jj] = d;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // row jj is read by the next rows (other lanes wrote it)
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        if (tid < 16) {
            const double d = Dg[tid * 17 + tid];
            Dg[tid * 17 + 16] = 1.0 / d;          // reciprocal pivots (the unused 17th column): the panel multiplies instead of dividing
            double lg = log(d);
            for (int o = 8; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 16);
            if (tid == 0) s_ld += lg;
        }
        __syncthreads();
        { const int a = tid >> 4, b = tid & 15; if (b <= a) P[(int64_t)(j0 + a) * DP + j0 + b] = Dg[a * 17 + b]; Wp[a * DP + j0 + b] = (b <= a) ? Dg[a * 17 + b] : 0.0; }
        // (2) panel: columns q < j0, one thread per column: L[j][q] = (P[j][q] - sum_{c > j} L[c][j] L[c][q]) / L[j][j]
        for (int q = tid; q < j0; q += 256) {
            double wv[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) wv[jj] = P[(int64_t)(j0 + jj) * DP + q];
#pragma unroll
            for (int jj = 15; jj >= 0; --jj) {
                double v = wv[jj];
#pragma unroll
                for (int c = jj + 1; c < 16; ++c) v -= Dg[c * 17 + jj] * wv[c];
                wv[jj] = v * Dg[jj * 17 + 16];
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) { P[(int64_t)(j0 + jj) * DP + q] = wv[jj]; Wp[jj * DP + q] = wv[jj]; }
        }
        __syncthreads();
        // (3) trailing update of rows / columns < j0: P[k][q] -= sum_c W[c][k] W[c][q], q <= k, in 4 x 4 tiles
        const int nt = j0 / 4;
        for (int t = tid; t < nt * (nt + 1) / 2; t += 256) {
            int tk = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while ((tk + 1) * (tk + 2) / 2 <= t) ++tk;
            while (tk * (tk + 1) / 2 > t) --tk;
            const int tq = t - tk * (tk + 1) / 2;
            const int kk = 4 * tk, qq = 4 * tq;
            // the tile of P is requested BEFORE the products (its latency hides behind them); entries above the diagonal of a
            // diagonal tile are read and written back unchanged (scratch)
            double acc[4][4], pv[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { acc[a][b] = 0.0; pv[a][b] = P[(int64_t)(kk + a) * DP + qq + b]; }
#pragma unroll 4
            for (int c = 0; c < 16; ++c) {
                const double *wr = Wp + c * DP;
                const double a0 = wr[kk], a1 = wr[kk + 1], a2 = wr[kk + 2], a3 = wr[kk + 3];
                const double b0 = wr[qq], b1 = wr[qq + 1], b2 = wr[qq + 2], b3 = wr[qq + 3];
                acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[0][2] += a0 * b2; acc[0][3] += a0 * b3;
                acc[1][0] += a1 * b0; acc[1][1] += a1 * b1; acc[1][2] += a1 * b2; acc[1][3] += a1 * b3;
                acc[2][0] += a2 * b0; acc[2][1] += a2 * b1; acc[2][2] += a2 * b2; acc[2][3] += a2 * b3;
                acc[3][0] += a3 * b0; acc[3][1] += a3 * b1; acc[3][2] += a3 * b2; acc[3][3] += a3 * b3;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    P[(int64_t)(kk + a) * DP + qq + b] = pv[a][b] - acc[a][b];
        }
        __syncthreads();
    }
    if (tid == 0) small[(int64_t)blockIdx.x * 4 + 3] = s_bad ? NAN : 2.0 * s_ld;
}

// ------------------------------------------------------------------------------------------------------------------ draw
// One workgroup per distribution (blockIdx.x = 3 k + w in CLUSTER order; slot from slot_of_cluster).  Y (scratch, [3K][DP][DP]).
__global__ __launch_bounds__(256) void niw_draw_kernel(NiwMasterArgs A, const int32_t *__restrict__ slot_of_cluster, uint32_t epoch,
                                                       double *__restrict__ Yall, float *__restrict__ logdet_sigma) {
    const int k = blockIdx.x / 3, w = blockIdx.x % 3;
    const int row = 3 * slot_of_cluster[k] + w;
    const int D = A.D, DP = A.DP, NB = DP / 16, tid = threadIdx.x;
    const double *L = A.fac + (int64_t)row * DP * DP;
    double *Y = Yall + (int64_t)blockIdx.x * DP * DP;
    const double nu = A.nu[row], kap = A.kap[row];
    const uint64_t id = (uint64_t)blockIdx.x;          // the streams are keyed by the position in cluster order, like the host's
    extern __shared__ double lds[];
    double *T = lds;                     // [16][DP]
    double *Ld = lds + 16 * DP;          // [16][17]
    double *xi = Ld + 16 * 17;           // [DP]
    double *Lp = xi + DP;                // [16][DP]  rows i0 .. i0 + 15 of L, columns < i0
    // Bartlett factor (lower): chi on the diagonal, standard normals below, identity in the padding.  One generator block gives
    // the two normals of an element pair (2p, 2p + 1) of a row (Box-Muller, both branches).
    const int HP = DP / 2;
    for (int p2 = tid; p2 < DP * HP; p2 += 256) {
        const int r = p2 / HP, c = 2 * (p2 - r * HP);
        double n0 = 0.0, n1 = 0.0;
        if (r >= D) { n0 = (c == r) ? 1.0 : 0.0; n1 = (c + 1 == r) ? 1.0 : 0.0; }
        else if (c < r) {
            const Philox4 g = philox4x32_10(A.seed, (id << 32) + (uint64_t)r * DP + c, epoch, STREAM_M_NORMAL);
            const double u1 = u53(g.v[0], g.v[1]), u2 = u53(g.v[2], g.v[3]);
            const double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincos(6.283185307179586476925 * u2, &sn, &cs);
            n0 = rad * cs;
            n1 = (c + 1 < r) ? rad * sn : 0.0;
        }
        Y[(int64_t)r * DP + c] = n0; Y[(int64_t)r * DP + c + 1] = n1;
    }
    __syncthreads();
    for (int r = tid; r < D; r += 256) Y[(int64_t)r * DP + r] = sqrt(2.0 * gamma_mt(0.5 * (nu - r), A.seed, (id << 16) + (uint64_t)r, epoch));
    for (int d = tid; d < DP; d += 256)
        xi[d] = d < D ? normal_from(philox4x32_10(A.seed, (id << 32) + (uint64_t)d, epoch, STREAM_M_XI)) : 0.0;
    __syncthreads();
    // L Y = A, block row by block row
    for (int ib = 0; ib < NB; ++ib) {
        const int i0 = 16 * ib, ncol = i0 + 16;             // columns 0 .. i0 + 15 of this block row can be non-zero
        // the block row's part of L (columns < i0) -> LDS once: the products below read it 4 values per inner step
        for (int kk = tid; kk < i0; kk += 256) {          // (i0 <= 240: one trip; the 16 loads of a thread are issued together)
            double lrow[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) lrow[a] = L[(int64_t)(i0 + a) * DP + kk];
#pragma unroll
            for (int a = 0; a < 16; ++a) Lp[a * DP + kk] = lrow[a];
        }
        { const int a = tid >> 4, b = tid & 15; Ld[a * 17 + b] = (b <= a) ? L[(int64_t)(i0 + a) * DP + i0 + b] : 0.0; }
        if (tid < 16) Ld[tid * 17 + 16] = 1.0 / L[(int64_t)(i0 + tid) * DP + i0 + tid];        // reciprocal pivots (17th column)
        __syncthreads();
        // T = A(ib, :) - L(ib, < i0) Y(< i0, :)   in 4 x 4 tiles: 4 row groups x ncol / 4 column groups
        const int ntile = 4 * (ncol / 4);
        for (int t = tid; t < ntile; t += 256) {
            const int rg = t & 3, cg = t >> 2;
            const int r0 = i0 + 4 * rg, q0 = 4 * cg;
            double acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = Y[(int64_t)(r0 + a) * DP + q0 + b];
            // Y[kk][q] is zero for q > kk: start the inner dimension at the first row that reaches column q0 (a multiple of 4, as
            // is i0: the loop runs in steps of four with all loads of a step issued together)
            const double *lp0 = Lp + (4 * rg) * DP;
            int kk = q0;
            for (; kk < i0; kk += 4) {
                double yv[4][4], lv[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double *yk = Y + (int64_t)(kk + u) * DP + q0;
                    yv[u][0] = yk[0]; yv[u][1] = yk[1]; yv[u][2] = yk[2]; yv[u][3] = yk[3];
#pragma unroll
                    for (int a = 0; a < 4; ++a) lv[u][a] = lp0[a * DP + kk + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] -= lv[u][a] * yv[u][b];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) T[(4 * rg + a) * DP + q0 + b] = acc[a][b];
        }
        __syncthreads();
        // 16 x 16 triangular part: one thread per column
        for (int q = tid; q < ncol; q += 256) {
            double y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double v = T[r * DP + q];
#pragma unroll
                for (int c = 0; c < r; ++c) v -= Ld[r * 17 + c] * y[c];
                y[r] = v * Ld[r * 17 + 16];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[(int64_t)(i0 + r) * DP + q] = (q <= i0 + r) ? y[r] : 0.0;
        }
        __syncthreads();
    }
    // log det Sigma = -2 sum log Y_jj (R = Y', Sigma^-1 = R' R)
    if (tid < 64) {
        double lg = 0.0;
        for (int d = tid; d < D; d += 64) lg += log(Y[(int64_t)d * DP + d]);
        for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o);
        if (tid == 0) logdet_sigma[blockIdx.x] = (float)(-2.0 * lg);
    }
    // mu = m' + R^-1 xi / sqrt(kappa'):  R v = xi by back substitution in the column (axpy) form (column c of R = row c of Y), blocked by
    // 16: the 16 x 16 diagonal block is solved from LDS by 16 lanes, the update of the remaining right-hand side is one coalesced pass of
    // all threads over the block's 16 rows of Y (D sequential steps with a global load in each cost 90 us at D = 256)
    double *vv = T;                                  // T is free now: v [DP]
    __syncthreads();
    for (int cb = NB - 1; cb >= 0; --cb) {
        const int c0 = 16 * cb;
        { const int a2 = tid >> 4, b2 = tid & 15; Ld[a2 * 17 + b2] = (b2 <= a2) ? Y[(int64_t)(c0 + a2) * DP + c0 + b2] : 0.0; }
        __syncthreads();
        if (tid < 16) {
            double xb = xi[c0 + tid];
            const double inv = 1.0 / Ld[tid * 17 + tid];
            for (int a2 = 15; a2 >= 0; --a2) {
                const double va = __shfl(xb, a2, 16) * __shfl(inv, a2, 16);
                if (tid == a2) vv[c0 + a2] = va;
                if (tid < a2) xb -= va * Ld[a2 * 17 + tid];
            }
        }
        __syncthreads();
        for (int r = tid; r < c0; r += 256) {
            double acc = xi[r];
#pragma unroll
            for (int a2 = 0; a2 < 16; ++a2) acc -= vv[c0 + a2] * Y[(int64_t)(c0 + a2) * DP + r];
            xi[r] = acc;
        }
        __syncthreads();
    }
    const double isk = 1.0 / sqrt(kap);
    const double *m = A.mean + (int64_t)row * DP;
    float *mu_out = A.mu_draw + (int64_t)blockIdx.x * DP;
    for (int d = tid; d < DP; d += 256) mu_out[d] = d < D ? (float)(m[d] + vv[d] * isk) : 0.f;
}

// ------------------------------------------------------------------------------------------------------------------ pack
// Y ([3K][DP][DP] Float64, R = Y') + mu -> the sweep kernels' fragment images (same layout as niw_pack_kernel), constants, tail records
__global__ void niw_master_pack_kernel(const double *__restrict__ Yall, const float *__restrict__ mu_draw, const float *__restrict__ logdet_sigma,
                                       const float *__restrict__ lr, const float *__restrict__ wts, float *__restrict__ Rp,
                                       float *__restrict__ mup, float *__restrict__ cst, float *__restrict__ tail, int D, int DPm, int NB,
                                       int nmat, unsigned long long *__restrict__ work) {
    const int NP = NB * (NB + 1) / 2;
    const int DP = 16 * NB;                   // the sweep kernels' padded dimension (>= DPm, the master's)
    if (work && blockIdx.x == 0 && threadIdx.x < 8) work[threadIdx.x] = 0ull;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nmat; e += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(e / 3), w = (int)(e % 3);
        const float lw = w == 0 ? logf(wts[k]) : logf(lr[2 * k + (w - 1)]);
        cst[e] = -0.5f * logdet_sigma[e] + lw;
    }
    const int64_t total = (int64_t)nmat * NP * 256;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int jj = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        const int64_t pj = e >> 8;
        const int pair = (int)(pj % NP);
        const int j = (int)(pj / NP);
        int bi = 0, rem = pair;
        while (rem >= NB - bi) { rem -= NB - bi; ++bi; }
        const int t = bi + rem;
        const int row = 16 * bi + (lane & 15);
        const int col = 16 * t + 4 * (lane >> 4) + jj;
        float v = 0.f;
        if (row < D && col < D && col >= row) v = (float)Yall[(int64_t)j * DPm * DPm + (int64_t)col * DPm + row];
        Rp[e] = v;
    }
    const int64_t totmu = (int64_t)nmat * DP;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < totmu; e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % DP);
        const int64_t j = e / DP;
        mup[e] = d < D ? mu_draw[j * DPm + d] : 0.f;          // the sweep pads to 16 NB (NB in {1, 2, 4, 8, 16}), the master to 16 ceil(D / 16)
    }
    if (tail && D >= 4) {
        const int f0 = D - 4, K = nmat / 3, NPR = (K + 1) / 2;
        for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)NPR * 32; e += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)(e & 1), q = (int)((e >> 1) & 15), k = 2 * (int)(e >> 5) + c;
            const int64_t j = k < K ? 3 * k : 0;
            const int tr[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, tc[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
            float v = 0.f;
            if (k >= K) v = (q == 14) ? -INFINITY : 0.f;
            else if (q < 10) v = (float)Yall[j * DPm * DPm + (int64_t)(f0 + tc[q]) * DPm + (f0 + tr[q])];
            else if (q < 14) v = mu_draw[j * DPm + f0 + (q - 10)];
            else if (q == 14) {
                v = -0.5f * logdet_sigma[j] + logf(wts[k]);
            }
            tail[e] = v;
        }
    }
}

// Pooled statistics of cluster pairs for the merge proposals (shared_actions.jl:21-27): job p pools the four stored rows of slots
// pairs[2p], pairs[2p+1]; P = nu' psi' of the pooled set -> scratch matrix p; small[4p + {0,1,2}] = N, kappa', nu'.
__global__ __launch_bounds__(256) void niw_form_pair_kernel(NiwMasterArgs A, const int32_t *__restrict__ pairs, double *__restrict__ scratch,
                                                            double *__restrict__ small) {
    const int p = blockIdx.x;
    const int D = A.D, DP = A.DP;
    const int64_t stride = A.packed_stride;
    const double *r0 = A.rows_store + (int64_t)(2 * pairs[2 * p]) * stride, *r1 = r0 + stride;
    const double *r2 = A.rows_store + (int64_t)(2 * pairs[2 * p + 1]) * stride, *r3 = r2 + stride;
    double *P = scratch + (int64_t)p * DP * DP;
    __shared__ double sm[DPMM_MASTER_MAXD], sm0[DPMM_MASTER_MAXD];
    const double N = r0[0] + r1[0] + r2[0] + r3[0];
    const double k0 = A.kappa0, v0 = A.nu0, k1 = k0 + N, v1 = v0 + N;
    for (int a = threadIdx.x; a < DP; a += blockDim.x) {
        const double m0 = a < D ? A.m0[a] : 0.0;
        double mv = m0;
        if (N != 0.0 && a < D) mv = (m0 * k0 + (r0[1 + a] + r1[1 + a] + r2[1 + a] + r3[1 + a])) / k1;
        sm[a] = mv; sm0[a] = m0;
    }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        double *o = small + (int64_t)p * 4;
        o[0] = N; o[1] = (N == 0.0) ? k0 : k1; o[2] = (N == 0.0) ? v0 : v1;
    }
    for (int a = blockIdx.y; a < DP; a += gridDim.y)
        for (int b = threadIdx.x; b < DP; b += blockDim.x) {
            double v = 0.0;
            if (b <= a) {
                if (a >= D) v = (a == b) ? 1.0 : 0.0;
                else {
                    const int64_t t = (int64_t)a * (a + 1) / 2 + b;
                    const double pab = A.psi_lo[t];
                    if (N == 0.0) v = pab * v0;
                    else {
                        const int64_t tt = 1 + D + t;
                        const double sab = r0[tt] + r1[tt] + r2[tt] + r3[tt];
                        v = ((v0 * pab + (k0 * sm0[a]) * sm0[b] - (k1 * sm[a]) * sm[b] + sab) / v1) * v1;
                    }
                }
            }
            P[(int64_t)a * DP + b] = v;
        }
}
hipError_t launch_niw_master_pairs(const NiwMasterArgs &a, const int32_t *pairs, int n, double *scratch, double *small, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(niw_form_pair_kernel, dim3(n, a.DP >= 64 ? 16 : 1), dim3(256), 0, s, a, pairs, scratch, small);
    NiwMasterArgs b = a;
    b.fac = scratch;                                  // the factorisation kernel in "matrix blockIdx.x of fac" mode
    hipLaunchKernelGGL(niw_chol_kernel, dim3(n), dim3(256), niw_master_lds_bytes(a.DP), s, b, (const int32_t *)nullptr, small);
    return hipGetLastError();
}

// rows_store[slots[i]] -> dst[i] (pinned host memory): the statistics rows of the listed slots in one pass
__global__ void niw_rows_gather_kernel(const double *__restrict__ rows_store, const int32_t *__restrict__ slots, int n, int64_t two_stride,
                                       double *__restrict__ dst) {
    const int64_t total = (int64_t)n * two_stride;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / two_stride);
        dst[e] = rows_store[(int64_t)slots[i] * two_stride + (e - (int64_t)i * two_stride)];
    }
}
hipError_t launch_niw_rows_gather(const double *rows_store, const int32_t *slots, int n, int64_t stride, double *dst, hipStream_t s) {
    const int64_t total = (int64_t)n * 2 * stride;
    const int grid = (int)std::min<int64_t>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(niw_rows_gather_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, rows_store, slots, n, 2 * stride, dst);
    return hipGetLastError();
}

size_t niw_master_lds_bytes(int DP) { return sizeof(double) * ((size_t)32 * DP + 16 * 17 + DP); }

hipError_t launch_niw_master_posterior(const NiwMasterArgs &a, const int32_t *jobs, int njobs, const double *rows, double *small, hipStream_t s) {
    if (njobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(niw_form_kernel, dim3(3 * njobs, a.DP >= 64 ? 16 : 1), dim3(256), 0, s, a, jobs, rows, small);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void *)niw_chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_master_lds_bytes(DPMM_MASTER_MAXD));
        hipFuncSetAttribute((const void *)niw_draw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_master_lds_bytes(DPMM_MASTER_MAXD));
        attr = true;
    }
    hipLaunchKernelGGL(niw_chol_kernel, dim3(3 * njobs), dim3(256), niw_master_lds_bytes(a.DP), s, a, jobs, small);
    return hipGetLastError();
}

// what: bit 0 = the draws (Y, mu_draw, logdet_sigma from the posteriors), bit 1 = the hand-over to the sweep kernels (needs lr / wts).
// The two halves may be launched apart (the draws do not depend on the weights): see dpmm_step_master_device.
hipError_t launch_niw_master_draw(const NiwMasterArgs &a, const int32_t *slot_of_cluster, int K, uint32_t epoch, double *Y, float *logdet_sigma,
                                  const float *lr, const float *wts, float *Rp, float *mup, float *cst, float *tail, int NB,
                                  unsigned long long *work, int what, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void *)niw_chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_master_lds_bytes(DPMM_MASTER_MAXD));
        hipFuncSetAttribute((const void *)niw_draw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_master_lds_bytes(DPMM_MASTER_MAXD));
        attr = true;
    }
    if (what & 1) hipLaunchKernelGGL(niw_draw_kernel, dim3(3 * K), dim3(256), niw_master_lds_bytes(a.DP), s, a, slot_of_cluster, epoch, Y, logdet_sigma);
    if (what & 2) hipLaunchKernelGGL(niw_master_pack_kernel, dim3(512), dim3(256), 0, s, Y, a.mu_draw, logdet_sigma, lr, wts, Rp, mup, cst, tail, a.D, a.DP, NB,
                                     3 * K, work);
    return hipGetLastError();
}

}  // namespace dpmm
