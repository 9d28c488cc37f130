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
//           diagonal block (wave 0, right-looking, the next pivot row kept up to date in registers, no divergent code) -> panel (four
//           lanes per column, DPP broadcasts) -> rank-16 trailing update of 16 x 16 blocks on the FP64 matrix cores.  D <= 128: one
//           kernel forms, factorises and writes out with the matrix resident in LDS (niw_post_lds_kernel); larger D: niw_form_kernel
//           + niw_chol_kernel with the matrix in global memory.  Also returned: log Gamma_D(nu' / 2), the lgamma terms of the master's
//           log-marginal.
//   noise   the standard normals of the Bartlett factors depend on (seed, epoch, position, element) only: niw_noise_kernel generates
//           them for the NEXT epoch on a second stream, several workgroups per matrix
//   draw    chi on the diagonal, L Y = A blocked by 16 rows (products on the FP64 matrix cores, B operands in two register groups in
//           flight, one thread per column for the 16 x 16 triangular part), R = Y' is the upper-triangular factor of a
//           Wishart(nu', P^-1) draw; mu = m' + R^-1 xi / sqrt(kappa') by a blocked back-substitution
//   pack    R, mu, additive constants -> the fragment images of the sweep kernels (no host staging, no copy)
// One workgroup of 256 threads per distribution.  At one wave per SIMD an instruction issues every ~5.5 cycles whatever it is: these
// kernels are written to minimise instructions on the sequential chains (5.6 Mflop per factorisation at D = 256: synchronisation and
// issue, not arithmetic, is what a 256 x 256 problem costs).  Phase cycles: -DDPMM_POST_STAMPS + scripts/post_stamps.py.  Measured and
// dropped: 512 threads per workgroup (factorisation 5 % faster, draws 30 % slower), 1024 (half the register budget: spills), an explicit
// one-step prefetch of the Y rows in the solve's product loop (+7 %), a products variant in which a wave owns four column blocks and
// walks the chunks once (its first chunks have one or two active blocks and wait for every load).
#include <algorithm>
#include "dpmm_device.h"
#include "dpmm_kernels.h"
#include "niw_b3.h"        // pair_ball_block: the lean kernel's pair-ball table, a role of the hand-over kernel

namespace dpmm {

enum : uint32_t { STREAM_M_NORMAL = 32, STREAM_M_CHI = 33, STREAM_M_XI = 34 };
typedef double f64x4m __attribute__((ext_vector_type(4)));
constexpr int NS = DPMM_MASTER_NSCALARS;      // doubles per distribution in `small`: N, kappa', nu', log det(nu' psi'), log Gamma_D(nu' / 2), 3 spare

// log Gamma_D(x) = D (D - 1) / 4 log pi + sum_{d=1..D} lgamma(x + (1 - d) / 2)   (utils.jl:66-72), by the 64 lanes of a wave: what the
// master's log-marginals need besides the log-determinant -- D lgamma evaluations per distribution, a pool job on the host
__device__ __forceinline__ double log_mv_gamma_wave(double x, int D, int lane) {
    double t = 0.0;
    for (int d = 1 + lane; d <= D; d += 64) t += lgamma(x + 0.5 * (double)(1 - d));
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    return t + (double)D * (double)(D - 1) * 0.25 * 1.1447298858494001741434273513530587;
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {          // (0, 1), 53 bits
    return ((double)((((uint64_t)a << 32) | b) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double normal_from(const Philox4 &r) {        // one standard normal from one generator block
    const double u1 = u53(r.v[0], r.v[1]), u2 = u53(r.v[2], r.v[3]);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
// Marsaglia-Tsang Gamma(a, 1); trial t of element `idx` uses generator blocks (idx, 2t) and (idx, 2t + 1).  a < 1: Gamma(a) = Gamma(a + 1)
// U^(1/a), U from block (idx, 63).  (Written without recursion and inlined: as a real call the kernel carried a dynamic stack and kept
// its uniform values in lanes of a spill register across the call.)
__device__ __forceinline__ double gamma_mt(double a, uint64_t seed, uint64_t idx, uint32_t epoch) {
    double boost = 1.0;
    if (a < 1.0) {
        const Philox4 ru = philox4x32_10(seed, idx * 64 + 63, epoch, STREAM_M_CHI);
        boost = pow(u53(ru.v[0], ru.v[1]), 1.0 / a);
        a += 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0;; ++t) {
        const double x = normal_from(philox4x32_10(seed, idx * 64 + 2 * t, epoch, STREAM_M_CHI));
        const Philox4 ru = philox4x32_10(seed, idx * 64 + 2 * t + 1, epoch, STREAM_M_CHI);
        const double u = u53(ru.v[0], ru.v[1]);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (u < 1.0 - 0.0331 * x * x * x * x) return (d * v) * boost;
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return (d * v) * boost;
        if (t > 60) return (d * v) * boost;      // unreachable in practice (acceptance > 95 %): bounded for safety
    }
}

// Diagonal of the Bartlett factor of a Wishart(nu, .) draw (Distributions.jl's sampler behind niw.jl:35): A_rr = chi with nu - r degrees
// of freedom, r = 0 .. D-1 (chi^2_dof = 2 Gamma(dof / 2)); `id` = position of the distribution in cluster order
__device__ __forceinline__ double bartlett_diag(uint64_t seed, uint64_t id, uint32_t epoch, int r, double nu) {
    return sqrt(2.0 * gamma_mt(0.5 * (nu - (double)r), seed, (id << 16) + (uint64_t)r, epoch));
}

// ------------------------------------------------------------------------------------------------------------------ form
// job j: cluster k = jobs[2j] (0-based), slot s = jobs[2j+1]; rows 3s + w, w = 0 (cluster = left + right), 1 (left), 2 (right).
// small[(3j + w) * NS + {0,1,2,4}] = N, kappa', nu', log Gamma_D(nu' / 2)   (entry 3: log det(nu' psi'), written by the factorisation)
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
        double *o = small + (int64_t)(3 * j + w) * NS;
        o[0] = N; o[1] = (N == 0.0) ? k0 : k1; o[2] = (N == 0.0) ? v0 : v1;
        A.kap[row] = o[1]; A.nu[row] = o[2];
    }
    if (blockIdx.y == 0 && threadIdx.x < 64) {
        const double lm = log_mv_gamma_wave(0.5 * ((N == 0.0) ? v0 : v1), D, threadIdx.x);
        if (threadIdx.x == 0) small[(int64_t)(3 * j + w) * NS + 4] = lm;
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

// lane s of every quad -> the whole quad (DPP quad_perm, no LDS traffic)
__device__ __forceinline__ double quad_bcast(double x, int s) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    switch (s) {
        case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true); break;
        case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xf, 0xf, true); break;
        case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0xaa, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xaa, 0xf, 0xf, true); break;
        default: lo = __builtin_amdgcn_mov_dpp(lo, 0xff, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xff, 0xf, 0xf, true); break;
    }
    return __hiloint2double(hi, lo);
}

// 1 / sqrt(x) for a pivot: the hardware estimate + two Newton steps (error ~1 ulp for normal x > 0; NaN / <= 0 propagate to the s_bad
// test of the caller).  The library rsqrt costs ~3x as much inside a 16-step dependent chain.
__device__ __forceinline__ double rsqrt_pivot(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = __builtin_fma(y, __builtin_fma(-hx * y, y, 0.5), y);
    y = __builtin_fma(y, __builtin_fma(-hx * y, y, 0.5), y);
    return y;
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
    // row stride DP + 16: the matrix-core operand fetch reads 16 consecutive doubles of rows r, r + 1, r + 2, r + 3 -- with a stride that
    // is 16 doubles modulo 32 the four rows fall into alternating halves of the 64 banks (two passes, the minimum for 64 x 8 bytes); the
    // plain stride DP put all four on the same 32 banks (four passes: the trailing update spent more time fetching than multiplying)
    const int WS = DP + 16;
    double *Wp = lds;                    // [16][WS]  factor rows of the current block, columns 0 .. j0 + 15
    double *Dg = lds + 16 * WS;          // [16][17]  diagonal block (symmetric working copy, then the factor; column 16: reciprocal pivots)
    __shared__ double s_ld;
    __shared__ int s_bad;
    __shared__ uint16_t tri[136];        // block index t of the trailing update -> (kb, qb), qb <= kb: t = kb (kb + 1) / 2 + qb
    if (tid == 0) { s_ld = 0.0; s_bad = 0; }
    if (tid < 16) for (int qb = 0; qb <= tid; ++qb) tri[tid * (tid + 1) / 2 + qb] = (uint16_t)(tid | (qb << 8));
#ifdef DPMM_POST_STAMPS
    unsigned long long C0 = __builtin_amdgcn_s_memtime(), Cd = 0, Cp = 0, Ct = 0, Ca = 0, Cb = 0;
#define CSTAMP(x) x
#else
#define CSTAMP(x)
#endif
    __syncthreads();
    for (int jb = NB - 1; jb >= 0; --jb) {
        const int j0 = 16 * jb;
        CSTAMP(Ca = __builtin_amdgcn_s_memtime();)
        // (1) diagonal block -> LDS, mirrored into a full symmetric block, factorised from its last row up by wave 0 with the
        //     look-ahead scheme of niw_post_lds_kernel (every step the same arithmetic in every lane, no divergent code)
        { const int a = tid >> 4, b = tid & 15; Dg[a * 17 + b] = (b <= a) ? P[(int64_t)(j0 + a) * DP + j0 + b] : P[(int64_t)(j0 + b) * DP + j0 + a]; }
        __syncthreads();
        if (tid < 64) {
            const int a = tid >> 2, cb = 4 * (tid & 3);
            double *mine = Dg + a * 17 + cb;
            double mv[4] = {mine[0], mine[1], mine[2], mine[3]};
            const double *r15 = Dg + 15 * 17;
            double pv = r15[15], ra = r15[a], q[4] = {r15[cb], r15[cb + 1], r15[cb + 2], r15[cb + 3]};
            bool badp = false;
            double lg = 0.0;
#pragma unroll
            for (int jj = 15; jj >= 0; --jj) {
                const int jn = jj > 0 ? jj - 1 : 0;
                const double *rn = Dg + jn * 17;                    // row jj - 1 as stored (up to date with the steps > jj)
                const double n_pv = rn[jn], n_ra = rn[a];
                const double n_q[4] = {rn[cb], rn[cb + 1], rn[cb + 2], rn[cb + 3]};
                const double inv = rsqrt_pivot(pv);
                badp |= !(pv > 0.0);
                if ((tid & 15) == jj) lg = pv;                        // lane jj keeps pivot jj for the log-determinant
                const double f = (ra * inv) * inv;
                const double alpha = a == jj ? inv : 1.0, beta = a < jj ? f : 0.0;
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) mv[i2] = __builtin_fma(-beta, q[i2], alpha * mv[i2]);
                mine[0] = mv[0]; mine[1] = mv[1]; mine[2] = mv[2]; mine[3] = mv[3];
                if (tid == 0) Dg[jj * 17 + 16] = inv;                 // reciprocal pivots (the unused 17th column): the panel multiplies
                const double qsrc = q[jn & 3];
                const double qjm = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(qsrc), jn >> 2),
                                                    __builtin_amdgcn_readlane(__double2loint(qsrc), jn >> 2));
                const double fn = (qjm * inv) * inv;
                pv = __builtin_fma(-fn, qjm, n_pv);
                ra = __builtin_fma(-fn, ra, n_ra);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) q[i2] = __builtin_fma(-fn, q[i2], n_q[i2]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (tid < 16) {
                double l2 = log(lg);                                  // log det P = sum log pivot
                for (int o = 8; o > 0; o >>= 1) l2 += __shfl_xor(l2, o, 16);
                if (tid == 0) { s_ld += l2; if (badp) s_bad = 1; }
            }
        }
        __syncthreads();
        { const int a = tid >> 4, b = tid & 15; if (b <= a) P[(int64_t)(j0 + a) * DP + j0 + b] = Dg[a * 17 + b]; Wp[a * WS + j0 + b] = (b <= a) ? Dg[a * 17 + b] : 0.0; }
        CSTAMP(Cb = __builtin_amdgcn_s_memtime(); Cd += Cb - Ca;)
        // (2) panel: column q < j0, four lanes per column (lane s of the quad keeps rows j0 + 4 i + s); coefficients of the diagonal
        //     block and the reciprocal pivots in registers before the first step
        {
            const int sq = tid & 3;
            double cf[16][4], dv[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                dv[jj] = Dg[jj * 17 + 16];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2)
                    if (4 * i2 < jj) cf[jj][i2] = Dg[jj * 17 + 4 * i2 + sq];
            }
            double wnx[4];                                               // the next trip's column, requested before this trip's 16 steps
            { const int qc0 = (tid >> 2) < j0 ? (tid >> 2) : j0 - 1;
#pragma unroll
              for (int i2 = 0; i2 < 4; ++i2) wnx[i2] = P[(int64_t)(j0 + 4 * i2 + sq) * DP + qc0]; }
            for (int q = tid >> 2; q < ((j0 + 63) & ~63); q += 64) {
                double wv[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) wv[i2] = wnx[i2];
                { const int qn = q + 64 < j0 ? q + 64 : j0 - 1;
#pragma unroll
                  for (int i2 = 0; i2 < 4; ++i2) wnx[i2] = P[(int64_t)(j0 + 4 * i2 + sq) * DP + qn]; }
#pragma unroll
                for (int jj = 15; jj >= 0; --jj) {
                    const double wj = quad_bcast(wv[jj >> 2] * dv[jj], jj & 3);
                    if (sq == (jj & 3)) wv[jj >> 2] = wj;
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) {
                        if (4 * i2 + 3 < jj) wv[i2] = __builtin_fma(-cf[jj][i2], wj, wv[i2]);
                        else if (4 * i2 < jj) wv[i2] = (4 * i2 + sq < jj) ? __builtin_fma(-cf[jj][i2], wj, wv[i2]) : wv[i2];
                    }
                }
                if (q < j0) {
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) { P[(int64_t)(j0 + 4 * i2 + sq) * DP + q] = wv[i2]; Wp[(4 * i2 + sq) * WS + q] = wv[i2]; }
                }
            }
        }
        __syncthreads();
        CSTAMP(Ca = __builtin_amdgcn_s_memtime(); Cp += Ca - Cb;)
        // (3) trailing update of the 16 x 16 blocks (kb, qb), qb <= kb < jb: P -= W' W on the FP64 matrix cores, one block per wave and
        //     trip (C straight from / to global memory, A / B operands from the LDS panel); diagonal blocks are updated whole -- their
        //     upper half is scratch, step (1) mirrors the lower half in
        {
            // blocks t = wave, wave + 4, ... in groups of TG: the C loads of a group are all in flight before its matrix instructions
            // start, and the next group's are requested before this group's results are stored (a block is 4 x 64 matrix-pipe cycles
            // against > 1000 cycles from L2: one block at a time left the pipe idle 80 % of the phase)
            constexpr int TG = 6;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = tid & 15, lg4 = (tid >> 4) & 3;
            const int nblk = jb * (jb + 1) / 2;
            // (the wave index as a scalar: everything derived from it -- block indices, LDS and global base addresses -- is then scalar
            // arithmetic; computed per lane, with a square root and two correction loops per call, it cost more than the matrix work)
            auto block_of = [&](int t, int &kb, int &qb) {
                const int e = tri[t];
                kb = e & 0xff; qb = e >> 8;
            };
            f64x4m ca[TG], cn[TG];
            auto loadc = [&](f64x4m (&c4)[TG], int t0) {
#pragma unroll
                for (int g2 = 0; g2 < TG; ++g2) {
                    const int t = t0 + 4 * g2 < nblk ? t0 + 4 * g2 : (nblk > 0 ? nblk - 1 : 0);        // clamped: loaded, not used
                    int kb, qb; block_of(t, kb, qb);
                    const double *cp = P + (int64_t)(16 * kb + lg4) * DP + 16 * qb + li;
                    c4[g2] = (f64x4m){cp[0], cp[(int64_t)4 * DP], cp[(int64_t)8 * DP], cp[(int64_t)12 * DP]};
                }
            };
            // gfx9 counts loads and stores in ONE in-order queue (vmcnt): a load issued behind a store cannot be waited for without
            // waiting for the store's acknowledgement (~1.5 k cycles).  Hence per group: all C loads of the NEXT group first, then the
            // matrix instructions of this group (no memory operation, no branch), then its stores -- waiting for the next group's
            // loads then allows this group's stores to be outstanding.  Loads are unconditional (clamped block index).
            auto compute = [&](f64x4m (&c4)[TG], int t0) {
#pragma unroll
                for (int g2 = 0; g2 < TG; ++g2) {
                    const int t = t0 + 4 * g2 < nblk ? t0 + 4 * g2 : nblk - 1;
                    int kb, qb; block_of(t, kb, qb);
                    const double *wa = Wp + lg4 * WS + 16 * kb + li, *wb = Wp + lg4 * WS + 16 * qb + li;
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) c4[g2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-wa[4 * t4 * WS], wb[4 * t4 * WS], c4[g2], 0, 0, 0);
                }
            };
            auto storec = [&](const f64x4m (&c4)[TG], int t0) {
#pragma unroll
                for (int g2 = 0; g2 < TG; ++g2) {
                    const int t = t0 + 4 * g2;
                    if (t < nblk) {
                        int kb, qb; block_of(t, kb, qb);
                        double *cp = P + (int64_t)(16 * kb + lg4) * DP + 16 * qb + li;
                        cp[0] = c4[g2][0]; cp[(int64_t)4 * DP] = c4[g2][1]; cp[(int64_t)8 * DP] = c4[g2][2]; cp[(int64_t)12 * DP] = c4[g2][3];
                    }
                }
            };
            if (wave < nblk) {
                loadc(ca, wave);
                for (int t0 = wave; t0 < nblk; t0 += 8 * TG) {
                    loadc(cn, t0 + 4 * TG);
                    compute(ca, t0);
                    storec(ca, t0);
                    if (t0 + 4 * TG < nblk) {
                        loadc(ca, t0 + 8 * TG);
                        compute(cn, t0 + 4 * TG);
                        storec(cn, t0 + 4 * TG);
                    }
                }
            }
        }
        __syncthreads();
        CSTAMP(Ct += __builtin_amdgcn_s_memtime() - Ca;)
    }
#ifdef DPMM_POST_STAMPS
    if (tid == 0 && blockIdx.x == 5 && jobs) { double *o = small + 5 * NS; o[0] = (double)Cd; o[1] = (double)Cp; o[2] = (double)Ct; o[3] = (double)(__builtin_amdgcn_s_memtime() - C0); return; }
#endif
    if (tid == 0) small[(int64_t)blockIdx.x * NS + 3] = s_bad ? NAN : s_ld;
}

// ------------------------------------------------------------------------------------------------------------------ draw
// The standard normals of the Bartlett factors (everything below the diagonal; zeros on and above it, identity in the padding) of
// distributions 0 .. gridDim.x - 1 for one epoch: they depend on (seed, epoch, position, element) only -- not on the posteriors, not on the
// cluster -> slot map -- so they are generated ahead, gridDim.y workgroups per matrix (the draw kernel has one workgroup per matrix and
// spent 22 % of its time here at D = 256).  One generator block gives the two normals of an element pair (2p, 2p + 1) of a row
// (Box-Muller, both branches).
__device__ __forceinline__ void bartlett_pair(const NiwMasterArgs &A, uint64_t id, uint32_t epoch, int r, int c, double &n0, double &n1) {
    const int D = A.D, DP = A.DP;
    n0 = 0.0; n1 = 0.0;
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
}
__global__ __launch_bounds__(256) void niw_noise_kernel(NiwMasterArgs A, uint32_t epoch, double *__restrict__ Yall) {
    const int DP = A.DP, HP = DP / 2;
    double *Y = Yall + (int64_t)blockIdx.x * DP * DP;
    for (int p2 = blockIdx.y * 256 + threadIdx.x; p2 < DP * HP; p2 += gridDim.y * 256) {
        const int r = p2 / HP, c = 2 * (p2 - r * HP);
        double n0, n1;
        bartlett_pair(A, (uint64_t)blockIdx.x, epoch, r, c, n0, n1);
        Y[(int64_t)r * DP + c] = n0; Y[(int64_t)r * DP + c + 1] = n1;
    }
}
hipError_t launch_niw_master_noise(const NiwMasterArgs &a, int nmat, uint32_t epoch, double *Y, hipStream_t s) {
    if (nmat <= 0) return hipSuccess;
    DPMM_LAUNCH(niw_noise_kernel, dim3(nmat, a.DP >= 128 ? 8 : (a.DP >= 48 ? 4 : 1)), dim3(256), 0, s, a, epoch, Y);
    return hipGetLastError();
}

// One workgroup per distribution (blockIdx.x = 3 k + w in CLUSTER order; slot from slot_of_cluster).  Y (scratch, [3K][DP][DP]).
__global__ __launch_bounds__(256) void niw_draw_kernel(NiwMasterArgs A, const int32_t *__restrict__ slot_of_cluster, uint32_t epoch,
                                                       double *__restrict__ Yall, float *__restrict__ logdet_sigma, int have_noise) {
    const int k = blockIdx.x / 3, w = blockIdx.x % 3;
    const int row = 3 * slot_of_cluster[k] + w;
    const int D = A.D, DP = A.DP, NB = DP / 16, tid = threadIdx.x;
    const double *L = A.fac + (int64_t)row * DP * DP;
    double *Y = Yall + (int64_t)blockIdx.x * DP * DP;
    const double nu = A.nu[row], kap = A.kap[row];
    const uint64_t id = (uint64_t)blockIdx.x;          // the streams are keyed by the position in cluster order, like the host's
#ifdef DPMM_POST_STAMPS
    unsigned long long D0 = __builtin_amdgcn_s_memtime(), Dn = 0, Ds = 0, Dp = 0, Dt = 0, Da = 0, Db = 0;
#define DSTAMP(x) x
#else
#define DSTAMP(x)
#endif
    extern __shared__ double lds[];
    double *T = lds;                     // [16][DP]
    double *Ld = lds + 16 * DP;          // [16][17]
    double *xi = Ld + 16 * 17;           // [DP]
    double *Lp = xi + DP;                // [DP][17]  rows i0 .. i0 + 15 of L, columns < i0, TRANSPOSED (column kk of L at Lp[kk * 17 + row])
    // Bartlett factor (lower): chi on the diagonal, standard normals below, identity in the padding.  The normals are usually there
    // already (niw_noise_kernel, launched ahead); have_noise == 0: generated here.
    if (!have_noise) {
        const int HP = DP / 2;
        for (int p2 = tid; p2 < DP * HP; p2 += 256) {
            const int r = p2 / HP, c = 2 * (p2 - r * HP);
            double n0, n1;
            bartlett_pair(A, id, epoch, r, c, n0, n1);
            Y[(int64_t)r * DP + c] = n0; Y[(int64_t)r * DP + c + 1] = n1;
        }
    }
    __syncthreads();
    for (int r = tid; r < D; r += 256) Y[(int64_t)r * DP + r] = bartlett_diag(A.seed, id, epoch, r, nu);
    for (int d = tid; d < DP; d += 256)
        xi[d] = d < D ? normal_from(philox4x32_10(A.seed, (id << 32) + (uint64_t)d, epoch, STREAM_M_XI)) : 0.0;
    __syncthreads();
    DSTAMP(Dn = __builtin_amdgcn_s_memtime() - D0;)
    // L Y = A, block row by block row
    { const int a2 = tid >> 4, b2 = tid & 15; const double dv = L[(int64_t)a2 * DP + b2]; Ld[a2 * 17 + b2] = (b2 <= a2) ? dv : 0.0; if (a2 == b2) Ld[a2 * 17 + 16] = 1.0 / dv; }
    __syncthreads();
    for (int ib = 0; ib < NB; ++ib) {
        const int i0 = 16 * ib, ncol = i0 + 16;             // columns 0 .. i0 + 15 of this block row can be non-zero
        DSTAMP(Da = __builtin_amdgcn_s_memtime();)
        // (the block row's part of L -- columns < i0 transposed in Lp, diagonal block + reciprocal pivots in Ld -- was staged during the
        // previous block row's triangular phase; row 0 before the loop)
        DSTAMP(Db = __builtin_amdgcn_s_memtime(); Ds += Db - Da;)
        // T = A(ib, :) - L(ib, < i0) Y(< i0, :) on the FP64 matrix cores (v_mfma_f64_16x16x4_f64: A[i][k = g] = -L[i0 + i][kk + g] from
        // the transposed LDS panel (conflict-free), B[k = g][col = i] = Y[kk + g][16 cb + i] from global memory, element r of C at row
        // g + 4 r, col i).  Column block cb needs the rows kk >= 16 cb only (Y is lower triangular): ib - cb chunks of 16 rows; the
        // blocks are dealt to the four waves in snake order (similar numbers of chunks), B operands in two register groups of PG
        // chunks: the loads of one group are in flight while the matrix instructions of the other run (PG x 4 x 64 cycles of cover
        // against > 1000 cycles from L2).  D = 256: 488 k cycles (4 x 4 register tiles on the vector ALU) -> 146 k.  (A variant in
        // which a wave owns four column blocks and walks the chunks once -- one A fragment per chunk, one pipeline fill per block row
        // -- was slower, 195 k: its first chunks have one or two active blocks and wait for every load.)
        {
            const int wave = tid >> 6, li = tid & 15, lg = (tid >> 4) & 3;
            for (int idx = 0; idx <= ib; ++idx) {
                const int turn = idx & 7;
                if ((turn < 4 ? turn : 7 - turn) != wave) continue;
                const int cb = idx;
                const double *yc = Y + (int64_t)lg * DP + 16 * cb + li;                 // + kk * DP
                f64x4m acc = {yc[(int64_t)(i0 + 0) * DP], yc[(int64_t)(i0 + 4) * DP], yc[(int64_t)(i0 + 8) * DP], yc[(int64_t)(i0 + 12) * DP]};
                const double *lp = Lp + lg * 17 + li;                                      // + kk * 17
                constexpr int PG = 5;
                const int nch = ib - cb;
                double bA[PG][4], bB[PG][4];
                auto loadg = [&](double (&buf)[PG][4], int ch0) {
#pragma unroll
                    for (int g2 = 0; g2 < PG; ++g2) {
                        const int ch = ch0 + g2 < nch ? ch0 + g2 : nch - 1;             // clamped: loaded, not used
#pragma unroll
                        for (int u = 0; u < 4; ++u) buf[g2][u] = yc[(int64_t)(16 * (cb + ch) + 4 * u) * DP];
                    }
                };
                auto mm = [&](const double (&buf)[PG][4], int ch0) {
#pragma unroll
                    for (int g2 = 0; g2 < PG; ++g2)
                        if (ch0 + g2 < nch) {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lp[(16 * (cb + ch0 + g2) + 4 * u) * 17], buf[g2][u], acc, 0, 0, 0);
                        }
                };
                if (nch > 0) {
                    loadg(bA, 0);
                    for (int ch0 = 0; ch0 < nch; ch0 += 2 * PG) {
                        if (ch0 + PG < nch) loadg(bB, ch0 + PG);
                        mm(bA, ch0);
                        if (ch0 + 2 * PG < nch) loadg(bA, ch0 + 2 * PG);
                        if (ch0 + PG < nch) mm(bB, ch0 + PG);
                    }
                }
                double *tp = T + lg * DP + 16 * cb + li;
                tp[0] = acc[0]; tp[4 * DP] = acc[1]; tp[8 * DP] = acc[2]; tp[12 * DP] = acc[3];
            }
        }
        __syncthreads();
        DSTAMP(Da = __builtin_amdgcn_s_memtime(); Dp += Da - Db;)
        // the NEXT block row's part of L is requested now and written to LDS after the triangular phase (which hides the latency):
        // columns < i0 + 16 of rows i0 + 16 .. i0 + 31 (thread kk keeps a column), its diagonal block (thread -> element)
        const int n0 = i0 + 16;
        const bool more = ib + 1 < NB;
        double lnx[16], dnx = 0.0;
        {
            const int kc = (more && tid < n0) ? tid : 0, rb = more ? n0 : 0;
#pragma unroll
            for (int a2 = 0; a2 < 16; ++a2) lnx[a2] = L[(int64_t)(rb + a2) * DP + kc];
            dnx = L[(int64_t)(rb + (tid >> 4)) * DP + rb + (tid & 15)];
        }
        // 16 x 16 triangular part: one thread per column, right-looking (every finished y_r is subtracted from the rows below at once)
        for (int q = tid; q < ncol; q += 256) {
            double t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = T[r * DP + q];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double yv = t[r] * Ld[r * 17 + 16];
                t[r] = yv;
#pragma unroll
                for (int r2 = r + 1; r2 < 16; ++r2) t[r2] = __builtin_fma(-Ld[r2 * 17 + r], yv, t[r2]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[(int64_t)(i0 + r) * DP + q] = (q <= i0 + r) ? t[r] : 0.0;
        }
        __syncthreads();
        if (more) {
            if (tid < n0) {
#pragma unroll
                for (int a2 = 0; a2 < 16; ++a2) Lp[tid * 17 + a2] = lnx[a2];
            }
            const int a2 = tid >> 4, b2 = tid & 15;
            Ld[a2 * 17 + b2] = (b2 <= a2) ? dnx : 0.0;
            if (a2 == b2) Ld[a2 * 17 + 16] = 1.0 / dnx;
        }
        __syncthreads();
        DSTAMP(Dt += __builtin_amdgcn_s_memtime() - Da;)
    }
    DSTAMP(Da = __builtin_amdgcn_s_memtime();)
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
        // everything this block needs from global memory is requested first: the diagonal block and, for the update pass, the 16 values
        // of column r of the block's rows (they do not depend on the solve)
        { const int a2 = tid >> 4, b2 = tid & 15; Ld[a2 * 17 + b2] = (b2 <= a2) ? Y[(int64_t)(c0 + a2) * DP + c0 + b2] : 0.0; }
        double yr[16];
        const int rr = tid < c0 ? tid : 0;
#pragma unroll
        for (int a2 = 0; a2 < 16; ++a2) yr[a2] = Y[(int64_t)(c0 + a2) * DP + rr];
        __syncthreads();
        if (tid < 64) {
            // 16 x 16 back substitution by the first 16 lanes (the whole wave runs it: uniform code): lane t keeps x_t and column t of the
            // block below the diagonal; step a2 broadcasts the finished v_a2 = x_a2 / L_a2a2 with v_readlane (no LDS round trip)
            const int t16 = tid & 15;
            double xb = xi[c0 + t16];
            const double inv = 1.0 / Ld[t16 * 17 + t16];
            double col[16];
#pragma unroll
            for (int a2 = 0; a2 < 16; ++a2) col[a2] = a2 > t16 ? Ld[a2 * 17 + t16] : 0.0;
            double mine_v = 0.0;
#pragma unroll
            for (int a2 = 15; a2 >= 0; --a2) {
                const double wq = xb * inv;
                const double va = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wq), a2), __builtin_amdgcn_readlane(__double2loint(wq), a2));
                mine_v = t16 == a2 ? va : mine_v;
                xb = __builtin_fma(-va, col[a2], xb);
            }
            if (tid < 16) vv[c0 + tid] = mine_v;
        }
        __syncthreads();
        for (int r = tid; r < c0; r += 256) {
            double acc = xi[r];
#pragma unroll
            for (int a2 = 0; a2 < 16; ++a2) acc = __builtin_fma(-vv[c0 + a2], yr[a2], acc);
            xi[r] = acc;
        }
        __syncthreads();
    }
    const double isk = 1.0 / sqrt(kap);
    const double *m = A.mean + (int64_t)row * DP;
    float *mu_out = A.mu_draw + (int64_t)blockIdx.x * DP;
    for (int d = tid; d < DP; d += 256) mu_out[d] = d < D ? (float)(m[d] + vv[d] * isk) : 0.f;
#ifdef DPMM_POST_STAMPS      // diagnostic build: phase cycles (scripts/post_stamps.py)
    __syncthreads();
    if (tid == 0 && blockIdx.x == 3) { float *o = logdet_sigma + 3; o[0] = (float)Dn; o[1] = (float)Ds; o[2] = (float)Dp; o[3] = (float)Dt; o[4] = (float)(__builtin_amdgcn_s_memtime() - Da); o[5] = (float)(__builtin_amdgcn_s_memtime() - D0); }
#endif
}

// Diagnostic: the random INPUTS of the draw of `epoch` for the distributions in cluster order -- the Bartlett factor A ([3K][D][D], lower:
// chi_{nu' - r} on the diagonal, standard normals below) and the mean normals xi ([3K][D]) -- produced by the same device functions
// with the same keys as niw_noise_kernel / niw_draw_kernel.  The draw is a deterministic function of (L, A, xi): tests recompute it in
// numpy from these (dpmm_debug_niw_draw_inputs).
__global__ __launch_bounds__(256) void niw_draw_inputs_kernel(NiwMasterArgs A, const int32_t *__restrict__ slot_of_cluster, uint32_t epoch,
                                                              double *__restrict__ Aout, double *__restrict__ xiout) {
    const int k = blockIdx.x / 3, w = blockIdx.x % 3;
    const int row = 3 * slot_of_cluster[k] + w;
    const int D = A.D, DP = A.DP, HP = DP / 2, tid = threadIdx.x;
    const double nu = A.nu[row];
    const uint64_t id = (uint64_t)blockIdx.x;
    double *Ao = Aout + (int64_t)blockIdx.x * D * D;
    for (int p2 = tid; p2 < DP * HP; p2 += 256) {
        const int r = p2 / HP, c = 2 * (p2 - r * HP);
        double n0, n1;
        bartlett_pair(A, id, epoch, r, c, n0, n1);
        if (r < D && c < D && c != r) Ao[(int64_t)r * D + c] = n0;
        if (r < D && c + 1 < D && c + 1 != r) Ao[(int64_t)r * D + c + 1] = n1;
    }
    for (int r = tid; r < D; r += 256) Ao[(int64_t)r * D + r] = bartlett_diag(A.seed, id, epoch, r, nu);
    for (int d = tid; d < D; d += 256)
        xiout[(int64_t)blockIdx.x * D + d] = normal_from(philox4x32_10(A.seed, (id << 32) + (uint64_t)d, epoch, STREAM_M_XI));
}
hipError_t launch_niw_draw_inputs(const NiwMasterArgs &a, const int32_t *slot_of_cluster, int K, uint32_t epoch, double *Aout, double *xiout, hipStream_t s) {
    DPMM_LAUNCH(niw_draw_inputs_kernel, dim3(3 * K), dim3(256), 0, s, a, slot_of_cluster, epoch, Aout, xiout);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------ pack
// Y ([3K][DP][DP] Float64, R = Y') + mu -> the sweep kernels' fragment images (same layout as niw_pack_kernel), constants, tail records
__global__ void niw_master_pack_kernel(const double *__restrict__ Yall, const float *__restrict__ mu_draw, const float *__restrict__ logdet_sigma,
                                       const float *__restrict__ lr, const float *__restrict__ wts, float *__restrict__ Rp,
                                       float *__restrict__ mup, float *__restrict__ cst, float *__restrict__ tail, int D, int DPm, int NB,
                                       int nmat, unsigned long long *__restrict__ work) {
    const int NP = NB * (NB + 1) / 2;
    const int DP = 16 * NB;                   // the sweep kernels' padded dimension (>= DPm, the master's)
    if (work && blockIdx.x == 0 && threadIdx.x < 8) { work[threadIdx.x] = 0ull; work[8 + 16 * threadIdx.x] = 0ull; }     // counters / queue heads of the next sweep
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
        // per-cluster records of the ball test, as in niw_pack_kernel: { m | T row 0 | T11 T12 T13 T22 | T23 T33 |T|_F cst }
        float *ball = tail + 32 * NPR;
        for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)K * 16; e += (int64_t)gridDim.x * blockDim.x) {
            const int q = (int)(e & 15), k = (int)(e >> 4);
            const int64_t j = 3 * (int64_t)k;
            const int tr[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, tc[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
            float v;
            if (q < 4) v = mu_draw[j * DPm + f0 + q];
            else if (q < 14) v = (float)Yall[j * DPm * DPm + (int64_t)(f0 + tc[q - 4]) * DPm + (f0 + tr[q - 4])];
            else if (q == 14) {
                float t10[10];
                for (int i = 0; i < 10; ++i) t10[i] = (float)Yall[j * DPm * DPm + (int64_t)(f0 + tc[i]) * DPm + (f0 + tr[i])];
                v = tail_opnorm_bound(t10);
            } else v = -0.5f * logdet_sigma[j] + logf(wts[k]);
            ball[e] = v;
        }
        // bf16 image of the cluster-level factors for the reference bracket of the D <= 64 sweep (refb_map, dpmm_device.h)
        if (NB == 4) {
            uint32_t *refb = reinterpret_cast<uint32_t *>(ball + 16 * (size_t)K);
            for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)K * REFB_WORDS; e += (int64_t)gridDim.x * blockDim.x) {
                const int64_t j = 3 * (e / REFB_WORDS);
                int row, c0, c1;
                refb_map((int)(e % REFB_WORDS), row, c0, c1);
                const float v0 = (row < D && c0 < D && c0 >= row) ? (float)Yall[j * DPm * DPm + (int64_t)c0 * DPm + row] : 0.f;
                const float v1 = (row < D && c1 < D && c1 >= row) ? (float)Yall[j * DPm * DPm + (int64_t)c1 * DPm + row] : 0.f;
                refb[e] = bf16_rne_bits(v0) | (bf16_rne_bits(v1) << 16);
            }
        }
    }
}

// The same hand-over with the work PARTITIONED OVER WORKGROUPS BY ROLE (round 6, DPMM_OPT_CHAIN_FUSION bit 2) + the three-plane images and
// offsets of the sub-cluster factors (niw_b3_pack_kernel's job, niw_lean.hip) in the same launch.  niw_master_pack_kernel walks its outputs
// one grid-stride loop after the other: a thread pays the latency of every loop in turn (9.5 us for a few hundred kilobytes), and the images
// followed in a launch of their own (5 us, the launch floor).  Here a workgroup has ONE role -- fragment images | constants, means, tail and
// ball records | bracket images | three-plane images | offsets -- and the roles run side by side: one latency chain instead of five and one
// launch instead of two.  Values: every output is computed from the same Float64 factors by the same expressions (the three-plane images and
// offsets from (float)Y as niw_b3_pack_kernel computes them from the Float32 fragment image that holds exactly those floats; the offsets'
// Float64 sums in the same order: four 16-column partial sums per row, combined by two butterfly steps).
struct PackRoles { int first[7]; };      // role r owns workgroups [first[r], first[r + 1])
__global__ __launch_bounds__(256) void niw_master_pack_roles_kernel(const double *__restrict__ Yall, const float *__restrict__ mu_draw, const float *__restrict__ logdet_sigma,
                                                                    const float *__restrict__ lr, const float *__restrict__ wts, float *__restrict__ Rp,
                                                                    float *__restrict__ mup, float *__restrict__ cst, float *__restrict__ tail, int D, int DPm, int NB,
                                                                    int nmat, unsigned long long *__restrict__ work, PackRoles R, int with_b3) {
    const int NP = NB * (NB + 1) / 2;
    const int DP = 16 * NB;
    const int K = nmat / 3;
    int role = 0;
    while (role < 5 && (int)blockIdx.x >= R.first[role + 1]) ++role;
    const int64_t t0 = ((int64_t)blockIdx.x - R.first[role]) * 256 + threadIdx.x, stride = (int64_t)(R.first[role + 1] - R.first[role]) * 256;
    auto Rel = [&](int64_t j, int row, int col) -> float {          // R[row][col] of matrix j as the sweep's Float32 (0 outside the upper triangle / beyond D)
        return (row < D && col < D && col >= row) ? (float)Yall[j * DPm * DPm + (int64_t)col * DPm + row] : 0.f;
    };
    if (role == 0) {                       // the Float32 fragment images: four consecutive columns of one row per thread, one 16-byte store
        const int64_t total4 = (int64_t)nmat * NP * 64;
        for (int64_t u = t0; u < total4; u += stride) {
            const int lane = (int)(u & 63);
            const int64_t pj = u >> 6;
            const int pair = (int)(pj % NP);
            const int64_t j = pj / NP;
            int bi = 0, rem = pair;
            while (rem >= NB - bi) { rem -= NB - bi; ++bi; }
            const int t = bi + rem;
            const int row = 16 * bi + (lane & 15), col = 16 * t + 4 * (lane >> 4);
            float4 v;
            v.x = Rel(j, row, col); v.y = Rel(j, row, col + 1); v.z = Rel(j, row, col + 2); v.w = Rel(j, row, col + 3);
            reinterpret_cast<float4 *>(Rp)[u] = v;
        }
        return;
    }
    if (role == 1) {                       // constants | padded means | tail pair records | ball records: ONE index space, one item per thread and trip
        if (work && t0 < 8) { work[t0] = 0ull; work[8 + 16 * t0] = 0ull; }     // counters / queue heads of the next sweep
        const bool tl = tail && D >= 4;
        const int f0 = D - 4, NPR = (K + 1) / 2;
        const int64_t n0 = nmat, n1 = n0 + (int64_t)nmat * DP, n2 = n1 + (tl ? (int64_t)NPR * 32 : 0), n3 = n2 + (tl ? (int64_t)K * 16 : 0);
        const int tr[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, tc[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
        for (int64_t i = t0; i < n3; i += stride) {
            if (i < n0) {
                const int k = (int)(i / 3), w = (int)(i % 3);
                const float lw = w == 0 ? logf(wts[k]) : logf(lr[2 * k + (w - 1)]);
                cst[i] = -0.5f * logdet_sigma[i] + lw;
            } else if (i < n1) {
                const int64_t e = i - n0;
                const int d = (int)(e % DP);
                const int64_t j = e / DP;
                mup[e] = d < D ? mu_draw[j * DPm + d] : 0.f;
            } else if (i < n2) {
                const int64_t e = i - n1;
                const int c = (int)(e & 1), q = (int)((e >> 1) & 15), k = 2 * (int)(e >> 5) + c;
                const int64_t j = k < K ? 3 * k : 0;
                float v = 0.f;
                if (k >= K) v = (q == 14) ? -INFINITY : 0.f;
                else if (q < 10) v = (float)Yall[j * DPm * DPm + (int64_t)(f0 + tc[q]) * DPm + (f0 + tr[q])];
                else if (q < 14) v = mu_draw[j * DPm + f0 + (q - 10)];
                else if (q == 14) v = -0.5f * logdet_sigma[j] + logf(wts[k]);
                tail[e] = v;
            } else {
                const int64_t e = i - n2;
                float *ball = tail + 32 * NPR;
                const int q = (int)(e & 15), k = (int)(e >> 4);
                const int64_t j = 3 * (int64_t)k;
                float v;
                if (q < 4) v = mu_draw[j * DPm + f0 + q];
                else if (q < 14) v = (float)Yall[j * DPm * DPm + (int64_t)(f0 + tc[q - 4]) * DPm + (f0 + tr[q - 4])];
                else if (q == 14) {
                    float t10[10];
#pragma unroll
                    for (int ii = 0; ii < 10; ++ii) t10[ii] = (float)Yall[j * DPm * DPm + (int64_t)(f0 + tc[ii]) * DPm + (f0 + tr[ii])];
                    v = tail_opnorm_bound(t10);
                } else v = -0.5f * logdet_sigma[j] + logf(wts[k]);
                ball[e] = v;
            }
        }
        return;
    }
    if (!(tail && D >= 4 && NB == 4)) return;
    const int NPR = (K + 1) / 2;
    float *ball = tail + 32 * NPR;
    uint32_t *refb = reinterpret_cast<uint32_t *>(ball + 16 * (size_t)K);
    if (role == 2) {                       // bf16 image of the cluster-level factors (reference bracket)
        for (int64_t e = t0; e < (int64_t)K * REFB_WORDS; e += stride) {
            const int64_t j = 3 * (e / REFB_WORDS);
            int row, c0, c1;
            refb_map((int)(e % REFB_WORDS), row, c0, c1);
            refb[e] = bf16_rne_bits(Rel(j, row, c0)) | (bf16_rne_bits(Rel(j, row, c1)) << 16);
        }
        return;
    }
    if (role == 5) {                       // the lean kernel's pair-ball table (niw_b3.h pair_ball_block): one workgroup per cluster, from the floats roles 0 / 1 write
        const int j = (int)blockIdx.x - R.first[5];
        pair_ball_block<true>([&](int row, int col) -> float { return Rel(3 * (int64_t)j, row, col); },
                              [&](int kk, int c) -> float { return c < D ? mu_draw[(int64_t)(3 * kk) * DPm + c] : 0.f; }, K, j,
                              const_cast<float *>(pair_ball_table(tail, K)));
        return;
    }
    if (!with_b3) return;
    uint32_t *img = refb + (size_t)K * REFB_WORDS;                                         // = b3_images(tail, K)
    float *dvec = reinterpret_cast<float *>(img + (size_t)3 * K * B3_WORDS);              // = b3_offsets(tail, K)
    if (role == 3) {                       // the three planes of the 2K sub-cluster factors: one element pair -> three dwords
        for (int64_t i = t0; i < (int64_t)2 * K * REFB_WORDS; i += stride) {
            const int e = (int)(i % REFB_WORDS);
            const int64_t m = i / REFB_WORDS, j = 3 * (m >> 1) + 1 + (m & 1);
            int row, c0, c1;
            refb_map(e, row, c0, c1);
            const float v0 = Rel(j, row, c0), v1 = Rel(j, row, c1);
            uint32_t *out = img + (size_t)j * B3_WORDS + e;
#pragma unroll
            for (int p = 0; p < 3; ++p) out[(size_t)p * REFB_WORDS] = bf16x3_plane_bits(v0, p) | (bf16x3_plane_bits(v1, p) << 16);
        }
        return;
    }
    // role 4: the offsets d = R_s (mu_k - mu_s), Float64: four threads per row (16 columns each), the order of niw_b3_pack_kernel
    const int64_t nd = (int64_t)2 * K * 256;
    for (int64_t i0 = t0 - (threadIdx.x & 63); i0 < nd; i0 += stride) {          // (wave-uniform trip count: the butterfly needs its four lanes)
        const int64_t i = i0 + (threadIdx.x & 63);
        const bool on = i < nd;
        const int part = (int)(i & 3), row = (int)((i >> 2) & 63);
        const int64_t m = on ? (i >> 8) : 0, k = m >> 1, j = 3 * k + 1 + (m & 1);
        double rv[16], dm[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {                                          // all sixteen loads of a thread in flight at once
            const int c = 16 * part + cc;
            rv[cc] = (double)Rel(j, row, c);
            const float mk = c < D ? mu_draw[3 * k * DPm + c] : 0.f, ms = c < D ? mu_draw[j * DPm + c] : 0.f;
            dm[cc] = (double)mk - (double)ms;
        }
        double acc = 0.0;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc)
            if (16 * part + cc >= row) acc += rv[cc] * dm[cc];
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (on && part == 0) dvec[(size_t)j * B3_DVEC + row] = (float)acc;
    }
}

// Pooled statistics of cluster pairs for the merge proposals (shared_actions.jl:21-27): job p pools the four stored rows of slots
// pairs[2p], pairs[2p+1]; P = nu' psi' of the pooled set -> scratch matrix p; small[NS p + {0,1,2,4}] = N, kappa', nu', log Gamma_D(nu' / 2).
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
        double *o = small + (int64_t)p * NS;
        o[0] = N; o[1] = (N == 0.0) ? k0 : k1; o[2] = (N == 0.0) ? v0 : v1;
    }
    if (blockIdx.y == 0 && threadIdx.x < 64) {
        const double lm = log_mv_gamma_wave(0.5 * ((N == 0.0) ? v0 : v1), D, threadIdx.x);
        if (threadIdx.x == 0) small[(int64_t)p * NS + 4] = lm;
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
// ------------------------------------------------------------------------------------------------- form + factor, D <= 128
// The whole scale matrix of a distribution lives in LDS (DP x (DP + 1) doubles: 33 KiB at D = 64, 129 KiB at D = 128): formed from the
// rows, factorised in place and written out once -- no global round trips between the phases and one launch instead of two (D = 64,
// 96 distributions: 11 + 41 us -> see DESIGN 3.5).  Same formula for P as niw_form_kernel / niw_form_pair_kernel; the factorisation is
// right-looking inside a 16-block (every step updates the rest of the block at once: a dependent chain of 16 short steps instead of 16
// dot products of growing length):
//   diagonal block   wave 0, lane -> (row a = lane / 4, columns 4 (lane % 4) ..): step jj scales row jj by 1 / sqrt(pivot) and
//                    subtracts its outer product from the rows above -- one wave in lockstep, LDS operations of a wave complete in
//                    order, so no barrier inside the 16 steps
//   panel            one thread per column q < j0: the 16 x 16 triangular solve in registers, right-looking as well
//   trailing update  4 x 4 register tiles, operands from LDS
// PAIRS = false: job j = blockIdx.x / 3 (cluster jobs[2j], slot jobs[2j+1]), w = blockIdx.x % 3; writes fac / mean / kap / nu /
// rows_store like the two kernels it replaces.  PAIRS = true: job p = blockIdx.x pools the four stored rows of slots jobs[2p],
// jobs[2p+1]; only small[NS p + 0..4] is written.
template <bool PAIRS>
__device__ __forceinline__ void niw_post_lds_body(const NiwMasterArgs &A, const int32_t *__restrict__ jobs, const double *__restrict__ rows,
                                                  double *__restrict__ small, const int bid) {
    const int D = A.D, DP = A.DP, NB = DP / 16, LD = DP + 1, tid = threadIdx.x;
#ifdef DPMM_POST_STAMPS
    unsigned long long T0 = __builtin_amdgcn_s_memtime(), Tf = 0, Td = 0, Tp = 0, Tt = 0, Ta = 0, Tb = 0;
#define PSTAMP(x) x
#else
#define PSTAMP(x)
#endif
    const int64_t stride = A.packed_stride;
    extern __shared__ double lds[];
    double *Pm = lds;                    // [DP][LD]
    double *sm = Pm + (size_t)DP * LD;   // [DP] posterior mean
    double *sm0 = sm + DP;               // [DP] prior mean
    double *piv = sm0 + DP;              // [DP] pivots (squares of the diagonal of L)
    double *dinv = piv + DP;             // [16] reciprocal diagonal of the current block
    __shared__ int s_bad;
    const double *r0, *r1, *r2 = nullptr, *r3 = nullptr;
    double c0 = 1.0, c1 = 1.0;
    int row = 0, w = 0, slot = 0;
    if constexpr (PAIRS) {
        // rows == nullptr: jobs name SLOTS, the rows are the stored ones; else jobs name CLUSTERS of the statistics pass `rows` (the pair
        // jobs launched together with the posteriors of that pass: the stored rows are being written by the neighbours)
        const double *src = rows ? rows : A.rows_store;
        r0 = src + (int64_t)(2 * jobs[2 * bid]) * stride; r1 = r0 + stride;
        r2 = src + (int64_t)(2 * jobs[2 * bid + 1]) * stride; r3 = r2 + stride;
    } else {
        const int j = bid / 3;
        w = bid % 3;
        slot = jobs[2 * j + 1];
        r0 = rows + (int64_t)(2 * jobs[2 * j]) * stride; r1 = r0 + stride;
        c0 = (w != 2) ? 1.0 : 0.0; c1 = (w != 1) ? 1.0 : 0.0;
        row = 3 * slot + w;
    }
    // the first batch of the form phase is requested before anything waits for a load (scalars, means): one global round trip for all
    constexpr int FB = 9;                                     // D = 64: 2080 packed elements = 8.1 per thread -> one batch
    const int T = D * (D + 1) / 2;
    double ps[FB], sa[FB], sb[FB], sc[PAIRS ? FB : 1], sd[PAIRS ? FB : 1];
    auto form_loads = [&](int t0) {
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int t = t0 + u * 256 + tid, tc = t < T ? t : T - 1;
            ps[u] = A.psi_lo[tc]; sa[u] = r0[1 + D + tc]; sb[u] = r1[1 + D + tc];
            if constexpr (PAIRS) { sc[u] = r2[1 + D + tc]; sd[u] = r3[1 + D + tc]; }
        }
    };
    form_loads(0);
    double N;
    if constexpr (PAIRS) N = r0[0] + r1[0] + r2[0] + r3[0];
    else N = c0 * r0[0] + c1 * r1[0];
    const double k0 = A.kappa0, v0 = A.nu0, k1 = k0 + N, v1 = v0 + N;
    if (tid == 0) s_bad = 0;
    if constexpr (!PAIRS) {
        if (w == 1) {      // keep the statistics of the slot (what the host keeps in its packed rows)
            double *dst = A.rows_store + (int64_t)(2 * slot) * stride;
            for (int64_t e = tid; e < 2 * stride; e += 256) dst[e] = r0[e];
        }
    }
    for (int a = tid; a < DP; a += 256) {
        const double m0 = a < D ? A.m0[a] : 0.0;
        double mv = m0;
        if (N != 0.0 && a < D) {
            double sx;
            if constexpr (PAIRS) sx = r0[1 + a] + r1[1 + a] + r2[1 + a] + r3[1 + a];
            else sx = c0 * r0[1 + a] + c1 * r1[1 + a];
            mv = (m0 * k0 + sx) / k1;
        }
        sm[a] = mv; sm0[a] = m0;
        if constexpr (!PAIRS) A.mean[(int64_t)row * DP + a] = mv;
    }
    if (tid == 0) {
        double *o = small + (int64_t)bid * NS;
        o[0] = N; o[1] = (N == 0.0) ? k0 : k1; o[2] = (N == 0.0) ? v0 : v1;
        if constexpr (!PAIRS) { A.kap[row] = o[1]; A.nu[row] = o[2]; }
    }
    __syncthreads();
    // ---- form: lower triangle of nu' psi' into LDS.  The loop runs over the PACKED index t of the rows (no idle iterations, fully
    // coalesced) in batches of FB elements per thread whose loads are all issued before the first is used: one global round trip per
    // batch instead of one per element (the element-wise loop spent 20 k of this kernel's 65 k cycles waiting at D = 64).
    {
        for (int t0 = 0; t0 < T; t0 += 256 * FB) {
            if (t0 > 0) form_loads(t0);
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const int t = t0 + u * 256 + tid;
                if (t >= T) continue;
                int a = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
                if ((a + 1) * (a + 2) / 2 <= t) ++a;
                if (a * (a + 1) / 2 > t) --a;
                const int b = t - a * (a + 1) / 2;
                double v;
                if (N == 0.0) v = ps[u] * v0;
                else {
                    double sab;
                    if constexpr (PAIRS) sab = sa[u] + sb[u] + sc[u] + sd[u];
                    else sab = c0 * sa[u] + c1 * sb[u];
                    v = ((v0 * ps[u] + (k0 * sm0[a]) * sm0[b] - (k1 * sm[a]) * sm[b] + sab) / v1) * v1;     // psi' then nu' psi' (niw.jl:29,35)
                }
                Pm[a * LD + b] = v;
            }
        }
        for (int e = tid; e < (DP - D) * DP; e += 256) {        // padding rows: identity
            const int a = D + e / DP, b = e % DP;
            if (b <= a) Pm[a * LD + b] = (a == b) ? 1.0 : 0.0;
        }
        // The strict upper halves of the diagonal blocks start as zeros.  The factorisation reads rows of a diagonal block WHOLE (its first
        // look-ahead fetches row 14 before any lane has stored a mirrored row) and multiplies what it finds right of the diagonal by a
        // zero coefficient for the rows that are finished: with whatever an earlier kernel left in LDS there, a NaN or Inf bit pattern
        // turned L[15][15] of the block into NaN (0 x NaN) -- silently, the pivots and the log-determinant were already taken.  Seen as
        // a chain that differed on a GPU whose previous tenant had left such patterns behind.
        for (int e = tid; e < NB * 256; e += 256) {
            const int blk = e >> 8, a = (e >> 4) & 15, b = e & 15;
            if (b > a) Pm[(16 * blk + a) * LD + 16 * blk + b] = 0.0;
        }
    }
    __syncthreads();
    PSTAMP(Tf = __builtin_amdgcn_s_memtime() - T0;)
    // ---- factor: P = L' L from the last block row up
    for (int jb = NB - 1; jb >= 0; --jb) {
        const int j0 = 16 * jb;
        PSTAMP(Ta = __builtin_amdgcn_s_memtime();)
        if (tid < 64) {
            const int a = tid >> 2, cb = 4 * (tid & 3);
            double *mine = Pm + (j0 + a) * LD + j0 + cb;
            // The lane's four elements of the SYMMETRIC working block (columns above the diagonal mirrored in): every step then applies
            // the same update to every lane -- rows >= jj compute garbage that is never stored -- instead of per-lane cases (exec-mask
            // branches cost more than the arithmetic at one wave per SIMD: an instruction issues every ~5.5 cycles whatever it is).
            double mv[4];
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) mv[i2] = (cb + i2 <= a) ? mine[i2] : Pm[(j0 + cb + i2) * LD + j0 + a];
            // Row jj as every lane needs it: pivot, element a, elements cb .. cb + 3.  The copy of row jj - 1 is loaded one step ahead
            // (after the stores of step jj + 1, which it must see) and brought up to date in registers with step jj's update, so the LDS
            // round trip of a step overlaps the reciprocal square root of the previous one.  Entries right of the diagonal of a stored
            // row are garbage (rows are stored whole): nothing reads them, the write-out masks them.
            const double *r15 = Pm + (j0 + 15) * LD + j0;
            double pv = r15[15], ra = r15[a], q[4] = {r15[cb], r15[cb + 1], r15[cb + 2], r15[cb + 3]};
            bool badp = false;
#pragma unroll
            for (int jj = 15; jj >= 0; --jj) {
                const int jn = jj > 0 ? jj - 1 : 0;
                const double *rn = Pm + (j0 + jn) * LD + j0;        // row jj - 1 as stored (up to date with the steps > jj)
                const double n_pv = rn[jn], n_ra = rn[a];
                const double n_q[4] = {rn[cb], rn[cb + 1], rn[cb + 2], rn[cb + 3]};
                const double inv = rsqrt_pivot(pv);
                badp |= !(pv > 0.0);
                const double f = (ra * inv) * inv;                    // L[jj][a] / L[jj][jj]
                // new row a = alpha (row a) - beta (row jj):  a < jj: (1, f) the update; a == jj: (1 / L[jj][jj], 0) the finished row;
                // a > jj: (1, 0) finished earlier, kept.  One unconditional store per step, no divergent code.
                const double alpha = a == jj ? inv : 1.0, beta = a < jj ? f : 0.0;
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) mv[i2] = __builtin_fma(-beta, q[i2], alpha * mv[i2]);
                mine[0] = mv[0]; mine[1] = mv[1]; mine[2] = mv[2]; mine[3] = mv[3];
                if (tid == 0) { piv[j0 + jj] = pv; dinv[jj] = inv; }
                // row jj - 1 after this step: minus (element jj - 1 of row jj / pivot) x row jj
                const double qsrc = q[jn & 3];
                const double qjm = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(qsrc), jn >> 2),
                                                    __builtin_amdgcn_readlane(__double2loint(qsrc), jn >> 2));   // (lane c < 4 holds columns 4c .. of row jj)
                const double fn = (qjm * inv) * inv;
                pv = __builtin_fma(-fn, qjm, n_pv);
                ra = __builtin_fma(-fn, ra, n_ra);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) q[i2] = __builtin_fma(-fn, q[i2], n_q[i2]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next step's look-ahead reads what other lanes wrote in this one
                __builtin_amdgcn_wave_barrier();
            }
            if (badp && tid == 0) s_bad = 1;
        }
        __syncthreads();
        PSTAMP(Tb = __builtin_amdgcn_s_memtime(); Td += Tb - Ta;)
        if (j0 == 0) break;
        // panel: column q of rows j0 .. j0 + 15, the 16 x 16 triangular solve right-looking, four lanes per column (lane s of the quad
        // keeps elements 4 i + s): per step the owner scales its element, the quad gets it by a DPP broadcast, every lane updates its
        // elements above -- ~10 instructions per step and lane instead of a 136-term chain in one thread
        {
            const int sq = tid & 3;
            // the coefficients a lane needs -- L[jj][4 i + sq] of the diagonal block for 4 i + sq < jj, 30 of them -- and the reciprocal
            // pivots, all requested before the first step: the 16 steps are then register arithmetic and DPP moves only
            double cf[16][4], dv[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                dv[jj] = dinv[jj];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2)
                    if (4 * i2 < jj) cf[jj][i2] = Pm[(j0 + jj) * LD + j0 + 4 * i2 + sq];
            }
            for (int q = tid >> 2; q < ((j0 + 63) & ~63); q += 64) {
                const int qc = q < j0 ? q : j0 - 1;                       // (whole quads stay active for the DPP moves; the stores are masked)
                double wv[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) wv[i2] = Pm[(j0 + 4 * i2 + sq) * LD + qc];
#pragma unroll
                for (int jj = 15; jj >= 0; --jj) {
                    const double wj = quad_bcast(wv[jj >> 2] * dv[jj], jj & 3);      // element jj, finished, from its owner lane
                    if (sq == (jj & 3)) wv[jj >> 2] = wj;
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) {
                        if (4 * i2 + 3 < jj) wv[i2] = __builtin_fma(-cf[jj][i2], wj, wv[i2]);                      // elements 4 i2 + sq < jj for every lane
                        else if (4 * i2 < jj) wv[i2] = (4 * i2 + sq < jj) ? __builtin_fma(-cf[jj][i2], wj, wv[i2]) : wv[i2];
                    }
                }
                if (q < j0) {
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) Pm[(j0 + 4 * i2 + sq) * LD + q] = wv[i2];
                }
            }
        }
        __syncthreads();
        PSTAMP(Ta = __builtin_amdgcn_s_memtime(); Tp += Ta - Tb;)
        // trailing update of the 16 x 16 blocks (kb, qb), qb <= kb < jb: P -= W' W with W = rows j0 .. j0 + 15, on the FP64 matrix
        // cores (v_mfma_f64_16x16x4_f64: A[i][k = g], B[k = g][col = i], element r of C at row g + 4 r, col i), one block per wave and
        // trip; diagonal blocks are updated whole (their upper half is scratch: the diagonal phase mirrors the lower half in)
        {
            const int wave = tid >> 6, li = tid & 15, lg = (tid >> 4) & 3;
            for (int t = wave; t < jb * (jb + 1) / 2; t += 4) {
                int kb = 0;
                while ((kb + 1) * (kb + 2) / 2 <= t) ++kb;
                const int qb = t - kb * (kb + 1) / 2;
                double *cp = Pm + (16 * kb + lg) * LD + 16 * qb + li;
                f64x4m acc = {cp[0], cp[4 * LD], cp[8 * LD], cp[12 * LD]};
                const double *wa = Pm + (j0 + lg) * LD + 16 * kb + li, *wb = Pm + (j0 + lg) * LD + 16 * qb + li;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-wa[4 * t4 * LD], wb[4 * t4 * LD], acc, 0, 0, 0);
                cp[0] = acc[0]; cp[4 * LD] = acc[1]; cp[8 * LD] = acc[2]; cp[12 * LD] = acc[3];
            }
        }
        __syncthreads();
        PSTAMP(Tt += __builtin_amdgcn_s_memtime() - Ta;)
    }
    PSTAMP(Ta = __builtin_amdgcn_s_memtime();)
    // ---- log det P = sum log pivot; L out
    if (tid < 64) {
        double lg = 0.0;
        for (int a = tid; a < DP; a += 64) lg += log(piv[a]);
        for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o);
        if (tid == 0) small[(int64_t)bid * NS + 3] = s_bad ? NAN : lg;
    } else if (tid < 128) {      // (the second wave, meanwhile)
        const double lm = log_mv_gamma_wave(0.5 * ((N == 0.0) ? v0 : v1), D, tid - 64);
        if (tid == 64) small[(int64_t)bid * NS + 4] = lm;
    }
    if constexpr (!PAIRS) {
        double *F = A.fac + (int64_t)row * DP * DP;
        for (int e = tid; e < DP * DP; e += 256) {
            const int a = e / DP, b = e - a * DP;
            F[e] = b <= a ? Pm[a * LD + b] : 0.0;
        }
    }
#ifdef DPMM_POST_STAMPS          // diagnostic build: phase cycles of two workgroups overwrite their scalars (scripts/post_stamps.py)
    __syncthreads();
    if (tid == 0 && bid == 5) { double *o = small + 5 * NS; o[0] = (double)Tf; o[1] = (double)Td; o[2] = (double)Tp; o[3] = (double)Tt; }
    if (tid == 0 && bid == 6) { double *o = small + 6 * NS; o[0] = (double)(__builtin_amdgcn_s_memtime() - Ta); o[1] = (double)(__builtin_amdgcn_s_memtime() - T0); }
#endif
}
template <bool PAIRS>
__global__ __launch_bounds__(256) void niw_post_lds_kernel(NiwMasterArgs A, const int32_t *__restrict__ jobs, const double *__restrict__ rows,
                                                           double *__restrict__ small) {
    niw_post_lds_body<PAIRS>(A, jobs, rows, small, (int)blockIdx.x);
}
// the 3 njobs posteriors of a statistics pass and npairs pooled pairs of the same pass in ONE launch (the pairs behind the posteriors in the
// grid): the pair jobs name clusters and read the rows of the pass, so they depend on nothing the posterior workgroups write
// (round 6, DPMM_OPT_CHAIN_FUSION bit 16) ... and, behind the pairs in the grid, the standard normals of the draws that follow this launch on the
// same stream (niw_noise_kernel's job, `nsplit` workgroups per matrix): the posteriors keep 96 of 256 compute units busy for 27 us, the normals
// fit beside them.  On a second stream beside the sweep they competed with the sweep for two thirds of its duration, and the draws had to wait
// for that stream's event in front of them -- a barrier packet between posteriors and draws, 6 us of the chain that does not shrink with n.
__global__ __launch_bounds__(256) void niw_post_both_kernel(NiwMasterArgs A, const int32_t *__restrict__ jobs, const double *__restrict__ rows,
                                                            double *__restrict__ small, int nposts, const int32_t *__restrict__ pair_jobs,
                                                            double *__restrict__ pair_small, int npairs, uint32_t noise_epoch, double *__restrict__ noise_Y, int nsplit) {
    if ((int)blockIdx.x < nposts) niw_post_lds_body<false>(A, jobs, rows, small, (int)blockIdx.x);
    else if ((int)blockIdx.x < nposts + npairs) niw_post_lds_body<true>(A, pair_jobs, rows, pair_small, (int)blockIdx.x - nposts);
    else {
        const int nb = (int)blockIdx.x - nposts - npairs, mat = nb / nsplit, part = nb - mat * nsplit;
        const int DP = A.DP, HP = DP / 2;
        double *Y = noise_Y + (int64_t)mat * DP * DP;
        for (int p2 = part * 256 + threadIdx.x; p2 < DP * HP; p2 += nsplit * 256) {
            const int r = p2 / HP, c = 2 * (p2 - r * HP);
            double n0, n1;
            bartlett_pair(A, (uint64_t)mat, noise_epoch, r, c, n0, n1);
            Y[(int64_t)r * DP + c] = n0; Y[(int64_t)r * DP + c + 1] = n1;
        }
    }
}
size_t niw_post_lds_bytes(int DP) { return sizeof(double) * ((size_t)DP * (DP + 1) + 3 * (size_t)DP + 16); }
constexpr int NIW_POST_LDS_MAXDP = 128;

bool niw_master_can_fuse_pairs(const NiwMasterArgs &a) { return a.DP <= NIW_POST_LDS_MAXDP; }
// posteriors of njobs clusters + npairs pooled pairs (CLUSTER indices of the same pass) in one launch; D <= 128 only
hipError_t launch_niw_master_posterior_pairs(const NiwMasterArgs &a, const int32_t *jobs, int njobs, const double *rows, double *small,
                                             const int32_t *cluster_pairs, int npairs, double *pair_small, hipStream_t s,
                                             int noise_nmat, uint32_t noise_epoch, double *noise_Y) {
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void *)niw_post_both_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_post_lds_bytes(NIW_POST_LDS_MAXDP)); attr = true; }
    const int nsplit = a.DP >= 128 ? 8 : (a.DP >= 48 ? 4 : 1);      // (workgroups per matrix, as launch_niw_master_noise)
    const int nnoise = (noise_Y && noise_nmat > 0) ? noise_nmat * nsplit : 0;
    DPMM_LAUNCH(niw_post_both_kernel, dim3(3 * njobs + npairs + nnoise), dim3(256), niw_post_lds_bytes(a.DP), s, a, jobs, rows, small, 3 * njobs,
                       cluster_pairs, pair_small, npairs, noise_epoch, noise_Y, nsplit);
    return hipGetLastError();
}
hipError_t launch_niw_master_pairs(const NiwMasterArgs &a, const int32_t *pairs, int n, double *scratch, double *small, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (a.DP <= NIW_POST_LDS_MAXDP) {
        static bool attr = false;
        if (!attr) { hipFuncSetAttribute((const void *)niw_post_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_post_lds_bytes(NIW_POST_LDS_MAXDP)); attr = true; }
        DPMM_LAUNCH(niw_post_lds_kernel<true>, dim3(n), dim3(256), niw_post_lds_bytes(a.DP), s, a, pairs, (const double *)nullptr, small);
        return hipGetLastError();
    }
    DPMM_LAUNCH(niw_form_pair_kernel, dim3(n, a.DP >= 64 ? 16 : 1), dim3(256), 0, s, a, pairs, scratch, small);
    NiwMasterArgs b = a;
    b.fac = scratch;                                  // the factorisation kernel in "matrix blockIdx.x of fac" mode
    DPMM_LAUNCH(niw_chol_kernel, dim3(n), dim3(256), niw_master_lds_bytes(a.DP), s, b, (const int32_t *)nullptr, small);
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
    DPMM_LAUNCH(niw_rows_gather_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, rows_store, slots, n, 2 * stride, dst);
    return hipGetLastError();
}

size_t niw_master_lds_bytes(int DP) { return sizeof(double) * ((size_t)34 * DP + 16 * 17 + 256); }      // draw: T 16 DP, Ld, xi DP, Lp 17 DP (the factorisation needs less)

hipError_t launch_niw_master_posterior(const NiwMasterArgs &a, const int32_t *jobs, int njobs, const double *rows, double *small, hipStream_t s) {
    if (njobs <= 0) return hipSuccess;
    if (a.DP <= NIW_POST_LDS_MAXDP) {
        static bool attr_l = false;
        if (!attr_l) { hipFuncSetAttribute((const void *)niw_post_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_post_lds_bytes(NIW_POST_LDS_MAXDP)); attr_l = true; }
        DPMM_LAUNCH(niw_post_lds_kernel<false>, dim3(3 * njobs), dim3(256), niw_post_lds_bytes(a.DP), s, a, jobs, rows, small);
        return hipGetLastError();
    }
    DPMM_LAUNCH(niw_form_kernel, dim3(3 * njobs, a.DP >= 64 ? 16 : 1), dim3(256), 0, s, a, jobs, rows, small);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void *)niw_chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_master_lds_bytes(DPMM_MASTER_MAXD));
        hipFuncSetAttribute((const void *)niw_draw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)niw_master_lds_bytes(DPMM_MASTER_MAXD));
        attr = true;
    }
    DPMM_LAUNCH(niw_chol_kernel, dim3(3 * njobs), dim3(256), niw_master_lds_bytes(a.DP), s, a, jobs, small);
    return hipGetLastError();
}

// what: bit 0 = the draws (Y, mu_draw, logdet_sigma from the posteriors), bit 1 = the hand-over to the sweep kernels (needs lr / wts),
// bit 2 = the normals of this epoch are in Y already (launch_niw_master_noise).
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
    if (what & 1) DPMM_LAUNCH(niw_draw_kernel, dim3(3 * K), dim3(256), niw_master_lds_bytes(a.DP), s, a, slot_of_cluster, epoch, Y, logdet_sigma, (what & 4) ? 1 : 0);
    if ((what & 2) && (what & 8)) {
        // the hand-over partitioned by role (+ the three-plane images when what & 16): workgroups per role from the number of items
        const int nmat = 3 * K, NP = NB * (NB + 1) / 2, DPs = 16 * NB;
        const bool tl = tail != nullptr && a.D >= 4;
        auto blocks = [](int64_t items, int cap) -> int { return (int)std::max<int64_t>(1, std::min<int64_t>((items + 255) / 256, cap)); };
        PackRoles R;
        R.first[0] = 0;
        R.first[1] = R.first[0] + blocks((int64_t)nmat * NP * 64, 2048);
        R.first[2] = R.first[1] + blocks((int64_t)nmat * (1 + DPs) + (tl ? (int64_t)((K + 1) / 2) * 32 + (int64_t)K * 16 : 0), 512);
        const bool img = tl && NB == 4;
        R.first[3] = R.first[2] + (img ? blocks((int64_t)K * REFB_WORDS, 1024) : 0);
        const bool b3 = img && (what & 16);
        R.first[4] = R.first[3] + (b3 ? blocks((int64_t)2 * K * REFB_WORDS, 2048) : 0);
        R.first[5] = R.first[4] + (b3 ? blocks((int64_t)2 * K * 256, 256) : 0);
        const bool pb = b3 && (what & 32) && K >= 2 && K <= PB_MAXK;
        R.first[6] = R.first[5] + (pb ? K : 0);
        DPMM_LAUNCH(niw_master_pack_roles_kernel, dim3(R.first[6]), dim3(256), 0, s, Y, a.mu_draw, logdet_sigma, lr, wts, Rp, mup, cst, tail, a.D, a.DP, NB,
                    nmat, work, R, b3 ? 1 : 0);
    } else if (what & 2) DPMM_LAUNCH(niw_master_pack_kernel, dim3(512), dim3(256), 0, s, Y, a.mu_draw, logdet_sigma, lr, wts, Rp, mup, cst, tail, a.D, a.DP, NB,
                                     3 * K, work);
    return hipGetLastError();
}

}  // namespace dpmm
