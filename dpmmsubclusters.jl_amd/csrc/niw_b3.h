// niw_b3.h -- device functions of the bf16 three-plane evaluation of the sub-cluster quadratic forms (D in 33 .. 64): shared by the kernels
// of niw_lean.hip and by the LIST instantiations of niw_sweep_direct_kernel (niw_sweep.hip), which finish the sub-labels of the spans they are
// handed in the same launch.  The arithmetic, the image layout and the invariant it keeps are described at the top of niw_lean.hip.
#pragma once
#include "dpmm_device.h"
#include "dpmm_kernels.h"
#include "niw_device.h"

namespace dpmm {

__host__ __device__ __forceinline__ const uint32_t *b3_images(const float *tail, int K) { return refb_records(tail, K) + (size_t)K * REFB_WORDS; }
__host__ __device__ __forceinline__ const float *b3_offsets(const float *tail, int K) { return reinterpret_cast<const float *>(b3_images(tail, K) + (size_t)3 * K * B3_WORDS); }

__host__ __device__ __forceinline__ const float *pair_ball_table(const float *tail, int K) { return b3_offsets(tail, K) + (size_t)3 * K * B3_DVEC; }      // pd [K][K] | sn [K] (= tail + niw_pair_ball_offset(K), dpmm_kernels.h)

// One workgroup (256 threads) tabulates column j of the pair-ball table and s_j (niw_lean.hip: the test and its derivation).  elemR(row, col): R_j's
// Float32 element for col >= row; mean(k, c): the Float32 cluster-level mean the sweep subtracts.
//   s_j^2 <= max row sum of |R_j' R_j| (>= its largest eigenvalue): thread (tr, tc) forms the 4 x 4 block of P = R' R at rows 4 tr, columns 4 tc from
//   two 16-byte LDS reads per row of R (64 x 16 multiply-adds per thread), the 16 threads of a row block add their partial row sums;
//   D_k,j: wave w takes k = w, w + 4, ..: lane = row of R_j, the 64 differences mu_k - mu_j broadcast by v_readlane.
// Slack: a Float32 dot product of 64 terms errs by <= 64 * 2^-24 of the sum of its terms' magnitudes -- 1e-3 on s_j^2 covers the row sums (the
// errors of a row add up to <= 5e-4 of the largest eigenvalue), 1e-4 s_j |v| comes off D, which is itself rounded DOWN by 1e-4.
// (COLMAJOR: consecutive threads of the staging loop take consecutive ROWS of a column -- for a source stored column by column)
template <bool COLMAJOR, class ElemR, class Mean>
__device__ __forceinline__ void pair_ball_block(ElemR elemR, Mean mean, int K, int j, float *__restrict__ pd) {
    float *sn = pd + (size_t)K * K;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int LD = 68;                                  // (rows 16-byte aligned)
    __shared__ __attribute__((aligned(16))) float Rs[64 * LD];
    __shared__ float rowsum[64];
    __shared__ float s_sh;
    for (int e = tid; e < 4096; e += 256) {
        const int row = COLMAJOR ? (e & 63) : (e >> 6), col = COLMAJOR ? (e >> 6) : (e & 63);
        Rs[row * LD + col] = col >= row ? elemR(row, col) : 0.f;
    }
    __syncthreads();
    {
        const int tr = tid >> 4, tc = tid & 15;
        float p[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) p[a][b] = 0.f;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) {                      // (zeros below the diagonal: no case distinction)
            const f32x4 ra = *reinterpret_cast<const f32x4 *>(Rs + i * LD + 4 * tr), rb = *reinterpret_cast<const f32x4 *>(Rs + i * LD + 4 * tc);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) p[a][b] = __builtin_fmaf(ra[a], rb[b], p[a][b]);
        }
        float rs[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            rs[a] = fabsf(p[a][0]) + fabsf(p[a][1]) + fabsf(p[a][2]) + fabsf(p[a][3]);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) rs[a] += __shfl_xor(rs[a], o);      // over the 16 column blocks (consecutive lanes)
        }
        if (tc == 0) { rowsum[4 * tr] = rs[0]; rowsum[4 * tr + 1] = rs[1]; rowsum[4 * tr + 2] = rs[2]; rowsum[4 * tr + 3] = rs[3]; }
    }
    __syncthreads();
    if (w == 0) {
        const float mx = wave_max_f32(rowsum[lane]);
        if (lane == 0) { const float sj = __builtin_sqrtf(mx) * 1.001f; s_sh = sj; sn[j] = (mx == mx && mx < INFINITY) ? sj : INFINITY; }
    }
    __syncthreads();
    const float sj = s_sh;
    const float mj = mean(j, lane);
    for (int k0 = w; k0 < K; k0 += 16) {                     // four clusters per trip and wave (wave-uniform): one LDS read of R per column serves all four
        float v[4], y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int k = k0 + 4 * q; v[q] = (k < K ? mean(k, lane) : mj) - mj; y[q] = 0.f; }      // the subtraction the sweep's z = x - mu rests on
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            const float rc = Rs[lane * LD + c];             // (zeros left of the diagonal)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                y[q] = __builtin_fmaf(rc, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[q]), c)), y[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y2 = y[q] * y[q], v2 = v[q] * v[q];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { y2 += __shfl_xor(y2, o); v2 += __shfl_xor(v2, o); }
            const int k = k0 + 4 * q;
            if (lane == 0 && k < K) {
                float d = __builtin_sqrtf(y2) * 0.9999f - 1e-4f * sj * __builtin_sqrtf(v2);
                if (!(d > 0.f) || k == j) d = 0.f;            // (NaN, or nothing to gain: the test then never fires)
                pd[(size_t)k * K + j] = d;
            }
        }
    }
}

struct B3Z { u32x4_t p[4][2][3]; };      // [point group][32-feature slice][plane]: the B operands of the wave's 64 points (96 registers)

__device__ __forceinline__ void b3_split_pair(float a, float b, uint32_t &ph, uint32_t &pm, uint32_t &pl) {
    ph = pack_bf16_pair(a, b);
    const float ra = a - __uint_as_float(ph << 16), rb = b - __uint_as_float(ph & 0xffff0000u);        // exact
    pm = pack_bf16_pair(ra, rb);
    const float sa = ra - __uint_as_float(pm << 16), sb = rb - __uint_as_float(pm & 0xffff0000u);      // exact; fits 8 bits
    pl = pack_bf16_pair(sa, sb);
}
// z = x - mk for all four point groups (mk: a cluster-level mean in the x registers' layout), split into planes
// NZ: also the lane's sum of squares of z per point group, in direction_far's order of additions (the direction screen's |z0|, niw_device.h)
template <bool NZ = false>
__device__ __forceinline__ void b3_convert(const f32x4 (&x)[4][4], const f32x4 (&mk)[4], B3Z &Z, float *part = nullptr, int part_stride = 1) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float pn = 0.f;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const f32x4 lo = x[n][2 * sl] - mk[2 * sl], hi = x[n][2 * sl + 1] - mk[2 * sl + 1];
            if constexpr (NZ) {
                pn = __builtin_fmaf(lo.x, lo.x, pn); pn = __builtin_fmaf(lo.y, lo.y, pn); pn = __builtin_fmaf(lo.z, lo.z, pn); pn = __builtin_fmaf(lo.w, lo.w, pn);
                pn = __builtin_fmaf(hi.x, hi.x, pn); pn = __builtin_fmaf(hi.y, hi.y, pn); pn = __builtin_fmaf(hi.z, hi.z, pn); pn = __builtin_fmaf(hi.w, hi.w, pn);
            }
            uint32_t H[4], M[4], L[4];
            b3_split_pair(lo.x, lo.y, H[0], M[0], L[0]);
            b3_split_pair(lo.z, lo.w, H[1], M[1], L[1]);
            b3_split_pair(hi.x, hi.y, H[2], M[2], L[2]);
            b3_split_pair(hi.z, hi.w, H[3], M[3], L[3]);
            Z.p[n][sl][0] = (u32x4_t){H[0], H[1], H[2], H[3]};
            Z.p[n][sl][1] = (u32x4_t){M[0], M[1], M[2], M[3]};
            Z.p[n][sl][2] = (u32x4_t){L[0], L[1], L[2], L[3]};
        }
        if constexpr (NZ) part[n * part_stride] = pn;          // (niw_lean_kernel: LDS, so that the four sums are not four registers across the bracket)
    }
}
// Both sub-cluster values of the wave's points for cluster k (wave-uniform) from the planes of z = x - mu_k: bl / br = cst - |R_s z + d_s|^2 / 2
// for "this lane's point" (point lane & 15 of point group lane >> 4).  ONE pipeline over the eight row blocks of the two matrices (left
// 0..3, right 0..3): the fragments of a row block (two 32-feature slices for row blocks 0 and 1, one for 2 and 3; three planes each) and its
// four offsets are requested two row blocks ahead of their matrix instructions.
// (B3Head: the first two row blocks' fragments, their offsets and the two constants -- a caller with other work in front of the evaluation
// requests them there: b3_head)
struct B3Head { u32x4_t a[2][2][3]; f32x4 d[2]; float cl, cr; };
__device__ __forceinline__ B3Head b3_head(const float *__restrict__ tail, const float *__restrict__ cst, int K, int k, int lane, int g) {
    const u32x4_t *F = reinterpret_cast<const u32x4_t *>(b3_images(tail, K) + (size_t)(3 * k + 1) * B3_WORDS) + (unsigned)lane;
    const float *dvec = b3_offsets(tail, K) + (size_t)(3 * k + 1) * B3_DVEC;
    B3Head H;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int p = 0; p < 3; ++p) H.a[bi][s2][p] = F[64 * (6 * p + 2 * bi + s2)];
        H.d[bi] = *reinterpret_cast<const f32x4 *>(dvec + 16 * bi + 4u * (unsigned)g);
    }
    H.cl = cst[3 * k + 1]; H.cr = cst[3 * k + 2];
    return H;
}
__device__ __forceinline__ void b3_eval(const float *__restrict__ tail, int K, int k, const B3Z &Z, const B3Head &H, int lane, int g, float &bl, float &br) {
    const uint32_t *img = b3_images(tail, K) + (size_t)(3 * k + 1) * B3_WORDS;
    const float *dvec = b3_offsets(tail, K) + (size_t)(3 * k + 1) * B3_DVEC;
    const u32x4_t *F = reinterpret_cast<const u32x4_t *>(img) + (unsigned)lane;          // fragment f of plane p of matrix m: F[64 (18 m + 6 p + f)]
    constexpr int F0[4] = {0, 2, 4, 5};                                        // first fragment of a row block (refb_map's order: (0,0) (0,1) (1,0) (1,1) (2,1) (3,1))
    u32x4_t Af[8][2][3];                                                       // [block 4 m + bi][slice of the block][plane]; SSA values: nothing is copied
    f32x4 dv[8];
    auto load_block = [&](int b8) {
        const int m = b8 >> 2, bi = b8 & 3;
#pragma unroll
        for (int s2 = 0; s2 < (bi < 2 ? 2 : 1); ++s2)
#pragma unroll
            for (int p = 0; p < 3; ++p) Af[b8][s2][p] = F[64 * (18 * m + 6 * p + F0[bi] + s2)];
        dv[b8] = *reinterpret_cast<const f32x4 *>(dvec + B3_DVEC * m + 16 * bi + 4u * (unsigned)g);
    };
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int p = 0; p < 3; ++p) Af[bi][s2][p] = H.a[bi][s2][p];
        dv[bi] = H.d[bi];
    }
    const float cl = H.cl, cr = H.cr;
    auto terms = [&](f32x4 acc, const u32x4_t (&a)[3], const u32x4_t (&z)[3]) -> f32x4 {       // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[2]), __builtin_bit_cast(bf16x8_t, z[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[0]), __builtin_bit_cast(bf16x8_t, z[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[1]), __builtin_bit_cast(bf16x8_t, z[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[1]), __builtin_bit_cast(bf16x8_t, z[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[0]), __builtin_bit_cast(bf16x8_t, z[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[0]), __builtin_bit_cast(bf16x8_t, z[0]), acc, 0, 0, 0);
        return acc;
    };
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    float sel_l = 0.f, sel_r = 0.f;
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
        const int bi = b8 & 3;
        if (b8 + 2 < 8) load_block(b8 + 2);
        __builtin_amdgcn_sched_barrier(0);          // (the requests stay in front of this row block's matrix instructions; none of a later row block joins them)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 acc = dv[b8];
            if (bi < 2) { acc = terms(acc, Af[b8][0], Z.p[n][0]); acc = terms(acc, Af[b8][1], Z.p[n][1]); }
            else acc = terms(acc, Af[b8][0], Z.p[n][1]);
            q[n] = __builtin_fmaf(acc[0], acc[0], q[n]); q[n] = __builtin_fmaf(acc[1], acc[1], q[n]);
            q[n] = __builtin_fmaf(acc[2], acc[2], q[n]); q[n] = __builtin_fmaf(acc[3], acc[3], q[n]);
        }
        if (bi == 3) {                              // a matrix is complete: sum over the four row groups of a column (the ones-MFMA), this lane's point
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f32x4 tot = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, q[n], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (g == n) { if (b8 == 3) sel_l = tot[0]; else sel_r = tot[0]; }
                q[n] = 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    bl = __builtin_fmaf(-0.5f, sel_l, cl);
    br = __builtin_fmaf(-0.5f, sel_r, cr);
}
// the cluster-level mean of cluster k in the x registers' layout
__device__ __forceinline__ void b3_mean(const float *__restrict__ mup, int k, int g, f32x4 (&mk)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) mk[t] = *reinterpret_cast<const f32x4 *>(mup + (size_t)(3 * k) * 64 + 16 * t + 4u * (unsigned)g);
}
// the lane's 16 bytes of every (point group, 16-feature slice) of the B operand, as the sweep kernels hold x: point (n, ci) = p[16 n + ci]
// (Every load unconditional: a column without a point reads row 0, a slice beyond ldx the row's last four floats, and a select zeroes them --
// only on a wave that has such a lane.  With the load inside `cond ? load : 0` the compiler emitted sixteen exec-mask branches per tile, each
// behind a wait for its index shuffle, and 130 register clears.)
__device__ __forceinline__ void gather_x64(const float *__restrict__ X, int64_t ldx, int myp32, int ci, int g, f32x4 (&x)[4][4]) {
    const bool plain = ldx >= 64 && __ballot(myp32 < 0) == 0ull;      // (wave-uniform)
    int pn[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) pn[n] = __shfl(myp32, 16 * n + ci);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const float *row = X + (int64_t)(pn[n] >= 0 ? pn[n] : 0) * ldx;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = 16 * t + 4 * g;
            x[n][t] = *reinterpret_cast<const f32x4 *>(row + (e < ldx ? e : (int)ldx - 4));
        }
    }
    if (!plain) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (!(pn[n] >= 0 && 16 * t + 4 * g < ldx)) x[n][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace dpmm
