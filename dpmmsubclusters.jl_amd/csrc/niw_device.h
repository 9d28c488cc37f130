// niw_device.h -- device helpers shared by the NIW sweep kernels (niw_sweep.hip) and the lean / sub-label kernels (niw_lean.hip):
// vector typedefs, the 4-row tail screen, the cluster-per-lane ball test, the record layout behind `tail`, bf16 packing.  gfx950 only.
#pragma once
#include "dpmm_device.h"
#include "dpmm_kernels.h"

namespace dpmm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NB>
__host__ __device__ constexpr int pair_base(int bi) {
    return bi * NB - (bi * (bi - 1)) / 2;
}

// ---------------------------------------------------------------------------------------
// Tail screen (shared by both sweep kernels).  Rows D-4..D-1 of y = R z need the last four features only (R upper
// triangular), so q >= |T_k (x_t - m_t)|^2 with the 4x4 tail factor T_k.  The 15 constants of a cluster
// {T00 T01 T02 T03 | T11 T12 T13 T22 | T23 T33 m0 m1 | m2 m3 cst} are wave-uniform: read through the constant address
// space they arrive by scalar loads in SGPRs.  Lane = point: `xt` holds the four tail features of the lane's point,
// `thr` = reference value - margin (+inf for lanes without a point).
// Records are stored for PAIRS of clusters, element i of clusters 2p and 2p+1 side by side ([pair][16][2]): one packed-f32
// instruction (v_pk_*) then serves both clusters.  Instruction count is what matters here: a wave that shares its SIMD with
// an MFMA-streaming wave issues about one instruction per two MFMAs (scripts/microbench/issue_overlap.hip), so the phases of
// the two resident waves do not overlap -- every VALU / SALU instruction saved is ~5 cycles of SIMD time.  For the same
// reason the next record is NOT prefetched (it would cost 32 more SGPRs, i.e. copies or spills): while this wave waits for
// its scalar load the other wave runs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct TailPair { f32x2 v[16]; };
__device__ __forceinline__ TailPair tail_load_pair(const float *tail, int pair) {
    typedef const float __attribute__((address_space(4))) *cfp4;
    const cfp4 P = (cfp4)(tail + (size_t)pair * 32);
    TailPair T;
#pragma unroll
    for (int q = 0; q < 16; ++q) T.v[q] = (f32x2){P[2 * q], P[2 * q + 1]};
    return T;
}
// bit 0 / bit 1: cluster 2p / 2p+1 is below `thr` for every lane of the wave (lanes without a point carry thr = +inf)
__device__ __forceinline__ unsigned tail_pair_far(const TailPair &T, const f32x4 &xt, float thr) {
    const f32x2 x0 = (f32x2){xt.x, xt.x}, x1 = (f32x2){xt.y, xt.y}, x2 = (f32x2){xt.z, xt.z}, x3 = (f32x2){xt.w, xt.w};
    const f32x2 z0 = x0 - T.v[10], z1 = x1 - T.v[11], z2 = x2 - T.v[12], z3 = x3 - T.v[13];
    const f32x2 y3 = T.v[9] * z3;
    const f32x2 y2 = __builtin_elementwise_fma(T.v[7], z2, T.v[8] * z3);
    const f32x2 y1 = __builtin_elementwise_fma(T.v[4], z1, __builtin_elementwise_fma(T.v[5], z2, T.v[6] * z3));
    const f32x2 y0 = __builtin_elementwise_fma(T.v[0], z0, __builtin_elementwise_fma(T.v[1], z1, __builtin_elementwise_fma(T.v[2], z2, T.v[3] * z3)));
    f32x2 q4 = y3 * y3;
    q4 = __builtin_elementwise_fma(y2, y2, q4); q4 = __builtin_elementwise_fma(y1, y1, q4); q4 = __builtin_elementwise_fma(y0, y0, q4);
    const f32x2 ub = __builtin_elementwise_fma((f32x2){-0.5f, -0.5f}, q4, T.v[14]);
    const unsigned fa = (__ballot(ub.x < thr) == ~0ull) ? 1u : 0u;
    const unsigned fb = (__ballot(ub.y < thr) == ~0ull) ? 2u : 0u;
    return fa | fb;
}

// Ball test in front of the per-point tail screen, lane = CLUSTER.  The wave's points sit in a ball around the tail mean c of its
// reference cluster (radius r = max_i |x_i,tail - c|, one wave reduction); for every x in that ball
//   |T_k (x - m_k)| >= |T_k (c - m_k)| - |T_k|_2 r,
// so cst_k - 1/2 max(0, |T_k (c - m_k)| - |T_k|_F r)^2 bounds a_k for ALL points of the wave at once: lane j tests cluster j against the
// wave's lowest threshold -- ~60 instructions for 64 clusters, where the per-point screen spends ~22 per PAIR of clusters.  On
// label-homogeneous waves of well-separated data it clears nearly every cluster; whatever is left goes to the per-point screens.
// Per-cluster records [K][16] = { m0 m1 m2 m3 | T00 T01 T02 T03 | T11 T12 T13 T22 | T23 T33 b cst } behind the pair records, b >= |T|_2 a
// certified bound of the spectral norm (tail_opnorm_bound: within 19 %; the Frobenius norm, used first, can be twice the norm).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
// max / min over the 64 lanes (NaN operands are ignored by v_max / v_min), wave-uniform result
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f32<0xB1, 0xF>(v));       // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f32<0x4E, 0xF>(v));       // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f32<0x141, 0xF>(v));      // row_half_mirror
    v = fmaxf(v, dpp_f32<0x140, 0xF>(v));      // row_mirror: every lane holds the maximum of its row of 16
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v));      // row_bcast15 -> rows 1, 3
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v));      // row_bcast31 -> rows 2, 3: lane 63 holds the maximum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_min_f32(float v) {
    v = fminf(v, dpp_f32<0xB1, 0xF>(v));
    v = fminf(v, dpp_f32<0x4E, 0xF>(v));
    v = fminf(v, dpp_f32<0x141, 0xF>(v));
    v = fminf(v, dpp_f32<0x140, 0xF>(v));
    v = fminf(v, dpp_f32<0x142, 0xA>(v));
    v = fminf(v, dpp_f32<0x143, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__host__ __device__ __forceinline__ const float *ball_records(const float *tail, int K) { return tail + 32 * ((K + 1) >> 1); }
struct BallWave { f32x4 c; float r, thr; bool ok; };
// the wave's ball: centre = tail mean of reference cluster k0 (wave-uniform), radius over the valid lanes, lowest threshold.  Not usable
// (ok = false) when a point's tail features or threshold are not finite: such a point is covered by no ball, and the per-point
// screens keep every cluster for it.
__device__ __forceinline__ BallWave ball_of_wave(const float *tail, int K, int k0, const f32x4 &xt, float my_thr, bool valid) {
    typedef const float __attribute__((address_space(4))) *cfp4;
    const cfp4 P = (cfp4)(ball_records(tail, K) + 16 * (size_t)k0);
    BallWave B;
    B.c = (f32x4){P[0], P[1], P[2], P[3]};
    const f32x4 dx = xt - B.c;
    float r2 = dx.x * dx.x;
    r2 = __builtin_fmaf(dx.y, dx.y, r2); r2 = __builtin_fmaf(dx.z, dx.z, r2); r2 = __builtin_fmaf(dx.w, dx.w, r2);
    B.ok = __ballot(valid && !(r2 < INFINITY && my_thr == my_thr)) == 0ull;
    B.r = __builtin_amdgcn_sqrtf(wave_max_f32(valid ? r2 : 0.f)) * 1.00001f;      // (v_sqrt_f32, 1 ulp: covered by the slack factors)
    B.thr = wave_min_f32(my_thr);
    return B;
}
// lanes j: cluster base + j is below the wave's lowest threshold for every point of the ball
// (rec: this lane's record -- in global memory or in a workgroup's LDS copy; in_range: the lane names a cluster)
__device__ __forceinline__ unsigned long long ball_far_rec(const float *rec, bool in_range, const BallWave &B);
__device__ __forceinline__ unsigned long long ball_far(const float *tail, int K, int base, int lane, const BallWave &B) {
    const int j = base + lane;
    return ball_far_rec(ball_records(tail, K) + 16 * (size_t)(j < K ? j : 0), j < K, B);
}
__device__ __forceinline__ unsigned long long ball_far_rec(const float *rec, bool in_range, const BallWave &B) {
    const f32x4 m = *reinterpret_cast<const f32x4 *>(rec), t0 = *reinterpret_cast<const f32x4 *>(rec + 4);
    const f32x4 t1 = *reinterpret_cast<const f32x4 *>(rec + 8), t2 = *reinterpret_cast<const f32x4 *>(rec + 12);
    const f32x4 d = B.c - m;
    const float y3 = t2.y * d.w;
    const float y2 = __builtin_fmaf(t1.w, d.z, t2.x * d.w);
    const float y1 = __builtin_fmaf(t1.x, d.y, __builtin_fmaf(t1.y, d.z, t1.z * d.w));
    const float y0 = __builtin_fmaf(t0.x, d.x, __builtin_fmaf(t0.y, d.y, __builtin_fmaf(t0.z, d.z, t0.w * d.w)));
    float qn = y3 * y3;
    qn = __builtin_fmaf(y2, y2, qn); qn = __builtin_fmaf(y1, y1, qn); qn = __builtin_fmaf(y0, y0, qn);
    float dn = d.x * d.x;
    dn = __builtin_fmaf(d.y, d.y, dn); dn = __builtin_fmaf(d.z, d.z, dn); dn = __builtin_fmaf(d.w, d.w, dn);
    // rounding slack: |T d| is computed to ~1e-6 |T|_F |d|; the margin of the screen (tens of nats) dwarfs it anyway
    const float lb = fmaxf(__builtin_fmaf(-t2.z, __builtin_fmaf(1e-5f, __builtin_amdgcn_sqrtf(dn), B.r), __builtin_amdgcn_sqrtf(qn) * 0.99999f), 0.f);
    const float ub = __builtin_fmaf(-0.5f * lb, lb, t2.w);
    return __ballot(in_range && ub < B.thr);
}

// Reference BRACKET (D in 33 .. 64).  On a wave whose points all carried label k0 the cluster-level value a_k0(x) = cst - q(x) / 2,
// q = |R (x - mu)|^2, usually decides nothing: every other cluster is excluded by the screens and the draw returns k0 whatever the value
// is.  The screens only need a LOWER bound of it.  Two bf16 matrix passes give a certified one at ~1/7 of the Float32 evaluation's cycles:
//   y^ = R~ z~ (R~, z~ = bf16 round-to-nearest-even of R and of the Float32 z = x - mu; bf16 products are exact in the Float32
//   accumulator),  e^ = |R~| |z~|.
// bf16 carries 8 significand bits: ONE rounding has unit round-off u = 2^-8, |R - R~| <= u |R~| and |z - z~| <= u |z~| (half an ulp of
// the operand's binade, and the rounded value is never below that binade's base).  BOTH operands are rounded:
//   |y_i - y^_i| <= sum_j |R - R~||z| + |R~||z - z~| <= sum_j u |R~| (1 + u) |z~| + u |R~||z~| = (2u + u^2) e^_i = 0.0078278 e^_i
// (worst case R = z = 1 + 2^-8 -> R~ = z~ = 1: y - y^ = 0.0078278).  Float32 accumulation of the 64 exact products in y^ and e^ and the
// rounding of the Float32 evaluation this bracket stands in for (another summation order of the same 64 terms, |R||z| <= (1 + u)^2
// e^) add 3 * 64 * 2^-24 = 1.2e-5:
//   |y_i (as the Float32 evaluation computes it)| <= |y^_i| + REFB_C e^_i,   REFB_C = 0.00785 > 0.0078278 + 0.0000115,
//   q <= sum_i (|y^_i| + REFB_C e^_i)^2 (1 + 1e-4) =: q_hi   (the lane's four rows per block, then the ones-MFMA sum over the four row
// groups; the factor covers the ~30 Float32 roundings of the two sums of squares).  Subnormal z that the conversion may flush are an
// absolute error of 2^-126 |R~| per term -- nothing next to the 1e-20 added at the end for any factor the Float32 evaluation itself can
// handle.  Checked per point on adversarial operands (every entry just below a bf16 midpoint, displaced trailing features, far
// outliers) by tests/test_gpu_niw.py::test_reference_bracket_is_an_upper_bound through dpmm_debug_ref_bracket.
// (Round 3 shipped REFB_C = 0.00395 = u (1 + 2u) + ..., i.e. ONE rounding of half the size: q_hi < q on exactly those operands.  It changed
// no label -- the screens' 50-nat margin has ~32 nats of slack -- but it was not a bound.)
// On the bench's clusters (condition number 35 000) q_hi - q is ~14 of q ~ 64: the thresholds move by a few nats of a 50-nat margin.
// A non-finite x or parameter makes q_hi non-finite, every screen comparison false, and the wave takes the Float32 path.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr float REFB_C = 0.00785f;
__host__ __device__ __forceinline__ const uint32_t *refb_records(const float *tail, int K) {
    return reinterpret_cast<const uint32_t *>(ball_records(tail, K) + 16 * (size_t)K);
}
__device__ __forceinline__ uint32_t pack_bf16_pair(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
}

// bf16 BOTTOM screen (see the comment block on the bf16 screens in niw_sweep.hip): a certified lower bound of the last block row's part of the
// quadratic form of a candidate cluster from two bf16 matrix passes; true when it excludes the cluster for every point of the wave.
template <int NG>
__device__ __forceinline__ bool bf16_bottom_excludes(const u32x4_t a, const f32x4 (&x3)[NG], const f32x4 m4, float cst, const float (&thr)[NG]) {
    static_assert(NG % 2 == 0, "point groups are taken two at a time");
    const u32x4_t absm = (u32x4_t){0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu};
    const u32x4_t aa = a & absm;              // a: fragment 5 of the cluster's bf16 image -- block row 3 x features 32 .. 63 (zero for 32 .. 47)
    bool skip = true;
#pragma unroll
    for (int n0 = 0; n0 < NG; n0 += 2) {
        f32x4 y[2], e[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 z = x3[n0 + h] - m4;
            const u32x4_t zb = (u32x4_t){0u, 0u, pack_bf16_pair(z.x, z.y), pack_bf16_pair(z.z, z.w)};      // k-slots 0 .. 3: features 32 .. 47, zero rows of the fragment
            y[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, zb), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            e[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aa), __builtin_bit_cast(bf16x8_t, zb & absm), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float ql = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = fmaxf(__builtin_fmaf(-REFB_C, e[h][r], fabsf(y[h][r])), 0.f);
                ql = __builtin_fmaf(t, t, ql);
            }
            // (a NaN anywhere makes the comparison false: not excluded.  thr = +inf for columns without a point)
            unsigned long long mk = __ballot(__builtin_fmaf(-0.5f * 0.9999f, ql, cst) < thr[n0 + h]);
            mk |= mk >> 32;
            mk |= mk >> 16;
            skip = skip && ((mk & 0xFFFFull) == 0xFFFFull);     // every point: one of its four row-group lanes proves the bound
        }
    }
    return skip;
}


// ---------------------------------------------------------------------------------------
// The direction screen's matrix product and bounds (niw_sweep.hip: direction_far describes the bound) on operands that exist: zb = the bf16 pairs of
// z0 = x - mu_k0 in the x registers' layout ([point group][32-feature slice]), nz = |z0| per point (direction_norm of the lane's partial sum of
// squares).  niw_sweep_direct_kernel<.., DIR> forms both from x; niw_lean_kernel passes plane h of its three-plane split -- the same
// subtraction, the same v_cvt_pk_bf16_f32 -- and the norm it accumulated during the conversion.  Returns the mask of excluded clusters (bit k).
__device__ __forceinline__ float direction_norm(float part) {      // part: the lane's sum of squares over its 16 features of the point
    const f32x4 tot = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, part, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);      // |z0|^2 of point (n, lane & 15)
    return __builtin_sqrtf(tot[0]) * 1.00001f;
}
template <int NG>
__device__ __forceinline__ unsigned long long direction_far_core(const uint32_t *__restrict__ frag, const float *__restrict__ cons, const u32x4_t (&zb)[NG][2],
                                                                 const float (&nz)[NG], const float (&thr)[NG], int lane, int g, int K) {
    unsigned long long far = 0ull;
    const int nblk = (K + 15) >> 4;
    for (int blk = 0; blk < nblk; ++blk) {
        const u32x4_t a0 = reinterpret_cast<const u32x4_t *>(frag)[(2 * blk) * 64 + lane], a1 = reinterpret_cast<const u32x4_t *>(frag)[(2 * blk + 1) * 64 + lane];
        const f32x4 cB = *reinterpret_cast<const f32x4 *>(cons + 16 * blk + 4 * g), cE = *reinterpret_cast<const f32x4 *>(cons + 64 + 16 * blk + 4 * g),
                    cK = *reinterpret_cast<const f32x4 *>(cons + 128 + 16 * blk + 4 * g);
        bool ok[4] = {true, true, true, true};
        const f32x4 tau = cB * 0.02f;
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, zb[n][0]), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, zb[n][1]), sv, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {       // cluster 16 blk + 4 g + r against point (n, lane & 15); thr = +inf for a column without a point, NaN excludes nothing
                float t = __builtin_fmaf(-cE[r], nz[n], fabsf(sv[r] + cB[r]));
                t = t >= tau[r] ? t : 0.f;                                   // (NaN: 0)
                const float ub = __builtin_fmaf(-0.4995f * t, t, cK[r]);
                ok[r] = ok[r] && (ub < thr[n]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned long long m = __ballot(ok[r]);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (((m >> (16 * gg)) & 0xFFFFull) == 0xFFFFull) far |= 1ull << (16 * blk + 4 * gg + r);
        }
    }
    return K >= 64 ? far : far & ((1ull << K) - 1ull);
}

}  // namespace dpmm
