// dpmm_device.h -- device-side helpers shared by all kernels (gfx950 only).
//
// philox4x32_10 / u01 / exp_det are defined operation-for-operation like their
// counterparts in oracle/dpmm_oracle.c so that, given the same Float32 log-likelihood
// table, the categorical draw (reference: src/utils.jl:19-31) is bit-identical on the
// CPU oracle and on the GPU.  The translation unit is compiled with -ffp-contract=off;
// every fused multiply-add below is an explicit fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpmm {

constexpr int WAVE = 64;

enum : uint32_t { STREAM_SWEEP = 0, STREAM_INIT = 1, STREAM_SPLIT = 2, STREAM_RESET = 3 };

struct Philox4 {
    uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t idx, uint32_t epoch, uint32_t stream) {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = epoch, c3 = stream;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// (0, 1): the odd multiples of 2^-24 -- 23 random bits, never 0, never 1, every value exact in Float32.  Rounds 1-5 used [0, 1) with 24 bits:
// a uniform of exactly 0 (once in 2^24 draws: 0.6 points per sweep at N = 1e7) makes the inverse-CDF scan stop at INDEX 0 whatever that
// cluster's probability -- an artefact the reference's 53-bit rand() does not have in practice, and the one case the lean sweep had to hand on
// to the general kernel (a cold launch of 45 us in 45 % of the sweeps).  The oracle's u01 is the same function (oracle/dpmm_oracle.c).
__device__ __forceinline__ float u01(uint32_t r) { return (float)((r >> 8) | 1u) * (1.0f / 16777216.0f); }

// Deterministic expf for max-shifted arguments (x <= 0); see oracle/dpmm_oracle.c exp_det.
__device__ __forceinline__ float exp_det(float x) {
    if (!(x >= -86.0f)) return 0.0f;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 5040.0f;
    p = __builtin_fmaf(p, r, 1.0f / 720.0f);
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const uint32_t sb = (uint32_t)((int)n + 127) << 23;
    return p * __uint_as_float(sb);
}

__device__ __forceinline__ float nan_to_ninf(float a) { return (a != a) ? -INFINITY : a; }

// 2-way draw of create_subclusters_labels! (src/local_clusters_actions.jl:83-95 -> utils.jl:19-31
// with two columns).  Returns 0 (left, sub-label 1) or 1 (right, sub-label 2).
__device__ __forceinline__ int draw2(float b0, float b1, float u) {
    b0 = nan_to_ninf(b0);
    b1 = nan_to_ninf(b1);
    float m = -INFINITY;
    if (b0 > m) m = b0;
    if (b1 > m) m = b1;
    if (m == -INFINITY) return 0;
    const float p0 = exp_det(b0 - m), p1 = exp_det(b1 - m);
    float s = 0.0f;
    s += p0;
    s += p1;
    const float t = u * s;
    float cw = 0.0f;
    cw += p0;
    return (cw < t) ? 1 : 0;
}

// Certified upper bound of the spectral norm of an upper-triangular 4x4 matrix T = {T00 T01 T02 T03 | T11 T12 T13 | T22 T23 | T33}:
// ||T||_2^2 = lambda_max(G), G = T'T (symmetric positive semi-definite), and lambda_max(G) <= tr(G^4)^(1/4) <= 4^(1/4) lambda_max(G):
// within 19 % of the norm, where the Frobenius norm can be twice it.  Used by the ball test of the NIW sweeps (the pack kernels).
__device__ __forceinline__ float tail_opnorm_bound(const float (&t)[10]) {
    const double T[4][4] = {{t[0], t[1], t[2], t[3]}, {0., t[4], t[5], t[6]}, {0., 0., t[7], t[8]}, {0., 0., 0., t[9]}};
    double G[4][4], H[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { double s = 0.; for (int r = 0; r < 4; ++r) s += T[r][i] * T[r][j]; G[i][j] = s; }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { double s = 0.; for (int r = 0; r < 4; ++r) s += G[i][r] * G[r][j]; H[i][j] = s; }       // G^2
    double tr4 = 0.;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) tr4 += H[i][j] * H[j][i];                                          // tr(G^4) = ||G^2||_F^2
    const double fro2 = G[0][0] + G[1][1] + G[2][2] + G[3][3];
    double lam = sqrt(sqrt(tr4));
    if (!(lam <= fro2)) lam = fro2;                      // (never larger than the Frobenius bound; NaN / Inf fall back to it)
    return (float)(sqrt(lam) * 1.00001);
}

// bf16 bits of a Float32, round to nearest even (finite inputs; a NaN / Inf stays non-finite or becomes one: callers fall back on those)
__device__ __forceinline__ uint32_t bf16_rne_bits(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// bf16 image of a cluster-level factor R (upper triangular, D <= 64, NB = 4) for the reference BRACKET of the D <= 64 sweep
// (niw_sweep_direct_kernel): six A-operand fragments of v_mfma_f32_16x16x32_bf16, fragment f = (row block bi, feature slice s) in the
// order (0,0) (0,1) (1,0) (1,1) (2,1) (3,1) -- the others are zero for an upper-triangular matrix -- [6][64 lanes][4 dwords]: dword d of
// lane (i, g) holds k-slots 8g + 2d, 8g + 2d + 1 of row 16 bi + i, and k-slot j of lane group g is FEATURE 32 s + 4 g + j (j < 4) or
// 32 s + 16 + 4 g + (j - 4): exactly the features the sweep's x registers x[n][2s], x[n][2s+1] hold for that lane.
// REFB_WORDS dwords per cluster; element e of a cluster's image -> (row, the two feature indices of its halves)
__host__ __device__ __forceinline__ void refb_map(int e, int &row, int &col0, int &col1) {
    const int d = e & 3, lane = (e >> 2) & 63, f = e >> 8;
    const int bi = f < 2 ? 0 : (f < 4 ? 1 : f - 2), sl = f < 4 ? (f & 1) : 1;
    const int i = lane & 15, g = lane >> 4;
    row = 16 * bi + i;
    const int j0 = 2 * d;                  // (j0, j0 + 1) lie on the same side of 4
    col0 = 32 * sl + (j0 < 4 ? 4 * g + j0 : 16 + 4 * g + (j0 - 4));
    col1 = col0 + 1;
}

// THREE-PLANE bf16 split of a Float32 (the sub-cluster evaluations of niw_lean.hip): v = h + m + l exactly, h = bf16(v), m = bf16(v - h),
// l = bf16(v - h - m) (each round-to-nearest-even; the residuals are exact in Float32 and the third one fits 8 bits: 3 x 8 significand bits =
// the 24 of the Float32).  plane 0 / 1 / 2 -> bits of h / m / l.  Non-finite v: every plane non-finite.
__host__ __device__ __forceinline__ uint32_t bf16x3_plane_bits(float v, int plane) {
    auto rne = [](float f) -> uint32_t { uint32_t u; __builtin_memcpy(&u, &f, 4); return ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16) & 0xffffu; };
    auto up = [](uint32_t b) -> float { const uint32_t u = b << 16; float f; __builtin_memcpy(&f, &u, 4); return f; };
    const uint32_t h = rne(v);
    if (plane == 0) return h;
    const float r1 = v - up(h);
    const uint32_t m = rne(r1);
    if (plane == 1) return m;
    return rne(r1 - up(m));
}

}  // namespace dpmm
