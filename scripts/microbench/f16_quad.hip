// Dev microbenchmark AND record of an experiment that did not ship (round 4; DESIGN.md section 10, docs/experiments/r04_f16_split.patch):
// the D <= 64 quadratic form as a TWO-TERM fp16 SPLIT -- R 2^-f = Rh + Rl, (x - mu) 2^s = zh + zl, y' = Rh zh + Rh zl + Rl zh with the fp16
// matrix instruction (16 cycles for 32 features) instead of the Float32 one (32 cycles for 4): 72 + 4 matrix instructions per matrix and
// 64 points instead of 164 + 4, ~22 significand bits per operand.  This file times the two evaluations with NOTHING else in the kernel:
// every wave keeps one 64-point tile in registers and walks K matrices back to back, 2 workgroups of 4 waves per CU as the sweep runs.
// Measured on MI355X: Float32 6 455 cycles per evaluation and SIMD, fp16 split 2 641 (2.45x).  Inside niw_sweep_direct_kernel the same
// code gave 1.47 ms against 1.44 ms on the headline (overlap legs -3 %, K = 256 +7 %): with two waves per SIMD the tile is bound by its
// chain of memory latencies (x gather, fragments), not by the matrix pipe, and the conversion's vector instructions no longer hide under
// the other wave's matrix work.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I dpmmsubclusters.jl_amd/csrc scripts/microbench/f16_quad.hip -o scripts/microbench/f16_quad.bin
#include "niw_sweep.hip"
#include <cstdio>
#include <vector>
namespace dpmm {
void (*g_prelaunch)(void *) = nullptr;
void *g_prelaunch_arg = nullptr;
constexpr int F16_FRAGS = 12, F16_WORDS = F16_FRAGS * 256;      // per matrix: (block row, 32-feature slice) fragments in the bf16 bracket image's order, hi then lo plane each

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16_pair(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));      // v_cvt_pk_f16_f32: round to nearest even
}
// residual of the conversion: a - float(lo half of p), b - float(hi half of p)   (v_fma_mix_f32: the fp16 half is an operand, no conversion back)
__device__ __forceinline__ void f16_residual(uint32_t p, float a, float b, float &ra, float &rb) {
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p), "v"(b));
}
// largest-|feature| exponent of each of the wave's points (once per tile): byte n of the result for point (n, lane & 15), the same in its
// four lanes, as a signed byte (|x| < 2^e; e = 128 is stored as 127: at |x| >= 2^127 the quadratic form overflows Float32 anyway)
template <int NG>
__device__ __forceinline__ int f16_point_exponents(const f32x4 (&x)[NG][4]) {
    static_assert(NG <= 4, "four exponents to a register");
    int packed = 0;
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        float m = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) m = fmaxf(fmaxf(fabsf(x[n][t].x), fabsf(x[n][t].y)), fmaxf(m, fmaxf(fabsf(x[n][t].z), fabsf(x[n][t].w))));
        m = fmaxf(m, __shfl_xor(m, 16));                  // the point's other features sit in the lanes of the other three row groups
        m = fmaxf(m, __shfl_xor(m, 32));
        int e = __builtin_amdgcn_frexp_expf(m);           // |x| < 2^e   (0 for m = 0, Inf, NaN: those stay what they are)
        e = e > 127 ? 127 : e;
        packed |= (e & 0xff) << (8 * n);
    }
    return packed;
}
template <int NG>
__device__ __forceinline__ float quad_f16_stream(const float *__restrict__ Rm, const float *__restrict__ Rnext, const float *__restrict__ mup_next, int pe,
                                                 int xe, f32x4 (&rb0)[4], f32x4 (&mu)[4], const f32x4 (&x)[NG][4], int lane, int g,
                                                 float (&tot_all)[NG], int prio) {
    if (prio) __builtin_amdgcn_s_setprio(0);
    const int fe = (int)((unsigned)pe << 16) >> 16, me = pe >> 16;      // exponents of the factor image and of the largest |mean| (f16img_exponents)
    // block row 0 (fragments 0 .. 3: slice 0 hi / lo, slice 1 hi / lo) came with load_rb0 / the previous evaluation's prefetch; the rest now
    u32x4_t F[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) F[i] = *reinterpret_cast<const u32x4_t *>(Rm + (4 + i) * 256 + lane * 4);
    f32x4 nx0[4], mun[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { nx0[t] = rb0[t]; mun[t] = mu[t]; }
    __builtin_amdgcn_sched_barrier(0);                   // (the fragment requests stay HERE: the operand conversion below is what covers their L2 latency)
    float part[NG];
    int sc[NG];
    auto mm = [](const u32x4_t a, const u32x4_t b, const f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        // |x - mu| < 2^xe + 2^me <= 2^(max + 1): the scaled operand stays below 2^15 (fp16 holds 2^16 - 32), and keeps 22 bits as long as
        // the point's largest |x - mu| is above 2^-18 of that bound (beyond it the Float32 subtraction has no bits left either)
        __builtin_amdgcn_sched_barrier(0);               // (one group's conversion and matrix work at a time: pipelined across groups the kernel spills)
        int xen;                                          // (volatile: extracted HERE -- hoisted to the top of the tile the four values are spilled and each reload waits on vmcnt(0))
        asm volatile("v_bfe_i32 %0, %1, %2, 8" : "=v"(xen) : "v"(xe), "n"(8 * n));
        const int sh = 14 - (xen > me ? xen : me);
        sc[n] = sh;
        u32x4_t zh[2], zl[2];                             // [feature slice]
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const f32x4 z0 = x[n][2 * sl] - mu[2 * sl], z1 = x[n][2 * sl + 1] - mu[2 * sl + 1];
            const float v[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float a = __builtin_ldexpf(v[2 * d], sh), b = __builtin_ldexpf(v[2 * d + 1], sh);
                hw[d] = pack_f16_pair(a, b);
                float ra, rb;
                f16_residual(hw[d], a, b, ra, rb);
                lw[d] = pack_f16_pair(ra, rb);
            }
            zh[sl] = (u32x4_t){hw[0], hw[1], hw[2], hw[3]};
            zl[sl] = (u32x4_t){lw[0], lw[1], lw[2], lw[3]};
        }
        if (n == NG - 1 && Rnext) {                      // the next matrix's block row 0 and means: requested under the last group's matrix work
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                nx0[t] = *reinterpret_cast<const f32x4 *>(Rnext + t * 256 + lane * 4);
                mun[t] = *reinterpret_cast<const f32x4 *>(mup_next + 16 * t + 4 * g);
            }
        }
        // four independent accumulator chains, one per block row: rows 0 and 1 see both feature slices, rows 2 and 3 the second only
        const u32x4_t a00h = __builtin_bit_cast(u32x4_t, rb0[0]), a00l = __builtin_bit_cast(u32x4_t, rb0[1]);
        const u32x4_t a01h = __builtin_bit_cast(u32x4_t, rb0[2]), a01l = __builtin_bit_cast(u32x4_t, rb0[3]);
        f32x4 y0 = mm(a00h, zh[0], zero), y1 = mm(F[0], zh[0], zero), y2 = mm(F[4], zh[1], zero), y3 = mm(F[6], zh[1], zero);
        y0 = mm(a01h, zh[1], y0); y1 = mm(F[2], zh[1], y1); y2 = mm(F[4], zl[1], y2); y3 = mm(F[6], zl[1], y3);
        y0 = mm(a00h, zl[0], y0); y1 = mm(F[0], zl[0], y1); y2 = mm(F[5], zh[1], y2); y3 = mm(F[7], zh[1], y3);
        y0 = mm(a01h, zl[1], y0); y1 = mm(F[2], zl[1], y1);
        y0 = mm(a00l, zh[0], y0); y1 = mm(F[1], zh[0], y1);
        y0 = mm(a01l, zh[1], y0); y1 = mm(F[3], zh[1], y1);
        float pr = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) pr = __builtin_fmaf(y0[r], y0[r], pr);
#pragma unroll
        for (int r = 0; r < 4; ++r) pr = __builtin_fmaf(y1[r], y1[r], pr);
#pragma unroll
        for (int r = 0; r < 4; ++r) pr = __builtin_fmaf(y2[r], y2[r], pr);
#pragma unroll
        for (int r = 0; r < 4; ++r) pr = __builtin_fmaf(y3[r], y3[r], pr);
        part[n] = pr;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { rb0[t] = nx0[t]; mu[t] = mun[t]; }
    float sel = 0.f;
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        const f32x4 tot = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, part[n], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);      // sum over the four row groups of a column
        tot_all[n] = __builtin_ldexpf(tot[0], 2 * (fe - sc[n]));
        if (g == n) sel = tot_all[n];
    }
    if (prio) __builtin_amdgcn_s_setprio(2);
    return sel;
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void quad_bench_kernel(const float *__restrict__ X, const float *__restrict__ Rimg, const float *__restrict__ mup,
                                                            const int32_t *__restrict__ rexp, int K, int reps, float *__restrict__ out) {
    constexpr int NB = 4, NG = 4;
    const int lane = threadIdx.x & 63, ci = lane & 15, g = lane >> 4;
    const size_t wbase = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    f32x4 x[NG][NB];
#pragma unroll
    for (int n = 0; n < NG; ++n)
#pragma unroll
        for (int t = 0; t < NB; ++t) x[n][t] = *reinterpret_cast<const f32x4 *>(X + (wbase + 16 * n + ci) * 64 + 16 * t + 4 * g);
    int xe = 0;
    if constexpr (F16) xe = f16_point_exponents<NG>(x);
    constexpr size_t MSZ = F16 ? (size_t)F16_WORDS : (size_t)2560;
    f32x4 rb0[NB], mu[NB];
    float tot_all[NG], acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        load_rb0<NB>(Rimg, mup, rb0, mu, lane, g);
        for (int k = 0; k < K; ++k) {
            const float *Rn = k + 1 < K ? Rimg + (size_t)(k + 1) * MSZ : nullptr;
            if constexpr (F16) acc += quad_f16_stream<NG>(Rimg + (size_t)k * MSZ, Rn, mup + (size_t)(k + 1) * 64, rexp[k], xe, rb0, mu, x, lane, g, tot_all, 0);
            else acc += quad_stream<NB, NG>(Rimg + (size_t)k * MSZ, Rn, mup + (size_t)(k + 1) * 64, rb0, mu, x, lane, g, true, tot_all, 0);
        }
    }
    out[wbase + lane] = acc;
}
}  // namespace dpmm

int main() {
    using namespace dpmm;
    const int K = 96, reps = 8, grid = 256 * 2 * 4;
    const size_t npts = (size_t)grid * 256;
    std::vector<float> hX(npts * 64), hR((size_t)K * 3072, 0.f), hmu((size_t)(K + 1) * 64, 0.1f);
    for (size_t i = 0; i < hX.size(); ++i) hX[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    // any finite words do for timing: f32 image [K][2560] floats / fp16 image [K][3072] words of small normal halves
    std::vector<uint32_t> hH((size_t)K * 3072);
    for (size_t i = 0; i < hH.size(); ++i) hH[i] = 0x3c003c00u ^ (uint32_t)((i * 40503u) & 0x03ff03ffu);
    for (size_t i = 0; i < hR.size(); ++i) hR[i] = (float)((i * 40503u) % 1000) / 1000.f;
    std::vector<int32_t> hE(K, 0);
    float *dX, *dR, *dmu, *dout; uint32_t *dH; int32_t *dE;
    (void)hipMalloc(&dX, hX.size() * 4); (void)hipMalloc(&dR, hR.size() * 4); (void)hipMalloc(&dH, hH.size() * 4); (void)hipMalloc(&dmu, hmu.size() * 4);
    (void)hipMalloc(&dE, K * 4); (void)hipMalloc(&dout, npts * 4);
    (void)hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dR, hR.data(), hR.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dH, hH.data(), hH.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dmu, hmu.data(), hmu.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dE, hE.data(), K * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(quad_bench_kernel<false>, dim3(grid), dim3(256), 0, 0, dX, dR, dmu, dE, K, reps, dout);
            else hipLaunchKernelGGL(quad_bench_kernel<true>, dim3(grid), dim3(256), 0, 0, dX, reinterpret_cast<const float *>(dH), dmu, dE, K, reps, dout);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // per SIMD: grid * 4 waves / 1024 SIMDs wave-tiles, each K * reps evaluations
        const double evals_per_simd = (double)grid * 4 / 1024.0 * K * reps;
        printf("%s: %.3f ms  -> %.3f us per evaluation per SIMD (%.0f cycles at 2.4 GHz); hipError %d\n", mode ? "fp16 split" : "Float32   ", best,
               1e3 * best / evals_per_simd, 2.4e6 * best / evals_per_simd, (int)hipGetLastError());
    }
    return 0;
}
