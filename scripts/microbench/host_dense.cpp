// Dev microbenchmark: the two dense kernels of the master (dense.h) in isolation, one thread.
//   g++ -O3 -std=c++17 -fopenmp-simd -mavx2 -mfma -I dpmmsubclusters.jl_amd/host/csrc -o host_dense scripts/microbench/host_dense.cpp && ./host_dense 256
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include "dense.h"
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 200;
    std::vector<double> A((size_t)D * D), P((size_t)D * D), L((size_t)D * D), Y((size_t)D * D);
    srand(1);
    std::vector<double> G((size_t)D * (D + 8));
    for (auto &g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = i == j ? 1.0 : 0.0;
            for (int k = 0; k < D + 8; ++k) s += G[(size_t)i * (D + 8) + k] * G[(size_t)j * (D + 8) + k];
            A[(size_t)i * D + j] = s;
        }
    double best = 1e9, ld = 0;
    for (int r = 0; r < reps; ++r) {
        P = A;
        const double t0 = now();
        ld = dpmmh::chol_ltl(P.data(), D, L.data());
        best = std::min(best, now() - t0);
    }
    const double fl = (double)D * D * D / 3.0;
    printf("D=%d chol_ltl   %8.1f us  %6.1f GF/s  (logdet %.6f)\n", D, best * 1e6, fl / best / 1e9, ld);
    best = 1e9;
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) Y[(size_t)i * D + j] = j <= i ? G[(size_t)i * (D + 8) + j] + (i == j ? 3.0 : 0.0) : 0.0;
        const double t0 = now();
        dpmmh::solve_lower_left(Y.data(), L.data(), D);
        best = std::min(best, now() - t0);
    }
    double chk = 0; for (double y : Y) chk += y;
    printf("D=%d solve_left %8.1f us  %6.1f GF/s  (checksum %.6f)\n", D, best * 1e6, fl / best / 1e9, chk);
    return 0;
}
