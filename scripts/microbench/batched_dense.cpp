// Dev microbenchmark: could the master's dense algebra (3K factorisations + triangular solves per sweep) run on the GPU through
// rocSOLVER / rocBLAS?  Times dpotrf_strided_batched and dtrsm_strided_batched for `batch` matrices of size D.
//   hipcc -O2 -o batched_dense.bin batched_dense.cpp -L/opt/rocm/lib -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 256, batch = argc > 2 ? atoi(argv[2]) : 96;
    const size_t DD = (size_t)D * D;
    std::vector<double> h(DD * batch), hb(DD * batch);
    srand(1);
    for (int b = 0; b < batch; ++b) {
        std::vector<double> G((size_t)D * (D + 8));
        for (auto &g : G) g = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double s = i == j ? 1.0 : 0.0;
                for (int k = 0; k < D + 8; ++k) s += G[(size_t)i * (D + 8) + k] * G[(size_t)j * (D + 8) + k];
                h[b * DD + (size_t)i * D + j] = s;
                hb[b * DD + (size_t)i * D + j] = j <= i ? G[(size_t)i * (D + 8) + j] : 0.0;
            }
    }
    double *dA, *dA0, *dB, *dB0; int *info;
    hipMalloc(&dA, DD * batch * 8); hipMalloc(&dA0, DD * batch * 8); hipMalloc(&dB, DD * batch * 8); hipMalloc(&dB0, DD * batch * 8); hipMalloc(&info, 4 * batch);
    hipMemcpy(dA0, h.data(), DD * batch * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB0, hb.data(), DD * batch * 8, hipMemcpyHostToDevice);
    rocblas_handle hd; rocblas_create_handle(&hd);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best_p = 1e9, best_t = 1e9;
    const double one = 1.0;
    for (int rep = 0; rep < 6; ++rep) {
        hipMemcpy(dA, dA0, DD * batch * 8, hipMemcpyDeviceToDevice);
        hipMemcpy(dB, dB0, DD * batch * 8, hipMemcpyDeviceToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        rocsolver_dpotrf_strided_batched(hd, rocblas_fill_lower, D, dA, D, DD, info, batch);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep) best_p = ms < best_p ? ms : best_p;
        hipEventRecord(e0);
        rocblas_dtrsm_strided_batched(hd, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, D, D, &one, dA, D, DD, dB, D, DD, batch);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (rep) best_t = ms < best_t ? ms : best_t;
    }
    std::vector<int> hi(batch); hipMemcpy(hi.data(), info, 4 * batch, hipMemcpyDeviceToHost);
    int bad = 0; for (int v : hi) bad += v != 0;
    printf("D=%d batch=%d: potrf_strided_batched %.3f ms, trsm_strided_batched (D right-hand sides) %.3f ms, failed factorisations %d\n", D, batch, best_p, best_t, bad);
    return 0;
}
