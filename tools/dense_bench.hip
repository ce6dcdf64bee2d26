// Standalone check + timing of the dense Cholesky kernel (ba_dense.h) on a random SPD system.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/dense_bench.hip -o tools/dense_bench && tools/dense_bench 276
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#ifndef BA_DENSE_H  // (-DBA_DENSE_H='"path"' -I structure_from_motion_amd/csrc: an experimental copy of the header)
#define BA_DENSE_H "../structure_from_motion_amd/csrc/ba_dense.h"
#endif
#include BA_DENSE_H

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int reps = 5;
    int rc = 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ba::k_dense_cholesky_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                           BA_DENSE_LDS_ROWS * BA_PS * (int)sizeof(double)));
    for (int ai = 1; ai < (argc > 1 ? argc : 2); ++ai) {
        const int n = argc > 1 ? atoi(argv[ai]) : 276;
        if (n < 6 || n > BA_DENSE_MAX || n % 6) { fprintf(stderr, "n must be a multiple of 6 in [6, %d]\n", BA_DENSE_MAX); return 1; }
        std::mt19937_64 rng(11 + n);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        std::vector<double> A((size_t)n * n), g(n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double v = U(rng);
                if (i == j) v += n + 2.0;
                A[(size_t)i * n + j] = A[(size_t)j * n + i] = v;
            }
        for (auto &v : g) v = U(rng);
        // CPU reference
        std::vector<double> L = A, xr(n);
        for (int c = 0; c < n; ++c) {
            double d = L[(size_t)c * n + c];
            for (int k = 0; k < c; ++k) d -= L[(size_t)c * n + k] * L[(size_t)c * n + k];
            d = std::sqrt(d);
            L[(size_t)c * n + c] = d;
            for (int r = c + 1; r < n; ++r) {
                double s = L[(size_t)r * n + c];
                for (int k = 0; k < c; ++k) s -= L[(size_t)r * n + k] * L[(size_t)c * n + k];
                L[(size_t)r * n + c] = s / d;
            }
        }
        for (int r = 0; r < n; ++r) {
            double s = -g[r];
            for (int k = 0; k < r; ++k) s -= L[(size_t)r * n + k] * xr[k];
            xr[r] = s / L[(size_t)r * n + r];
        }
        for (int r = n - 1; r >= 0; --r) {
            double s = xr[r];
            for (int k = r + 1; k < n; ++k) s -= L[(size_t)k * n + r] * xr[k];
            xr[r] = s / L[(size_t)r * n + r];
        }
        double *dA, *dg, *dx; int *dinfo;
        CK(hipMalloc(&dA, ((size_t)n * n + 8) * 8)); CK(hipMalloc(&dg, n * 8)); CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dinfo, 4));
        CK(hipMemcpy(dg, g.data(), n * 8, hipMemcpyHostToDevice));
        const int rows = n > BA_NB ? n - BA_NB : 0, lds_rows = rows < BA_DENSE_LDS_ROWS ? rows : BA_DENSE_LDS_ROWS;
        const size_t dyn = (size_t)lds_rows * BA_PS * sizeof(double);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float tf = 0;
        for (int it = 0; it < reps + 1; ++it) {
            CK(hipMemcpy(dA, A.data(), (size_t)n * n * 8, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(ba::k_dense_cholesky_solve, dim3(1), dim3(BA_DENSE_THREADS), dyn, 0, n, dA, dg, dx, dinfo, lds_rows);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float a;
            CK(hipEventElapsedTime(&a, e0, e1));
            if (it > 0) tf += a;
        }
        std::vector<double> x(n), Lg((size_t)n * n);
        int info = -1;
        CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(Lg.data(), dA, (size_t)n * n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
        double err = 0, nrm = 0, lerr = 0;
        for (int i = 0; i < n; ++i) { err = std::max(err, std::fabs(x[i] - xr[i])); nrm = std::max(nrm, std::fabs(xr[i])); }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) lerr = std::max(lerr, std::fabs(Lg[(size_t)i * n + j] - L[(size_t)i * n + j]));
        printf("n=%d info=%d  max|x-xref|=%.3e (|x|max %.3e)  max|L-Lref|=%.3e  solve %.1f us\n", n, info, err, nrm, lerr, tf / reps * 1e3);
        if (!(info == 0 && err <= 1e-10 * std::max(1.0, nrm) && lerr <= 1e-10 * n)) rc = 2;
        CK(hipFree(dA)); CK(hipFree(dg)); CK(hipFree(dx)); CK(hipFree(dinfo));
    }
    return rc;
}
