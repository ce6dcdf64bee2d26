// Standalone check + timing of the block-banded Cholesky kernels (ba_banded.h) on a random SPD band system.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/band_bench.hip -o tools/band_bench && tools/band_bench 1776 15
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#ifndef BA_BAND_TIMING
#define BA_BAND_TIMING 100
#endif
#include "../structure_from_motion_amd/csrc/ba_banded.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1776, B = argc > 2 ? atoi(argv[2]) : 15, reps = argc > 3 ? atoi(argv[3]) : 5;
    const int P = B + 1, n = N * 6;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    // lower band, scalar storage A[i][i - jj], jj in [i - bw, i], bw = 6B + 5 covers all blocks
    std::vector<double> Lb((size_t)N * P * 36, 0.0), g(n);
    for (int j = 0; j < N; ++j)
        for (int d = 0; d <= B && j + d < N; ++d)
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    if (d == 0 && b > a) continue;
                    double v = U(rng);
                    if (d == 0 && a == b) v = 6.0 * (2 * B + 1) + 4.0 + v;
                    Lb[((size_t)j * P + d) * 36 + a * 6 + b] = v;
                    if (d == 0) Lb[((size_t)j * P) * 36 + b * 6 + a] = v;
                }
    for (auto &v : g) v = U(rng);
    // CPU reference: dense-band scalar Cholesky
    auto A = [&](int r, int c) -> double & {  // r >= c
        const int br = r / 6, bc = c / 6;
        return Lb[((size_t)bc * P + (br - bc)) * 36 + (r % 6) * 6 + (c % 6)];
    };
    std::vector<double> Lc = Lb;
    auto Lr = [&](int r, int c) -> double & {
        const int br = r / 6, bc = c / 6;
        return Lc[((size_t)bc * P + (br - bc)) * 36 + (r % 6) * 6 + (c % 6)];
    };
    (void)A;
    const int bw = 6 * B + 5;
    for (int c = 0; c < n; ++c) {
        double d = Lr(c, c);
        for (int k = std::max(0, c - bw); k < c; ++k) if (c / 6 - k / 6 <= B) d -= Lr(c, k) * Lr(c, k);
        d = std::sqrt(d);
        Lr(c, c) = d;
        for (int r = c + 1; r < n && r / 6 - c / 6 <= B; ++r) {
            double s = Lr(r, c);
            for (int k = std::max(0, r - bw); k < c; ++k) if (r / 6 - k / 6 <= B && c / 6 - k / 6 <= B) s -= Lr(r, k) * Lr(c, k);
            Lr(r, c) = s / d;
        }
    }
    std::vector<double> xr(n);
    for (int r = 0; r < n; ++r) {
        double s = -g[r];
        for (int k = std::max(0, r - bw); k < r; ++k) if (r / 6 - k / 6 <= B) s -= Lr(r, k) * xr[k];
        xr[r] = s / Lr(r, r);
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = xr[r];
        for (int k = r + 1; k < n && k / 6 - r / 6 <= B; ++k) s -= Lr(k, r) * xr[k];
        xr[r] = s / Lr(r, r);
    }
    // the solver's input format: upper blocks S(i, j), i <= j, keyed (i << 32 | j), row-major 6x6
    std::vector<long long> keys;
    std::vector<double> S;
    for (int i = 0; i < N; ++i)
        for (int j = i; j <= i + B && j < N; ++j) {
            keys.push_back(((long long)i << 32) | (unsigned)j);
            const double *low = &Lb[((size_t)i * P + (j - i)) * 36];  // lower block (j, i)
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) S.push_back(low[b * 6 + a]);
        }
    const int n_upper = (int)keys.size();
    long long *dkeys; double *dS, *dg, *dx; int *dinfo;
    ba::BandWork w{};
    CK(hipMalloc(&dkeys, keys.size() * 8)); CK(hipMalloc(&dS, S.size() * 8)); CK(hipMalloc(&dg, n * 8)); CK(hipMalloc(&dx, n * 8));
    CK(hipMalloc(&w.l, ba::band_l_doubles(N, B) * 8)); CK(hipMalloc(&w.y, ba::band_vec_doubles(N, B) * 8));
    CK(hipMalloc(&w.g, ba::band_vec_doubles(N, B) * 8)); CK(hipMalloc(&w.mid, ba::band_mid_doubles(B) * 8));
    CK(hipMalloc(&dinfo, 4));
    w.info = dinfo;
    CK(hipMemcpy(dkeys, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, g.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tf = 0;
    for (int it = 0; it < reps + 1; ++it) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        CK(ba::band_solve_launch(0, N, B, n_upper, dkeys, dS, dg, dx, w));
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float a;
        CK(hipEventElapsedTime(&a, e0, e1));
        if (it > 0) tf += a;
    }
    std::vector<double> x(n);
    int info = -1;
    CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (int i = 0; i < n; ++i) { err = std::max(err, std::fabs(x[i] - xr[i])); nrm = std::max(nrm, std::fabs(xr[i])); }
    printf("N=%d B=%d threads=%d two-sided=%d info=%d  max|x-xref|=%.3e (|x|max %.3e)  solve %.3f ms (%.3f us/col)\n",
           N, B, ba::band_threads(B), (int)ba::band_two_sided(N, B), info, err, nrm, tf / reps, tf / reps / N * 1e3);
    long long ts[16 * 8];
    CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(ba::g_band_t), sizeof(ts)));
    printf("cycle stamps of column %d relative to wave 0 stamp 0 (A start, after chol6, A end, after barrier, B end, after barrier):\n", BA_BAND_TIMING);
    for (int w = 0; w < ba::band_threads(B) / 64 && w < 16; ++w) {
        printf("  wave %2d:", w);
        for (int k = 0; k < 6; ++k) printf(" %6lld", ts[w * 8 + k] ? ts[w * 8 + k] - ts[0] : -1);
        printf("\n");
    }
    return (info == 0 && err <= 1e-10 * std::max(1.0, nrm)) ? 0 : 2;
}
