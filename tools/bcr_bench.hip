// Standalone check + timing of the block cyclic reduction solver (ba_bcr.h) on a random SPD chain / ring system.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/bcr_bench.hip -o tools/bcr_bench && tools/bcr_bench 1776 15 0
// arguments: free cameras, half bandwidth in cameras, ring (0 / 1), repetitions.  The check is the residual
// |S x + g| / |g| computed on the host from the block-sparse S (works at any size).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <random>
#include <utility>
#include <vector>
#include "../structure_from_motion_amd/csrc/ba_bcr.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1776, B = argc > 2 ? atoi(argv[2]) : 15, ring = argc > 3 ? atoi(argv[3]) : 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 5, shuffle = argc > 5 ? atoi(argv[5]) : 0;
    const int n = N * 6;
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    // upper blocks S(i, j), i <= j, keyed (i << 32 | j), row-major 6x6; diagonally dominant
    std::vector<long long> keys;
    std::vector<double> S;
    // shuffle != 0: the cameras are NUMBERED at random (the chain is still there, in another order)
    std::vector<int> name(N);
    for (int i = 0; i < N; ++i) name[i] = i;
    if (shuffle) std::shuffle(name.begin(), name.end(), std::mt19937(shuffle));
    std::vector<std::pair<long long, int>> klist;
    for (int i = 0; i < N; ++i)
        for (int j = i; j < N; ++j) {
            const int d = j - i;
            if (!(d <= B || (ring && N - d <= B))) continue;
            const int a = std::min(name[i], name[j]), b = std::max(name[i], name[j]);
            klist.push_back({((long long)a << 32) | (unsigned)b, 0});
        }
    std::sort(klist.begin(), klist.end());
    for (auto &kv : klist) {
        {
            const int i = (int)(kv.first >> 32), j = (int)(kv.first & 0xffffffff);
            keys.push_back(kv.first);
            double blk[36];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) blk[a * 6 + b] = U(rng);
            if (i == j)
                for (int a = 0; a < 6; ++a) {
                    for (int b = 0; b < a; ++b) blk[a * 6 + b] = blk[b * 6 + a];
                    blk[a * 6 + a] = 6.0 * (2 * B + 1) + 4.0 + blk[a * 6 + a];
                }
            for (int e = 0; e < 36; ++e) S.push_back(blk[e]);
        }
    }
    std::vector<double> g(n);
    for (auto &v : g) v = U(rng);
    const int n_upper = (int)keys.size();
    ba::BcrPlan P;
    std::vector<int64_t> k64(keys.begin(), keys.end());
    if (!ba::bcr_make_plan(N, k64.data(), k64.size(), P)) { printf("no plan for N=%d B=%d ring=%d\n", N, B, ring); return 3; }
    printf("plan: %sm=%d blocks of %d scalars (nt=%d), band %d, cyclic %d, levels %d, slots %d, elims %zu, ops %zu, LDS %zu B\n", P.reordered ? "(RCM order) " : "", P.m, P.nb, P.nt,
           P.band, (int)P.cyclic, P.levels(), P.n_slots, P.elims.size(), P.ops.size(), ba::bcr_factor_lds_bytes(P.nt));
    double *dS, *dg, *dx;
    ba::BcrDev d{};
    ba::BcrElim *delims; ba::BcrOp *dops; int32_t *dsrc, *dbs; int *dinfo;
    CK(hipMalloc(&dS, S.size() * 8)); CK(hipMalloc(&dg, n * 8)); CK(hipMalloc(&dx, n * 8));
    CK(hipMalloc(&delims, P.elims.size() * sizeof(ba::BcrElim))); CK(hipMalloc(&dops, (P.ops.size() + 1) * sizeof(ba::BcrOp)));
    CK(hipMalloc(&dsrc, P.src.size() * 4)); CK(hipMalloc(&dbs, P.pos_f.size() * 4));
    CK(hipMalloc(&d.DG, ba::bcr_matrix_doubles(P) * 8)); CK(hipMalloc(&d.bv, ba::bcr_vec_doubles(P) * 8));
    CK(hipMalloc(&d.linv, ba::bcr_linv_doubles(P) * 8)); CK(hipMalloc(&dinfo, 4));
    CK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, g.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(delims, P.elims.data(), P.elims.size() * sizeof(ba::BcrElim), hipMemcpyHostToDevice));
    if (!P.ops.empty()) CK(hipMemcpy(dops, P.ops.data(), P.ops.size() * sizeof(ba::BcrOp), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsrc, P.src.data(), P.src.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbs, P.pos_f.data(), P.pos_f.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dx, 0, n * 8));
    d.elims = delims; d.ops = dops; d.src = dsrc; d.pos_f = dbs; d.info = dinfo;
    CK(ba::bcr_prepare_device(P.nt));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tf = 0;
    for (int it = 0; it < reps + 1; ++it) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        CK(ba::bcr_solve_launch(0, P, d, dS, dg, dx));
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
    // residual r = S x + g
    std::vector<double> r(g);
    for (int k = 0; k < n_upper; ++k) {
        const int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffff);
        const double *blk = &S[(size_t)k * 36];
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) {
                if (i == j) {
                    r[i * 6 + a] += blk[a * 6 + b] * x[j * 6 + b];
                } else {
                    r[i * 6 + a] += blk[a * 6 + b] * x[j * 6 + b];
                    r[j * 6 + b] += blk[a * 6 + b] * x[i * 6 + a];
                }
            }
    }
    double rn = 0, gn = 0, xn = 0;
    for (int i = 0; i < n; ++i) { rn += r[i] * r[i]; gn += g[i] * g[i]; xn = std::max(xn, std::fabs(x[i])); }
    const double rel = std::sqrt(rn / gn);
    printf("N=%d B=%d ring=%d info=%d  |Sx+g|/|g| = %.3e (|x|max %.3e)  solve %.3f ms\n", N, B, ring, info, rel, xn, tf / reps);
#ifdef BA_BCR_TIMING
    {
        long long ts[16 * 40];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(ba::g_bcr_t), sizeof(ts)));
        printf("cycle stamps of workgroup 0 of the last factor launch, relative to wave 0 stamp 0\n"
               "(loaded, barrier | per step: A done, barrier, B done, barrier | C done, barrier, stored):\n");
        for (int w = 0; w < P.nt + 4; ++w) {
            printf("  wave %2d:", w);
            for (int k = 0; k < 5 + 4 * P.nt; ++k) printf(" %6lld", ts[w * 40 + k] ? (ts[w * 40 + k] - ts[0]) : -1);
            printf("\n");
        }
    }
#endif
    return (info == 0 && rel <= 1e-12) ? 0 : 2;
}
