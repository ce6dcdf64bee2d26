// Dependent-issue latencies of a single wave on gfx950 (cycles from s_memtime, ns from wall_clock64).
//   hipcc -O3 --offload-arch=gfx950 tools/lat_bench.hip -o tools/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
#define REP 512
template <int MODE>
__global__ void k_lat(double *out, long long *t, double seed, int nwave_active) {
    __shared__ double sbuf[256];
    const int lane = threadIdx.x & 63;
    double v = seed + lane * 1e-9, w = seed * 0.5;
    sbuf[threadIdx.x & 255] = v;
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    if (MODE == 0) {  // dependent fp64 fma chain
#pragma unroll 16
        for (int i = 0; i < REP; ++i) v = fma(v, w, 1e-3);
    } else if (MODE == 1) {  // 4 independent fma chains
        double a = v, b = v + 1, c = v + 2, d = v + 3;
#pragma unroll 8
        for (int i = 0; i < REP; ++i) { a = fma(a, w, 1e-3); b = fma(b, w, 1e-3); c = fma(c, w, 1e-3); d = fma(d, w, 1e-3); }
        v = a + b + c + d;
    } else if (MODE == 2) {  // dependent v_rsq_f64
#pragma unroll 16
        for (int i = 0; i < REP; ++i) v = __builtin_amdgcn_rsq(v) + 1.5;
    } else if (MODE == 3) {  // fma -> readlane -> fma (SGPR operand)
#pragma unroll 16
        for (int i = 0; i < REP; ++i) {
            const double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 8), __builtin_amdgcn_readlane(__double2loint(v), 8));
            v = fma(v, w, s);
        }
    } else if (MODE == 4) {  // LDS write -> read other lane -> fma
        for (int i = 0; i < REP; ++i) {
            sbuf[lane] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            v = fma(sbuf[(lane + 1) & 63], w, 1e-3);
        }
    } else if (MODE == 5) {  // dpp mov (both halves) + add
#pragma unroll 16
        for (int i = 0; i < REP; ++i) {
            int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, true);
            int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, true);
            v = v * 0.5 + __hiloint2double(hi, lo);
        }
    } else if (MODE == 6) {  // __syncthreads only
        for (int i = 0; i < REP; ++i) __syncthreads();
    } else if (MODE == 7) {  // ds_bpermute shuffle + add
#pragma unroll 16
        for (int i = 0; i < REP; ++i) v = v * 0.5 + __shfl_xor(v, 1, 64);
    } else if (MODE == 8) {  // dependent fp32 fma chain
        float f = (float)v, g = (float)w;
#pragma unroll 16
        for (int i = 0; i < REP; ++i) f = fmaf(f, g, 1e-3f);
        v = f;
    } else if (MODE == 9) {  // dependent fp64 mul chain
#pragma unroll 16
        for (int i = 0; i < REP; ++i) v = v * w;
    } else if (MODE == 10) {  // global load -> dependent address (L2 hit pointer chase)
        const long long *p = (const long long *)t + 64;
        long long idx = lane & 1;
        for (int i = 0; i < REP; ++i) idx = p[idx];
        v += (double)idx;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if ((threadIdx.x & 63) == 0) { t[2 * (threadIdx.x >> 6)] = c1 - c0; t[2 * (threadIdx.x >> 6) + 1] = w1 - w0; }
}
template <int MODE>
int run(const char *name, int threads) {
    double *out; long long *t, ht[32];
    CK(hipMalloc(&out, 8 * 1024)); CK(hipMalloc(&t, 8 * 256)); CK(hipMemset(t, 0, 8 * 256));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_lat<MODE>, dim3(1), dim3(threads), 0, 0, out, t, 1.0000001, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost));
    for (int w = 1; w < threads / 64; ++w) { if (ht[2 * w] > ht[0]) ht[0] = ht[2 * w]; if (ht[2 * w + 1] > ht[1]) ht[1] = ht[2 * w + 1]; }
    printf("%-44s threads %4d: %7.1f clk/op  %7.2f ns/op\n", name, threads, (double)ht[0] / REP, (double)ht[1] * 10.0 / REP);
    CK(hipFree(out)); CK(hipFree(t));
    return 0;
}
int main() {
    run<0>("fp64 fma dependent chain", 64);
    run<9>("fp64 mul dependent chain", 64);
    run<1>("fp64 fma 4 independent chains (per 4 ops)", 64);
    run<8>("fp32 fma dependent chain", 64);
    run<2>("v_rsq_f64 + add dependent", 64);
    run<3>("2x readlane + fma dependent", 64);
    run<4>("LDS write, wait, read, fma", 64);
    run<5>("2x mov_dpp + mul + add dependent", 64);
    run<7>("shfl_xor (bpermute) + mul + add", 64);
    run<6>("__syncthreads", 64);
    run<6>("__syncthreads", 256);
    run<6>("__syncthreads", 640);
    run<6>("__syncthreads", 1024);
    run<10>("global load pointer chase (L2/TCP hit)", 64);
    run<0>("fp64 fma dependent chain, 4 waves (1/SIMD)", 256);
    run<0>("fp64 fma dependent chain, 8 waves (2/SIMD)", 512);
    run<0>("fp64 fma dependent chain, 16 waves (4/SIMD)", 1024);
    run<1>("fp64 fma 4 chains (per 4 ops), 8 waves", 512);
    run<1>("fp64 fma 4 chains (per 4 ops), 16 waves", 1024);
    run<8>("fp32 fma dependent chain, 16 waves", 1024);
    return 0;
}
