// Micro-benchmark: LDS fp64 atomic-add and read rates on gfx950 (numbers quoted in DESIGN.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
    __shared__ double s[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) s[i] = 0.0;
    __syncthreads();
    double v = 1.0 + tid, acc = 0.0;
    int a;
    if (MODE == 0) a = tid;                 // conflict-free, consecutive doubles
    else if (MODE == 1) a = tid / 5;        // 5 lanes per address
    else if (MODE == 2) a = (tid % 36) + 36 * ((tid * 7 / 36) % 60);  // two 36-lane groups, random slots
    else if (MODE == 3) a = (tid / 6) % 10 * 33;  // ~6-way same address (camera tile)
    else if (MODE == 6) a = (tid % 5 + (tid / 5) / 8 % 3) * 33;            // k_build U pattern: point-major lanes, 7 camera tiles
    else if (MODE == 7) a = (tid % 5 + (tid / 5) / 8 % 3) * 33 + ((tid / 5) & 1) * (1024 + 8);  // same, two replicas by point parity (bank-shifted)
    else if (MODE == 8) a = (tid % 5 + (tid / 5) / 8 % 3) * 33 + ((tid / 5) & 3) * (1024 + 8);  // four replicas
    else if (MODE == 10) a = (tid % 5 + (tid / 5) / 8 % 3) * 33 + ((tid / 5) & 7) * (512 + 4);  // eight replicas
    else if (MODE == 9) a = (tid / 5) * 9;                                // k_build P pattern: 5 adjacent lanes per point
    else a = tid * 18;                      // MODE 4/5: reads at stride 144 B
    for (int it = 0; it < iters; ++it) {
        if (MODE <= 3 || MODE >= 6) {
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&s[MODE >= 6 ? a + u : ((a + u * 256) & 8191)], v);
        } else if (MODE == 4) {  // 18 doubles at stride 144 B per lane (b128 reads)
            const double *p = &s[(a + it * 18) & 4095];
#pragma unroll
            for (int u = 0; u < 18; ++u) acc += p[u];
        } else if (MODE == 5) {  // same, but lanes read consecutive doubles (conflict-free reference)
#pragma unroll
            for (int u = 0; u < 18; ++u) acc += s[(tid + u * 256 + it) & 8191];
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = acc + s[tid];
}

template <int MODE>
int run(const char *name, double *d, int iters, double lane_ops_per_iter) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 2;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 256 * iters * lane_ops_per_iter;
    printf("%-44s %8.3f ms  %8.2f lane-ops/ns chip  %6.2f lane-ops/clk/CU (2.1 GHz)\n", name, ms, ops / ms * 1e-6,
           ops / ms * 1e-6 / 256 / 2.1);
    return 0;
}

int main() {
    double *d; CHK(hipMalloc(&d, 512 * 256 * 8));
    run<0>("ds_add_f64 conflict-free", d, 2000, 8);
    run<1>("ds_add_f64 5 lanes/address", d, 2000, 8);
    run<2>("ds_add_f64 36-lane groups", d, 2000, 8);
    run<3>("ds_add_f64 ~6-way same address", d, 2000, 8);
    run<6>("ds_add_f64 U pattern (7 tiles, strided lanes)", d, 2000, 8);
    run<7>("ds_add_f64 U pattern, 2 replicas", d, 2000, 8);
    run<8>("ds_add_f64 U pattern, 4 replicas", d, 2000, 8);
    run<10>("ds_add_f64 U pattern, 8 replicas", d, 2000, 8);
    run<9>("ds_add_f64 P pattern (5 adjacent lanes)", d, 2000, 8);
    run<4>("ds_read 18 doubles/lane stride 144B", d, 2000, 18);
    run<5>("ds_read_b64 consecutive", d, 2000, 18);
    return 0;
}
