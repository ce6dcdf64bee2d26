// Micro-benchmark: ds_read_b128 gathers of 144-byte rows (the T rows of the build kernel's Schur phase) for different
// assignments of rows to lanes -- which lane groups have to hit distinct bank classes for a conflict-free read?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k(const int *rows, double *out, int iters) {
    __shared__ __attribute__((aligned(16))) double s[256 * 18];
    const int tid = threadIdx.x;
    for (int i = tid; i < 256 * 18; i += 256) s[i] = i;
    __syncthreads();
    double acc = 0.0;
    int r = rows[tid];
    for (int it = 0; it < iters; ++it) {
        const double2 *p = reinterpret_cast<const double2 *>(&s[r * 18]);
#pragma unroll
        for (int u = 0; u < 9; ++u) { const double2 v = p[u]; acc += v.x + v.y; }
        r = (r + 16) & 255;  // same class pattern every iteration
    }
    out[blockIdx.x * 256 + tid] = acc;
}

int run(const char *name, const int *hrows, int *drows, double *d) {
    CHK(hipMemcpy(drows, hrows, 256 * sizeof(int), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 3, iters = 2000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, drows, d, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, drows, d, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * 256 * iters * 144;
    printf("%-52s %8.3f ms  %7.1f B/clk/CU (2.4 GHz)\n", name, ms, bytes / (ms * 1e-3) / 256 / 2.4e9);
    return 0;
}

int main() {
    double *d; int *dr;
    CHK(hipMalloc(&d, 768 * 256 * 8)); CHK(hipMalloc(&dr, 256 * 4));
    int rows[256];
    for (int i = 0; i < 256; ++i) rows[i] = i;
    run("rows = lane (consecutive)", rows, dr, d);
    srand(1);
    for (int i = 0; i < 256; ++i) rows[i] = rand() & 255;
    run("rows random", rows, dr, d);
    for (int i = 0; i < 256; ++i) rows[i] = (i & 15) * 16 + (rand() & 15) * 0 ;  // every lane of a 16-block: same class? (row mod 16 == 0)
    for (int i = 0; i < 256; ++i) rows[i] = ((rand() & 15) * 16) & 255;
    run("all lanes one class (row mod 16 == 0)", rows, dr, d);
    for (int b = 0; b < 256; b += 16) {  // each aligned block of 16 lanes: a random permutation of the 16 classes
        int perm[16]; for (int i = 0; i < 16; ++i) perm[i] = i;
        for (int i = 15; i > 0; --i) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        for (int i = 0; i < 16; ++i) rows[b + i] = perm[i] + 16 * (rand() & 15);
    }
    run("distinct classes inside aligned 16-lane blocks", rows, dr, d);
    for (int b = 0; b < 256; b += 8) {   // aligned blocks of 8 lanes distinct, no constraint across
        int perm[16]; for (int i = 0; i < 16; ++i) perm[i] = i;
        for (int i = 15; i > 0; --i) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        for (int i = 0; i < 8; ++i) rows[b + i] = perm[i] + 16 * (rand() & 15);
    }
    run("distinct classes inside aligned 8-lane blocks", rows, dr, d);
    for (int b = 0; b < 256; b += 32) {  // 32-lane blocks: every class exactly twice
        int perm[32]; for (int i = 0; i < 32; ++i) perm[i] = i & 15;
        for (int i = 31; i > 0; --i) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        for (int i = 0; i < 32; ++i) rows[b + i] = perm[i] + 16 * (rand() & 15);
    }
    run("every class twice inside aligned 32-lane blocks", rows, dr, d);
    // class of lane l given by a formula; the row of a lane: its class's residue (9 r = c mod 16) + 16 * (a distinct multiple)
    struct { const char *name; int (*cls)(int); } pats[] = {
        {"class = lane & 15", [](int l) { return l & 15; }},
        {"class = (lane >> 1) & 15", [](int l) { return (l >> 1) & 15; }},
        {"class = (lane >> 2) & 15", [](int l) { return (l >> 2) & 15; }},
        {"class = (lane >> 3) & 15 (8 neighbours share)", [](int l) { return (l >> 3) & 15; }},
        {"class = (lane & 3) * 4 + (lane >> 4) (l, l+4, l+8, l+12 share)", [](int l) { return (l & 3) * 4 + ((l >> 4) & 3); }},
        {"class = (lane & 7) * 2 + (lane >> 5) (l, l+8, l+16, l+24 share)", [](int l) { return (l & 7) * 2 + ((l >> 5) & 1); }},
        {"class = lane & 7 (l, l+8, ... share; 8 classes)", [](int l) { return l & 7; }},
        {"class = (9 * lane) & 15 (= consecutive rows)", [](int l) { return (9 * l) & 15; }},
        {"class = (5 * lane) & 15", [](int l) { return (5 * l) & 15; }},
        {"class = (lane & 15) ^ (lane >> 4)", [](int l) { return (l & 15) ^ ((l >> 4) & 3); }},
    };
    for (auto &pt : pats) {
        int used[16] = {0};
        for (int i = 0; i < 256; ++i) {
            const int c = pt.cls(i & 63), r = (9 * c) & 15;
            rows[i] = (r + 16 * (used[c]++ & 15)) & 255;
        }
        run(pt.name, rows, dr, d);
    }
    for (int i = 0; i < 256; ++i) rows[i] = (i >> 1) * 2 % 256;  // lane pairs share a row (broadcast)
    run("lane pairs share a row, consecutive", rows, dr, d);
    return 0;
}
