// developer probe: does the runtime accept CU-masked streams here?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = 42; }
int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d\n", n_cu); fflush(stdout);
    const int words = (n_cu + 31) / 32;
    std::vector<uint32_t> m(words, 0u);
    for (int c = 0; c < n_cu; ++c) if (c % 16 == 0) m[c / 32] |= 1u << (c % 32);
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, words, m.data());
    printf("create: %s\n", hipGetErrorString(e)); fflush(stdout);
    if (e != hipSuccess) return 0;
    int *d; hipMalloc(&d, 4);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, s, d);
    e = hipStreamSynchronize(s);
    int h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("sync: %s value %d\n", hipGetErrorString(e), h);
    hipStreamDestroy(s);
    return 0;
}
