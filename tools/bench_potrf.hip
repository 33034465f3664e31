// micro-benchmark of the tile kernels (developer tool, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../xrsfm_amd/csrc/ba_chol.h"
using namespace xba;
__global__ void k_empty() {}
int main() {
    const int T = 8, n_pad = T * kNB;
    std::vector<double> h((size_t)n_pad * n_pad, 0.0);
    for (int i = 0; i < n_pad; ++i) for (int j = 0; j <= i; ++j) { double v = (i == j) ? 100.0 + i % 7 : 1.0 / (1.0 + abs(i - j)); h[(size_t)i * n_pad + j] = v; }
    CholDev c{}; c.n = n_pad; c.n_pad = n_pad; c.T = T;
    hipMalloc(&c.S, h.size() * 8); hipMalloc(&c.Linv, (size_t)T * kNB * kNB * 8);
    hipMalloc(&c.y, n_pad * 8); hipMalloc(&c.rhs, n_pad * 8); hipMalloc(&c.x, n_pad * 8);
    int* rows; hipMalloc(&rows, 64 * 4); std::vector<int> hr = {1, 2, 3, 1, 1, 2, 1, 2, 2, 3, 1, 3, 2, 3, 3}; hipMemcpy(rows, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    int* klist; hipMalloc(&klist, 64); std::vector<int> hk = {0, 1, 2, 3}; hipMemcpy(klist, hk.data(), 16, hipMemcpyHostToDevice);
    int* trows; hipMalloc(&trows, 64); std::vector<int> ht(T, 64); hipMemcpy(trows, ht.data(), T * 4, hipMemcpyHostToDevice); c.tile_rows = trows;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto f, int reps) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-12s %8.2f us/launch\n", name, ms * 1e3 / reps);
    };
    const size_t shm = 2 * (size_t)kNB * kLdT * sizeof(double);
    hipFuncSetAttribute((const void*)k_trsm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipFuncSetAttribute((const void*)k_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    timeit("empty", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, 0); }, 1000);
    timeit("potrf", [&] { hipMemcpyAsync(c.S, h.data(), 8, hipMemcpyHostToDevice, 0); hipLaunchKernelGGL(k_potrf, dim3(1), dim3(256), 0, 0, c, klist, (const int*)nullptr, (const int*)nullptr); }, 200);
    hipMemcpy(c.S, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    timeit("potrf_only", [&] { hipLaunchKernelGGL(k_potrf, dim3(1), dim3(256), 0, 0, c, klist + 1, (const int*)nullptr, (const int*)nullptr); }, 1000);
    timeit("trsm x3", [&] { hipLaunchKernelGGL(k_trsm, dim3(3), dim3(256), shm, 0, c, 0, rows); }, 1000);
    timeit("update x6", [&] { hipLaunchKernelGGL(k_update, dim3(6), dim3(256), shm, 0, c, 0, rows + 3); }, 1000);
    timeit("fwd", [&] { hipLaunchKernelGGL(k_fwd, dim3(4), dim3(256), 0, 0, c, 0, rows); }, 1000);
    timeit("bwd", [&] { hipLaunchKernelGGL(k_bwd, dim3(3), dim3(256), 0, 0, c, 3, rows); }, 1000);
    return 0;
}
