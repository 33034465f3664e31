// Developer aid: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by one-hot probing.
// wave w = (la, lb): A = 1 in lane la, B = 1 in lane lb; D (one double per lane) is 1 where D_b[i][j] = sum_k A_b[i][k] B_b[k][j] picks the pair up.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma4 tools/probe_mfma4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(double* out) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const int la = w >> 6, lb = w & 63;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[(size_t)w * 64 + lane] = d;
}
int main() {
    double* d; hipMalloc(&d, sizeof(double) * 4096 * 64);
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, d);
    std::vector<double> h(4096 * 64);
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // for every A lane: which B lanes pair with it (same block, same k) and where the product lands
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf("  B%d->D%d", lb, l);
        printf("\n");
    }
    return 0;
}
