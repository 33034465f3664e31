// Developer aid: sustained FP64 matrix-core rate (v_mfma_f64_16x16x4) with nothing else in the loop, for the occupancies the
// dense update kernels run at.   hipcc --offload-arch=gfx950 -O3 -o tools/bench_mfma tools/bench_mfma.hip ; gpurun -- tools/bench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[256 * 8192] = (double)(c1 - c0); out[256 * 8192 + 1] = (double)(w1 - w0); }
}
template <int NACC>
void run(int grid, int iters, double* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * NACC * 2048.0;
    double h[2]; hipMemcpy(h, d + 256 * 8192, sizeof(h), hipMemcpyDeviceToHost);
    printf("acc %2d grid %5d iters %6d: %8.3f ms  %6.1f TFLOP/s   shader clock %.0f MHz (s_memtime / 100 MHz wall clock), %.1f shader cycles per MFMA per wave\n",
           NACC, grid, iters, ms, flop / ms * 1e-9, h[0] / h[1] * 100.0, h[0] / ((double)iters * NACC));
}
int main() {
    double* d; hipMalloc(&d, sizeof(double) * (256 * 8192 + 2));
    for (int rep = 0; rep < 2; ++rep) {
        run<16>(256, 2000, d); run<16>(512, 2000, d); run<16>(1024, 2000, d); run<16>(2048, 1000, d);
        run<4>(512, 8000, d); run<4>(768, 8000, d); run<4>(1024, 8000, d);
        run<16>(512, 40000, d);      // ~0.2 s: sustained clocks
    }
    return 0;
}
