// Developer aid: do FP64 matrix instructions (v_mfma_f64_16x16x4, v_mfma_f64_4x4x4) and FP64 vector instructions of DIFFERENT waves
// of one SIMD overlap on gfx950, or do they share the issue slot / data path?  One workgroup of 1024 threads per CU = 4 waves per SIMD;
// the wave's role is picked from its index so that every SIMD holds the same mix:
//   mode 0: 4 vector waves          mode 1: 4 matrix waves (16x16x4)      mode 2: 2 vector + 2 matrix (16x16x4)
//   mode 3: 2 vector, 2 idle        mode 4: 2 matrix (16x16x4), 2 idle    mode 5: 4 matrix waves (4x4x4)      mode 6: 2 vector + 2 matrix (4x4x4)
// Every active wave runs `iters` rounds of 16 instructions of its kind (4 independent chains).  The answer is in the times:
// t(2) ~ max(t(3), t(4)) = the pipes overlap;  t(2) ~ t(3) + t(4) = they do not.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bench_pipes tools/bench_pipes.hip      run: tools/bench_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void k_pipes(double* out, int iters, int mode, double a0, double b0) {
    const int wave = threadIdx.x >> 6;          // waves 0-3 -> SIMD 0-3, 4-7 -> SIMD 0-3, ...
    const int slot = wave >> 2;                 // 0..3: the wave's slot on its SIMD
    int role;                                   // 0 idle, 1 vector, 2 matrix 16x16x4, 3 matrix 4x4x4
    switch (mode) {
        case 0: role = 1; break;
        case 1: role = 2; break;
        case 2: role = (slot & 1) ? 2 : 1; break;
        case 3: role = (slot & 1) ? 0 : 1; break;
        case 4: role = (slot & 1) ? 2 : 0; break;
        case 5: role = 3; break;
        default: role = (slot & 1) ? 3 : 1; break;
    }
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    double s = 0.0;
    const long long c0 = clock64();
    if (role == 1) {
        double x0 = a, x1 = b, x2 = a + b, x3 = a - b;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
            }
        }
        s = x0 + x1 + x2 + x3;
    } else if (role == 2) {
        v4d acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (role == 3) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        s = acc[0] + acc[1] + acc[2] + acc[3];
    }
    const long long c1 = clock64();
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[(size_t)1024 * 4096 + wave] = (double)(c1 - c0);
}

int main() {
    double* d; hipMalloc(&d, sizeof(double) * ((size_t)1024 * 4096 + 16));
    const int grid = 256, iters = 20000;
    const char* names[] = {"4 vector waves per SIMD", "4 matrix (16x16x4) waves", "2 vector + 2 matrix (16x16x4)", "2 vector, 2 idle", "2 matrix (16x16x4), 2 idle",
                           "4 matrix (4x4x4) waves", "2 vector + 2 matrix (4x4x4)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 7; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_pipes, dim3(grid), dim3(1024), 0, 0, d, iters, mode, 1.0, 1e-3);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_pipes, dim3(grid), dim3(1024), 0, 0, d, iters, mode, 1.0, 1e-3);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double h[16]; hipMemcpy(h, d + (size_t)1024 * 4096, sizeof(h), hipMemcpyDeviceToHost);
            double cmax = 0; for (int w = 0; w < 16; ++w) cmax = h[w] > cmax ? h[w] : cmax;
            printf("mode %d  %-34s %8.3f ms   slowest wave %.2f cycles per instruction of its own (16 x %d)   wave cycles: v %.0f  m %.0f\n", mode, names[mode], ms,
                   cmax / (16.0 * iters), iters, h[0], h[4]);
        }
    return 0;
}
