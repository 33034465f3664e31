// Instruction latency / issue-rate probe for the in-register 16x16 factorisation (developer tool): one wave, s_memtime around
// 256 copies of an instruction pattern.  build: hipcc --offload-arch=gfx950 -O3 -o tools/bench_lat tools/bench_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
__global__ void k(double* io, unsigned long long* out) {
    double x = io[threadIdx.x], a = io[64 + threadIdx.x], b = io[128 + threadIdx.x];
    double y0 = x + 1, y1 = x + 2, y2 = x + 3, y3 = x + 4, y4 = x + 5, y5 = x + 6, y6 = x + 7, y7 = x + 8;
    unsigned long long t[16];
    int n = 0;
#define T() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory"); t[n++] = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" ::: "memory"); }
    T()   // 0
    R256(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));)                                     // dependent fma
    T()   // 1
    R16(asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                     : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(a), "v"(b));)          // 8 independent chains
    T()   // 2
    R256(asm volatile("v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x));)                     // dependent 64-bit dpp move
    T()   // 3
    R16(asm volatile("v_mov_b64_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                     : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(a));)                     // independent 64-bit dpp moves
    T()   // 4
    R256(asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fma_f64 %1, %0, %2, %1" : "+v"(x), "+v"(y0) : "v"(a));)   // move -> dependent fma -> move ...
    T()   // 5
    R256(asm volatile("v_rcp_f64 %0, %0" : "+v"(x));)                                                                  // dependent rcp
    T()   // 6
    R256(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));)                                                      // dependent mul
    T()   // 7
    R16(asm volatile("v_mov_b32_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                     : "+v"(((int*)&y0)[0]), "+v"(((int*)&y1)[0]), "+v"(((int*)&y2)[0]), "+v"(((int*)&y3)[0]), "+v"(((int*)&y4)[0]), "+v"(((int*)&y5)[0]),
                       "+v"(((int*)&y6)[0]), "+v"(((int*)&y7)[0]) : "v"(((int*)&a)[0]));)                                               // independent 32-bit dpp moves
    T()   // 8
    R256(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(a), "v"(b));)          // dependent fmac dpp
    T()   // 9
    R16(asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                     : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(a), "v"(b));)          // independent fmac dpp
    T()   // 10
    if (threadIdx.x == 0) for (int i = 0; i < n; ++i) out[i] = t[i];
    io[threadIdx.x] = x + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}
int main() {
    double* io; unsigned long long* out;
    hipMalloc(&io, 256 * 8); hipMalloc(&out, 16 * 8);
    double h[256]; for (int i = 0; i < 256; ++i) h[i] = 1.0 + 1e-9 * i;
    hipMemcpy(io, h, sizeof(h), hipMemcpyHostToDevice);
    unsigned long long t[16];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, io, out); hipDeviceSynchronize(); }
    hipMemcpy(t, out, sizeof(t), hipMemcpyDeviceToHost);
    const char* names[] = {"dependent v_fma_f64", "8 independent chains v_fma_f64", "dependent v_mov_b64_dpp", "independent v_mov_b64_dpp",
                           "v_mov_b64_dpp -> (s_nop 1) -> v_fma_f64 -> ... (pair)", "dependent v_rcp_f64", "dependent v_mul_f64", "independent v_mov_b32_dpp",
                           "dependent v_fmac_f64_dpp", "independent v_fmac_f64_dpp"};
    for (int i = 0; i < 10; ++i) printf("%-58s %7.2f cycles per instruction%s\n", names[i], (double)(t[i + 1] - t[i]) / 256.0, i == 4 ? " pair" : "");
    return 0;
}
