// Micro-benchmark + bit-comparison of the 64x64 pivot-tile factorisation (potrf_lds_t of ba_chol.h) in its variants
// (developer tool, not part of the library):  PV 0 = two v_mov_b32_dpp per broadcast (rounds 1-3), 1 = one v_mov_b64_dpp,
// 4 = panels of four columns, trailing update on the matrix cores (round 5);  OVL = inverse blocks formed inside the block-column loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bench_potrf tools/bench_potrf.hip      run: tools/bench_potrf
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../xrsfm_amd/csrc/ba_chol.h"
using namespace xba;

template <int PV, bool OVL>
__global__ __launch_bounds__(256) void k_bench(const double* __restrict__ Ain, double* __restrict__ Lout, double* __restrict__ Liout, int nb, int reps) {
    __shared__ double A[kNB][kLdT];
    __shared__ double Li[kNB][kLdT];
    __shared__ double Tb[3][16][17];
    const int t = threadIdx.x;
    const double* src = Ain + (size_t)blockIdx.x * kNB * kNB;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = t; e < kNB * kNB; e += 256) {
            const int r = e >> 6, c = e & 63;
            A[r][c] = (c <= r) ? src[e] : 0.0;
            Li[r][c] = (r == c && r >= 16 * nb) ? 1.0 : 0.0;
        }
        __syncthreads();
        potrf_lds_t<PV, OVL>(A, Li, Tb, nb);
        __syncthreads();
    }
    for (int e = t; e < kNB * kNB; e += 256) {
        const int r = e >> 6, c = e & 63;
        Lout[(size_t)blockIdx.x * kNB * kNB + e] = A[r][c];
        Liout[(size_t)blockIdx.x * kNB * kNB + e] = (c <= r) ? Li[r][c] : 0.0;
    }
}

int main() {
    const int NT = 64;           // tiles (workgroups) per launch
    std::vector<double> h((size_t)NT * kNB * kNB);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
    for (int b = 0; b < NT; ++b) {            // SPD: M M^T + 64 I
        std::vector<double> M(kNB * kNB);
        for (auto& v : M) v = rnd();
        for (int i = 0; i < kNB; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = (i == j) ? 8.0 + 4.0 * (b % 5) : 0.0;
                for (int k = 0; k < kNB; ++k) s += M[i * kNB + k] * M[j * kNB + k];
                h[(size_t)b * kNB * kNB + i * kNB + j] = s; h[(size_t)b * kNB * kNB + j * kNB + i] = s;
            }
    }
    double *dA, *dL, *dLi;
    hipMalloc(&dA, h.size() * 8); hipMalloc(&dL, h.size() * 8); hipMalloc(&dLi, h.size() * 8);
    hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> refL, refLi;
    auto run = [&](const char* name, auto kern, int nb) {
        const int reps = 200;
        hipLaunchKernelGGL(kern, dim3(NT), dim3(256), 0, 0, dA, dL, dLi, nb, 1);
        hipDeviceSynchronize();
        std::vector<double> L(h.size()), Li(h.size());
        hipMemcpy(L.data(), dL, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(Li.data(), dLi, h.size() * 8, hipMemcpyDeviceToHost);
        // accuracy: |L L^T - A| and |Li L - I| on the leading 16 nb rows
        double e_f = 0.0, e_i = 0.0;
        const int n = 16 * nb;
        for (int b = 0; b < NT; b += 17) {
            const double* Lb = &L[(size_t)b * kNB * kNB]; const double* Ib = &Li[(size_t)b * kNB * kNB]; const double* Ab = &h[(size_t)b * kNB * kNB];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) {
                    double s = 0.0, u = 0.0;
                    for (int k = 0; k <= j; ++k) s += Lb[i * kNB + k] * Lb[j * kNB + k];
                    for (int k = j; k <= i; ++k) u += Ib[i * kNB + k] * Lb[k * kNB + j];
                    e_f = fmax(e_f, fabs(s - Ab[i * kNB + j]) / Ab[i * kNB + i]);
                    e_i = fmax(e_i, fabs(u - (i == j ? 1.0 : 0.0)));
                }
        }
        bool same = true;
        if (refL.empty() || (int)refL.size() != (int)L.size() || nb != 4) { if (nb == 4 && refL.empty()) { refL = L; refLi = Li; } }
        if (nb == 4) same = memcmp(refL.data(), L.data(), L.size() * 8) == 0 && memcmp(refLi.data(), Li.data(), Li.size() * 8) == 0;
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, dA, dL, dLi, nb, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms1; hipEventElapsedTime(&ms1, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(NT), dim3(256), 0, 0, dA, dL, dLi, nb, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms64; hipEventElapsedTime(&ms64, e0, e1);
        printf("%-22s nb %d  %7.2f us per tile (1 workgroup)  %7.2f us (%d workgroups)  |LL^T-A| %.1e  |Li L - I| %.1e  %s\n", name, nb,
               ms1 * 1e3 / reps, ms64 * 1e3 / reps, NT, e_f, e_i, nb == 4 ? (same ? "bit-identical to PV0" : "DIFFERS from PV0") : "");
    };
    for (int nb : {4, 3}) {
        run("PV0", k_bench<0, false>, nb);
        run("PV0 + OVL", k_bench<0, true>, nb);
        run("PV1 (mov_b64_dpp)", k_bench<1, false>, nb);
        run("PV1 + OVL", k_bench<1, true>, nb);
        run("PV4 (MFMA panels)", k_bench<4, false>, nb);
        run("PV4 + OVL", k_bench<4, true>, nb);
    }
    return 0;
}
