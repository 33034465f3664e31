// libxrsfm_ba.so — MI355X-native bundle adjustment behind include/xrsfm_ba.h.
//
// Host controller: the Levenberg-Marquardt trust-region loop that the reference
// delegates to ceres::Solve (/root/reference/src/optimization/ba_solver.cc:591,
// 636,672) with SPARSE_SCHUR + LEVENBERG_MARQUARDT (ba_solver.cc:74-75); the
// accept/reject/termination rules restate Ceres' TrustRegionMinimizer
// (SURVEY.md Appendix A.5/A.6).  All arithmetic on the state runs in the HIP
// kernels of ba_kernels.h; the host only sees a handful of scalars per step.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/xrsfm_ba.h"
#include "ba_kernels.h"
#include "ba_pack.h"

using namespace xba;

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "[xrsfm_ba] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_),   \
                    __FILE__, __LINE__, #expr);                                                \
            return XRSFM_BA_ENODEV;                                                            \
        }                                                                                      \
    } while (0)

// ---------------------------------------------------------------- RCCL (loaded lazily, only for n_ranks > 1)
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void*, int) = nullptr;   // ncclUniqueId passed by value (128 B struct)
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
};
struct UniqueId { char internal[128]; };
typedef int (*fn_init_rank)(void**, int, UniqueId, int);
typedef int (*fn_unique_id)(UniqueId*);

Rccl g_rccl;
bool load_rccl() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return false;
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, const void*, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    return g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy;
}
constexpr int kNcclFloat64 = 8;   // ncclDouble
constexpr int kNcclSum = 0, kNcclMax = 2;
}  // namespace

// ---------------------------------------------------------------- context
struct xrsfm_ba_context {
    int device = 0;
    hipStream_t stream = nullptr;
    Packed pk;
    Dev d{};
    std::vector<void*> allocs;
    // pristine copies for reset
    CamRec* cam0 = nullptr; double* P0 = nullptr;
    int n_points_caller = 0;
    double* h_scal = nullptr;       // pinned
    PcgStatus* h_st = nullptr;      // pinned
    // comm
    void* comm = nullptr; int n_ranks = 1, rank = 0;
    // profiling
    std::vector<hipEvent_t> ev;
    bool linearized = false; bool scaled = false;
    double dbg_radius = 0;
};

namespace {

template <typename T>
int dev_alloc(xrsfm_ba_context* c, T** p, size_t n) {
    void* q = nullptr;
    if (n == 0) n = 1;
    if (hipMalloc(&q, n * sizeof(T)) != hipSuccess) return XRSFM_BA_ENOMEM;
    c->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}
template <typename T>
int dev_upload(xrsfm_ba_context* c, T** p, const std::vector<T>& v) {
    int e = dev_alloc(c, p, v.size());
    if (e) return e;
    if (!v.empty() && hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return XRSFM_BA_ENODEV;
    return 0;
}

inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }

int allreduce(xrsfm_ba_context* c, double* buf, size_t n, int op) {
    if (c->n_ranks <= 1) return 0;
    const int e = g_rccl.AllReduce(buf, buf, n, kNcclFloat64, op, c->comm, c->stream);
    return e == 0 ? 0 : XRSFM_BA_ECOMM;
}

// Linearise at the current state (scale arrays as they are).  Leaves S_COST,
// S_XNORM2_PTS in d.scal and camlin / Hpp / gp filled.
int linearize(xrsfm_ba_context* c, double huber_a) {
    Dev& d = c->d;
    const int nb = cdiv(d.n_items, kWavesPerBlock);
    if (d.n_items > 0) hipLaunchKernelGGL(k_linearize, dim3(nb), dim3(kBlock), 0, c->stream, d, huber_a);
    if (d.n_cams > 0) hipLaunchKernelGGL(k_cam_segsum<12>, dim3(d.n_cams), dim3(kBlock), 0, c->stream, d.scat, d.cam_ptr,
                       d.camlin, (const PcgStatus*)nullptr);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(kPcgThreads), 0, c->stream, d.part, d.n_items, d.scal + S_COST);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(kPcgThreads), 0, c->stream, d.part + d.n_items, d.n_items,
                       d.scal + S_XNORM2_PTS);
    int e = allreduce(c, d.camlin, (size_t)d.n_cams * 12, kNcclSum);
    if (e) return e;
    e = allreduce(c, d.scal + S_COST, 2, kNcclSum);   // S_COST, S_XNORM2_PTS adjacent
    return e;
}

int fetch_scalars(xrsfm_ba_context* c) {
    HIPCHK(hipMemcpyAsync(c->h_scal, c->d.scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int gradient_max(xrsfm_ba_context* c, double* out) {
    Dev& d = c->d;
    HIPCHK(hipMemsetAsync(d.scal + S_GRADMAX_PTS, 0, sizeof(double), c->stream));
    if (d.n_pts > 0) hipLaunchKernelGGL(k_gradmax_pts, dim3(cdiv(d.n_pts, kBlock)), dim3(kBlock), 0, c->stream, d, d.scal + S_GRADMAX_PTS);
    if (d.n_cams > 0) hipLaunchKernelGGL(k_gradmax_cams, dim3(cdiv(d.n_cams, kBlock)), dim3(kBlock), 0, c->stream, d);
    hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(kPcgThreads), 0, c->stream, d.campart, d.n_cams, d.scal + S_GRADMAX_CAMS);
    int e = allreduce(c, d.scal + S_GRADMAX_PTS, 1, kNcclMax);
    if (e) return e;
    e = fetch_scalars(c);
    if (e) return e;
    *out = std::fmax(c->h_scal[S_GRADMAX_PTS], c->h_scal[S_GRADMAX_CAMS]);
    return 0;
}

// Build everything that depends on the radius: D^2, Hpp^-1, block-Jacobi
// preconditioner and the reduced right-hand side.
int prepare_step(xrsfm_ba_context* c, double radius) {
    Dev& d = c->d;
    const double dmin = 1e-6, dmax = 1e32;
    if (d.n_pts > 0) hipLaunchKernelGGL(k_point_prep, dim3(cdiv(d.n_pts, kBlock)), dim3(kBlock), 0, c->stream, d, radius, dmin, dmax);
    if (d.n_cams > 0) hipLaunchKernelGGL(k_cam_prep, dim3(cdiv((long long)d.n_cams * 6, kBlock)), dim3(kBlock), 0, c->stream, d, radius, dmin, dmax);
    if (d.n_slots > 0) hipLaunchKernelGGL(k_schur_prep, dim3(cdiv(d.n_slots, kBlock)), dim3(kBlock), 0, c->stream, d);
    if (d.n_cams > 0) hipLaunchKernelGGL(k_cam_segsum<28>, dim3(d.n_cams), dim3(kBlock), 0, c->stream, d.scat, d.cam_ptr, d.camS,
                       (const PcgStatus*)nullptr);
    int e = allreduce(c, d.camS, (size_t)d.n_cams * 28, kNcclSum);
    if (e) return e;
    if (d.n_cams > 0) hipLaunchKernelGGL(k_cam_factor, dim3(cdiv(d.n_cams, 64)), dim3(64), 0, c->stream, d);
    return 0;
}

// y = sum_obs F^T (F p - E Hinv E^T F p)  (+ D_c^2 p is added by the consumer)
int schur_product(xrsfm_ba_context* c, const double* p_dev, double* out_dev, bool timed, hipEvent_t e0, hipEvent_t e1) {
    Dev& d = c->d;
    if (timed) hipEventRecord(e0, c->stream);
    if (d.n_items > 0) hipLaunchKernelGGL(k_schur_matvec, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), 0, c->stream, d, p_dev);
    if (timed) hipEventRecord(e1, c->stream);
    if (d.n_cams > 0) hipLaunchKernelGGL(k_cam_segsum<6>, dim3(d.n_cams), dim3(kBlock), 0, c->stream, d.scat, d.cam_ptr, out_dev,
                       (const PcgStatus*)d.st);
    return allreduce(c, out_dev, (size_t)d.n_cams * 6, kNcclSum);
}

int pcg_solve(xrsfm_ba_context* c, const xrsfm_ba_options& opt, xrsfm_ba_summary* sum) {
    Dev& d = c->d;
    hipLaunchKernelGGL(k_pcg_init, dim3(1), dim3(kPcgThreads), 0, c->stream, d);
    const int chunk = 8;
    int launched = 0, it_prev = 0;
    while (true) {
        HIPCHK(hipMemcpyAsync(c->h_st, d.st, sizeof(PcgStatus), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (opt.profile && launched > 0) {
            // launches after `done` are no-ops and are not counted
            const int eff = c->h_st->it - it_prev;
            for (int i = 0; i < eff && i < chunk; ++i) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]) == hipSuccess) { sum->dom_kernel_ms += ms; sum->dom_kernel_launches++; }
            }
        }
        it_prev = c->h_st->it;
        if (c->h_st->done || launched >= opt.pcg_max_iterations) break;
        const bool timed = opt.profile != 0;
        if (timed)
            while ((int)c->ev.size() < 2 * chunk) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return XRSFM_BA_ENODEV; c->ev.push_back(ev); }
        for (int i = 0; i < chunk; ++i) {
            int e = schur_product(c, d.pp, d.pq, timed, timed ? c->ev[2 * i] : nullptr, timed ? c->ev[2 * i + 1] : nullptr);
            if (e) return e;
            hipLaunchKernelGGL(k_pcg_update, dim3(1), dim3(kPcgThreads), 0, c->stream, d, opt.pcg_tolerance, opt.pcg_max_iterations);
            ++launched;
        }
    }
    sum->pcg_iterations += c->h_st->it;
    return 0;
}

// back-substitute, build the candidate state, evaluate its cost; scalars end up in h_scal
int finish_step(xrsfm_ba_context* c, double huber_a) {
    Dev& d = c->d;
    if (d.n_items > 0) hipLaunchKernelGGL(k_backsub, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), 0, c->stream, d);
    if (d.n_cams > 0) hipLaunchKernelGGL(k_cam_update, dim3(cdiv(d.n_cams, kBlock)), dim3(kBlock), 0, c->stream, d);
    if (d.n_items > 0) hipLaunchKernelGGL(k_cost, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), 0, c->stream, d, huber_a);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(kPcgThreads), 0, c->stream, d.part, d.n_items, d.scal + S_COST_CAND);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(kPcgThreads), 0, c->stream, d.part + 2 * (size_t)d.n_items, d.n_items, d.scal + S_MODEL);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(kPcgThreads), 0, c->stream, d.part + 3 * (size_t)d.n_items, d.n_items, d.scal + S_STEP2_PTS);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(kPcgThreads), 0, c->stream, d.campart, d.n_cams, d.scal + S_STEP2_CAMS);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(kPcgThreads), 0, c->stream, d.campart + d.n_cams, d.n_cams, d.scal + S_XNORM2_CAMS);
    int e = allreduce(c, d.scal + S_COST_CAND, 3, kNcclSum);   // COST_CAND, MODEL, STEP2_PTS adjacent
    if (e) return e;
    return fetch_scalars(c);
}

void print_progress(const xrsfm_ba_options& o, int it, double cost, double change, double gmax, double step, double rho, double radius) {
    if (!o.verbose) return;
    if (it == 0) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n");
    printf("%4d  %.6e  %9.2e  %9.2e  %9.2e  %9.2e  %9.2e\n", it, cost, change, gmax, step, rho, radius);
}

}  // namespace

// ---------------------------------------------------------------- C-ABI
extern "C" {

void xrsfm_ba_default_options(xrsfm_ba_options* o) {
    if (!o) return;
    o->max_iterations = 50;            // ba_solver.cc:627
    o->function_tolerance = 1e-5;      // :628
    o->parameter_tolerance = 1e-6;     // :629
    o->gradient_tolerance = 1e-10;     // Ceres default
    o->initial_radius = 1e4;           // Ceres default
    o->huber_a = 5.99;                 // :343
    o->linear_solver = XRSFM_BA_SOLVER_PCG;
    o->pcg_tolerance = 1e-12;
    o->pcg_max_iterations = 1000;
    o->profile = 0;
    o->verbose = 0;
}

int xrsfm_ba_version(int* n_devices) {
    if (n_devices) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        *n_devices = n;
    }
    return XRSFM_BA_VERSION;
}

void xrsfm_ba_destroy(xrsfm_ba_context* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    for (hipEvent_t e : c->ev) hipEventDestroy(e);
    for (void* p : c->allocs) hipFree(p);
    if (c->h_scal) hipHostFree(c->h_scal);
    if (c->h_st) hipHostFree(c->h_st);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int xrsfm_ba_create(const xrsfm_ba_problem* p, int device, xrsfm_ba_context** out) {
    if (!p || !out) return XRSFM_BA_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[xrsfm_ba] no HIP device visible: the BA path has no CPU fallback\n");
        return XRSFM_BA_ENODEV;
    }
    if (device < 0 || device >= ndev) return XRSFM_BA_EINVAL;
    xrsfm_ba_context* c = new xrsfm_ba_context();
    c->device = device;
    int e = pack_problem(*p, c->pk);
    if (e) { delete c; return e; }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) { delete c; return XRSFM_BA_ENODEV; }
    const Packed& k = c->pk;
    Dev& d = c->d;
    c->n_points_caller = p->n_points;
    d.n_cams = k.n_cams; d.n_pts = k.n_pts; d.n_tiles = k.n_tiles; d.n_slots = k.n_slots; d.n_items = (int)k.items.size() / 2;
    // cameras
    std::vector<CamRec> cams(k.n_cams);
    std::vector<int> model(k.n_cams);
    std::vector<unsigned char> cconst(k.n_cams);
    for (int i = 0; i < k.n_cams; ++i) {
        CamRec& r = cams[i];
        for (int j = 0; j < 4; ++j) r.q[j] = p->cam_q[4 * (size_t)i + j];
        for (int j = 0; j < 3; ++j) r.t[j] = p->cam_t[3 * (size_t)i + j];
        r.pad = 0.0;
        const int ii = p->cam_intr[i];
        for (int j = 0; j < 8; ++j) r.intr[j] = p->intr_params[8 * (size_t)ii + j];
        model[i] = p->intr_model[ii];
        cconst[i] = p->cam_const ? p->cam_const[i] : 0;
    }
    std::vector<double> P(3 * (size_t)k.n_pts);
    for (int j = 0; j < k.n_pts; ++j)
        for (int a = 0; a < 3; ++a) P[3 * (size_t)j + a] = p->points[3 * (size_t)k.pt_orig[j] + a];
#define TRY(x) do { e = (x); if (e) { xrsfm_ba_destroy(c); return e; } } while (0)
    int* tmp_i; double* tmp_d; unsigned char* tmp_u; Item* tmp_it; CamRec* tmp_c;
    TRY(dev_upload(c, &tmp_i, k.slot_cam)); d.slot_cam = tmp_i;
    TRY(dev_upload(c, &tmp_i, k.slot_pt)); d.slot_pt = tmp_i;
    TRY(dev_upload(c, &tmp_i, k.slot_campos)); d.slot_campos = tmp_i;
    TRY(dev_upload(c, &tmp_d, k.slot_u)); d.slot_u = tmp_d;
    TRY(dev_upload(c, &tmp_d, k.slot_v)); d.slot_v = tmp_d;
    TRY(dev_alloc(c, &tmp_it, k.items.size() / 2));
    if (!k.items.empty() && hipMemcpy(tmp_it, k.items.data(), k.items.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { xrsfm_ba_destroy(c); return XRSFM_BA_ENODEV; }
    d.items = tmp_it;
    TRY(dev_upload(c, &tmp_c, cams)); d.cam = tmp_c;
    TRY(dev_upload(c, &tmp_c, cams)); d.cam_cand = tmp_c;
    TRY(dev_upload(c, &tmp_c, cams)); c->cam0 = tmp_c;
    TRY(dev_upload(c, &tmp_i, model)); d.cam_model = tmp_i;
    TRY(dev_upload(c, &tmp_u, cconst)); d.cam_const = tmp_u;
    TRY(dev_upload(c, &tmp_i, k.cam_ptr)); d.cam_ptr = tmp_i;
    TRY(dev_upload(c, &tmp_d, P)); d.P = tmp_d;
    TRY(dev_upload(c, &tmp_d, P)); d.P_cand = tmp_d;
    TRY(dev_upload(c, &tmp_d, P)); c->P0 = tmp_d;
    TRY(dev_upload(c, &tmp_u, k.pt_const)); d.pt_const = tmp_u;
    const size_t ns = (size_t)k.n_slots, nc = (size_t)k.n_cams, np = (size_t)k.n_pts;
    TRY(dev_alloc(c, &d.scale_c, nc * 6)); TRY(dev_alloc(c, &d.scale_p, np * 3));
    TRY(dev_alloc(c, &d.rt, ns * 2)); TRY(dev_alloc(c, &d.Fs, ns * 12)); TRY(dev_alloc(c, &d.Es, ns * 6));
    TRY(dev_alloc(c, &d.Hpp, np * 6)); TRY(dev_alloc(c, &d.gp, np * 3)); TRY(dev_alloc(c, &d.Hinv, np * 6));
    TRY(dev_alloc(c, &d.camlin, nc * 12)); TRY(dev_alloc(c, &d.Dc2, nc * 6)); TRY(dev_alloc(c, &d.camS, nc * 28));
    TRY(dev_alloc(c, &d.Minv, nc * 21)); TRY(dev_alloc(c, &d.b, nc * 6));
    TRY(dev_alloc(c, &d.px, nc * 6)); TRY(dev_alloc(c, &d.pr, nc * 6)); TRY(dev_alloc(c, &d.pz, nc * 6));
    TRY(dev_alloc(c, &d.pp, nc * 6)); TRY(dev_alloc(c, &d.pq, nc * 6));
    TRY(dev_alloc(c, &d.yp, np * 3));
    TRY(dev_alloc(c, &d.scat, (size_t)(k.n_obs > 0 ? k.n_obs : 1) * 28));
    TRY(dev_alloc(c, &d.part, (size_t)d.n_items * 4));
    TRY(dev_alloc(c, &d.campart, nc * 2));
    TRY(dev_alloc(c, &d.scal, (size_t)S_COUNT));
    TRY(dev_alloc(c, &d.st, (size_t)1));
#undef TRY
    if (hipHostMalloc((void**)&c->h_scal, sizeof(double) * S_COUNT) != hipSuccess ||
        hipHostMalloc((void**)&c->h_st, sizeof(PcgStatus)) != hipSuccess) { xrsfm_ba_destroy(c); return XRSFM_BA_ENOMEM; }
    if (hipMemset(d.scal, 0, sizeof(double) * S_COUNT) != hipSuccess || hipMemset(d.st, 0, sizeof(PcgStatus)) != hipSuccess ||
        hipMemset(d.scat, 0, sizeof(double) * 28 * (size_t)(k.n_obs > 0 ? k.n_obs : 1)) != hipSuccess) { xrsfm_ba_destroy(c); return XRSFM_BA_ENODEV; }
    if (hipDeviceSynchronize() != hipSuccess) { xrsfm_ba_destroy(c); return XRSFM_BA_ENODEV; }
    *out = c;
    return XRSFM_BA_OK;
}

int xrsfm_ba_comm_unique_id(unsigned char id[128]) {
    if (!id) return XRSFM_BA_EINVAL;
    if (!load_rccl()) return XRSFM_BA_ECOMM;
    UniqueId u;
    if (((fn_unique_id)g_rccl.GetUniqueId)(&u) != 0) return XRSFM_BA_ECOMM;
    memcpy(id, u.internal, 128);
    return 0;
}

int xrsfm_ba_comm_init(xrsfm_ba_context* c, int n_ranks, int rank, const unsigned char id[128]) {
    if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return XRSFM_BA_EINVAL;
    if (n_ranks == 1) { c->n_ranks = 1; c->rank = 0; return 0; }
    if (!load_rccl()) return XRSFM_BA_ECOMM;
    HIPCHK(hipSetDevice(c->device));
    UniqueId u;
    memcpy(u.internal, id, 128);
    void* comm = nullptr;
    if (((fn_init_rank)g_rccl.CommInitRank)(&comm, n_ranks, u, rank) != 0) return XRSFM_BA_ECOMM;
    c->comm = comm; c->n_ranks = n_ranks; c->rank = rank;
    return 0;
}

int xrsfm_ba_reset(xrsfm_ba_context* c) {
    if (!c) return XRSFM_BA_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d.cam, c->cam0, sizeof(CamRec) * (size_t)c->d.n_cams, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d.P, c->P0, sizeof(double) * 3 * (size_t)c->d.n_pts, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->linearized = false;
    return 0;
}

int xrsfm_ba_download(xrsfm_ba_context* c, double* cam_q, double* cam_t, double* points) {
    if (!c) return XRSFM_BA_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    const Packed& k = c->pk;
    if (cam_q || cam_t) {
        std::vector<CamRec> cams(k.n_cams);
        if (k.n_cams) HIPCHK(hipMemcpy(cams.data(), c->d.cam, sizeof(CamRec) * (size_t)k.n_cams, hipMemcpyDeviceToHost));
        for (int i = 0; i < k.n_cams; ++i) {
            if (cam_q) for (int j = 0; j < 4; ++j) cam_q[4 * (size_t)i + j] = cams[i].q[j];
            if (cam_t) for (int j = 0; j < 3; ++j) cam_t[3 * (size_t)i + j] = cams[i].t[j];
        }
    }
    if (points) {
        std::vector<double> P(3 * (size_t)k.n_pts);
        if (k.n_pts) HIPCHK(hipMemcpy(P.data(), c->d.P, sizeof(double) * P.size(), hipMemcpyDeviceToHost));
        for (int j = 0; j < k.n_pts; ++j)
            for (int a = 0; a < 3; ++a) points[3 * (size_t)k.pt_orig[j] + a] = P[3 * (size_t)j + a];
    }
    return 0;
}

int xrsfm_ba_run(xrsfm_ba_context* c, const xrsfm_ba_options* optp, xrsfm_ba_summary* sum) {
    if (!c || !optp || !sum) return XRSFM_BA_EINVAL;
    const xrsfm_ba_options opt = *optp;
    if (opt.linear_solver != XRSFM_BA_SOLVER_PCG) return XRSFM_BA_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    memset(sum, 0, sizeof(*sum));
    const auto t_begin = std::chrono::steady_clock::now();
    Dev& d = c->d;
    hipStream_t st = c->stream;
    sum->num_residuals = 2 * c->pk.n_obs;
    sum->num_effective_params = 3 * (c->pk.n_var_q + c->pk.n_var_t + c->pk.n_var_p);
    int e;
    auto finish = [&](int term, int reason, double cost) {
        sum->termination = term; sum->termination_reason = reason; sum->final_cost = cost;
        hipStreamSynchronize(st);
        sum->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        return XRSFM_BA_OK;
    };
    // iteration 0: Jacobi scaling from the unscaled column norms, then the scaled linearisation
    hipLaunchKernelGGL(k_fill, dim3(cdiv((long long)d.n_cams * 6, kBlock) + 1), dim3(kBlock), 0, st, d.scale_c, 1.0, (size_t)d.n_cams * 6);
    hipLaunchKernelGGL(k_fill, dim3(cdiv((long long)d.n_pts * 3, kBlock) + 1), dim3(kBlock), 0, st, d.scale_p, 1.0, (size_t)d.n_pts * 3);
    if ((e = linearize(c, opt.huber_a))) return e;
    {
        const long long n = std::max((long long)d.n_cams * 6, (long long)d.n_pts * 3);
        // point norms are local to the rank that owns the track; camera norms were all-reduced in linearize()
        hipLaunchKernelGGL(k_scale_from_norms, dim3(cdiv(n, kBlock) + 1), dim3(kBlock), 0, st, d);
    }
    if ((e = linearize(c, opt.huber_a))) return e;
    c->linearized = true; c->scaled = true;
    double gmax = 0.0;
    if ((e = gradient_max(c, &gmax))) return e;     // also fetches the scalars
    double cost = 0.5 * c->h_scal[S_COST];
    double xnorm2_pts = c->h_scal[S_XNORM2_PTS];
    sum->initial_cost = cost;
    double radius = opt.initial_radius, decrease = 2.0;
    print_progress(opt, 0, cost, 0.0, gmax, 0.0, 0.0, radius);
    if (gmax <= opt.gradient_tolerance) return finish(XRSFM_BA_CONVERGENCE, 1, cost);
    double xnorm = -1.0;   // camera part is produced by k_cam_update of the first step
    int it = 0, invalid = 0;
    const double max_radius = 1e16, min_radius = 1e-32, min_rel_decrease = 1e-3;
    while (true) {
        if (it >= opt.max_iterations) return finish(XRSFM_BA_NO_CONVERGENCE, 5, cost);
        ++it;
        sum->lm_steps_attempted++;
        if ((e = prepare_step(c, radius))) return e;
        if ((e = pcg_solve(c, opt, sum))) return e;
        if ((e = finish_step(c, opt.huber_a))) return e;
        const double* s = c->h_scal;
        const double model_change = s[S_MODEL];
        xnorm = std::sqrt(xnorm2_pts + s[S_XNORM2_CAMS]);
        if (!(model_change > 0.0) || !std::isfinite(model_change)) {
            ++invalid;
            sum->n_unsuccessful++;
            print_progress(opt, it, cost, 0.0, gmax, 0.0, 0.0, radius);
            if (invalid >= 5) return finish(XRSFM_BA_FAILURE, 6, cost);
            radius /= decrease; decrease *= 2.0;
            continue;
        }
        invalid = 0;
        const double cost_cand = 0.5 * s[S_COST_CAND];
        const double step_norm = std::sqrt(s[S_STEP2_PTS] + s[S_STEP2_CAMS]);
        if (step_norm <= opt.parameter_tolerance * (xnorm + opt.parameter_tolerance))
            return finish(XRSFM_BA_CONVERGENCE, 2, cost);
        const double cost_change = cost - cost_cand;
        if (std::fabs(cost_change) <= opt.function_tolerance * cost) return finish(XRSFM_BA_CONVERGENCE, 3, cost);
        const double rel = cost_change / model_change;
        if (rel > min_rel_decrease) {
            std::swap(d.cam, d.cam_cand);
            std::swap(d.P, d.P_cand);
            if ((e = linearize(c, opt.huber_a))) return e;
            if ((e = gradient_max(c, &gmax))) return e;
            cost = 0.5 * c->h_scal[S_COST];
            xnorm2_pts = c->h_scal[S_XNORM2_PTS];
            radius = std::fmin(max_radius, radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
            decrease = 2.0;
            sum->n_successful++;
            print_progress(opt, it, cost, cost_change, gmax, step_norm, rel, radius);
            if (gmax <= opt.gradient_tolerance) return finish(XRSFM_BA_CONVERGENCE, 1, cost);
        } else {
            radius /= decrease; decrease *= 2.0;
            sum->n_unsuccessful++;
            print_progress(opt, it, cost, cost_change, gmax, step_norm, rel, radius);
            if (radius < min_radius) return finish(XRSFM_BA_CONVERGENCE, 4, cost);
        }
    }
}

int xrsfm_ba_solve(const xrsfm_ba_options* opt, xrsfm_ba_problem* problem, xrsfm_ba_summary* summary) {
    if (!opt || !problem || !summary) return XRSFM_BA_EINVAL;
    xrsfm_ba_context* c = nullptr;
    int e = xrsfm_ba_create(problem, 0, &c);
    if (e) return e;
    e = xrsfm_ba_run(c, opt, summary);
    if (!e) e = xrsfm_ba_download(c, problem->cam_q, problem->cam_t, problem->points);
    xrsfm_ba_destroy(c);
    return e;
}

// ---------------------------------------------------------------- diagnostics
int xrsfm_ba_debug_linearize(xrsfm_ba_context* c, double huber_a, int use_scaling, double* r, double* Jc, double* Jp,
                             double* Hpp, double* gp, double* Hcc_diag, double* gc, double* cost) {
    if (!c) return XRSFM_BA_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    Dev& d = c->d;
    hipStream_t st = c->stream;
    int e;
    hipLaunchKernelGGL(k_fill, dim3(cdiv((long long)d.n_cams * 6, kBlock) + 1), dim3(kBlock), 0, st, d.scale_c, 1.0, (size_t)d.n_cams * 6);
    hipLaunchKernelGGL(k_fill, dim3(cdiv((long long)d.n_pts * 3, kBlock) + 1), dim3(kBlock), 0, st, d.scale_p, 1.0, (size_t)d.n_pts * 3);
    if ((e = linearize(c, huber_a))) return e;
    if (use_scaling) {
        const long long n = std::max((long long)d.n_cams * 6, (long long)d.n_pts * 3);
        hipLaunchKernelGGL(k_scale_from_norms, dim3(cdiv(n, kBlock) + 1), dim3(kBlock), 0, st, d);
        if ((e = linearize(c, huber_a))) return e;
    }
    if ((e = fetch_scalars(c))) return e;
    c->linearized = true;
    if (cost) *cost = 0.5 * c->h_scal[S_COST];
    const Packed& k = c->pk;
    const size_t ns = (size_t)k.n_slots;
    auto fetch = [&](const double* dev, size_t n, std::vector<double>& h) -> int {
        h.resize(n);
        if (n && hipMemcpy(h.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return XRSFM_BA_ENODEV;
        return 0;
    };
    std::vector<double> h;
    if (r) {
        if ((e = fetch(d.rt, ns * 2, h))) return e;
        for (size_t s = 0; s < ns; ++s) if (k.slot_obs[s] >= 0) { r[2 * (size_t)k.slot_obs[s]] = h[s]; r[2 * (size_t)k.slot_obs[s] + 1] = h[ns + s]; }
    }
    if (Jc) {
        if ((e = fetch(d.Fs, ns * 12, h))) return e;
        for (size_t s = 0; s < ns; ++s) if (k.slot_obs[s] >= 0) for (int q = 0; q < 12; ++q) Jc[12 * (size_t)k.slot_obs[s] + q] = h[q * ns + s];
    }
    if (Jp) {
        if ((e = fetch(d.Es, ns * 6, h))) return e;
        for (size_t s = 0; s < ns; ++s) if (k.slot_obs[s] >= 0) for (int q = 0; q < 6; ++q) Jp[6 * (size_t)k.slot_obs[s] + q] = h[q * ns + s];
    }
    if (Hpp) {
        if ((e = fetch(d.Hpp, (size_t)k.n_pts * 6, h))) return e;
        for (int j = 0; j < c->n_points_caller; ++j) for (int q = 0; q < 6; ++q) Hpp[6 * (size_t)j + q] = 0.0;
        for (int j = 0; j < k.n_pts; ++j) for (int q = 0; q < 6; ++q) Hpp[6 * (size_t)k.pt_orig[j] + q] = h[6 * (size_t)j + q];
    }
    if (gp) {
        if ((e = fetch(d.gp, (size_t)k.n_pts * 3, h))) return e;
        for (int j = 0; j < c->n_points_caller; ++j) for (int q = 0; q < 3; ++q) gp[3 * (size_t)j + q] = 0.0;
        for (int j = 0; j < k.n_pts; ++j) for (int q = 0; q < 3; ++q) gp[3 * (size_t)k.pt_orig[j] + q] = h[3 * (size_t)j + q];
    }
    if (Hcc_diag || gc) {
        if ((e = fetch(d.camlin, (size_t)k.n_cams * 12, h))) return e;
        for (int i = 0; i < k.n_cams; ++i) for (int q = 0; q < 6; ++q) {
            if (Hcc_diag) Hcc_diag[6 * (size_t)i + q] = h[12 * (size_t)i + q];
            if (gc) gc[6 * (size_t)i + q] = h[12 * (size_t)i + 6 + q];
        }
    }
    return 0;
}

int xrsfm_ba_debug_schur_product(xrsfm_ba_context* c, double radius, const double* x, double* y, double* b) {
    if (!c || !x || !y) return XRSFM_BA_EINVAL;
    if (!c->linearized) return XRSFM_BA_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    Dev& d = c->d;
    int e;
    if ((e = prepare_step(c, radius))) return e;
    hipLaunchKernelGGL(k_pcg_init, dim3(1), dim3(kPcgThreads), 0, c->stream, d);   // clears st->done unless b == 0
    HIPCHK(hipMemsetAsync(d.st, 0, sizeof(PcgStatus), c->stream));
    const size_t n = (size_t)d.n_cams * 6;
    HIPCHK(hipMemcpyAsync(d.pp, x, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((e = schur_product(c, d.pp, d.pq, false, nullptr, nullptr))) return e;
    std::vector<double> q(n), dc(n);
    HIPCHK(hipMemcpyAsync(q.data(), d.pq, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(dc.data(), d.Dc2, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (b) HIPCHK(hipMemcpyAsync(b, d.b, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; ++i) y[i] = q[i] + dc[i] * x[i];
    return 0;
}

}  // extern "C"
