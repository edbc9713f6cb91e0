// pair_batch.cpp — a batch of independent image pairs streamed end to end (BASELINE config C5:
// 256 stereo pairs x 4k descriptors): per pair  H2D -> matcher -> ratio + gather -> RANSAC-F -> D2H,
// i.e. one pass of main.cpp:46..98 per pair, with no host round trip inside a pair.
//
// A batch owns a few LANES.  A lane is one pm_ctx (its own HIP stream and scratch arena) plus the
// lane's device buffers and a pinned result slot, so consecutive pairs run on different streams:
// the copy engines move pair j+1 in and pair j-1 out while the kernels of pair j run, and the small
// kernels of neighbouring pairs fill each other's tails.  The host thread only enqueues; it waits
// for a lane when that lane's previous pair has to hand back its result slot.
#include <new>
#include <thread>

#include "pm_common.hpp"

namespace {

struct DevResult {            // one D2H copy per pair
    double F[9];
    uint64_t key;
    int32_t n_good;
    int32_t n_inliers;
};

struct Lane {
    pm_ctx* ctx = nullptr;
    // the inputs of a pair live in ONE device block (din: descriptors 1 | descriptors 2 | keypoints 1 | keypoints 2 at their
    // maximal sizes), so that a job whose four host arrays are one contiguous block travels in ONE copy
    char* din = nullptr;
    size_t din_cap = 0;
    float *dq = nullptr, *dt = nullptr, *dkp1 = nullptr, *dkp2 = nullptr, *dxy1 = nullptr, *dxy2 = nullptr;
    pm_match *dknn = nullptr, *dgood = nullptr;
    uint8_t* dmask = nullptr;
    DevResult* dres = nullptr;
    // pinned
    DevResult* hres = nullptr;
    pm_match* hgood = nullptr;
    uint8_t* hmask = nullptr;
    hipEvent_t done = nullptr;
    int pending = -1;          // job whose results sit in (or are on their way to) the pinned slot
    int pending_n1 = 0;
};

}  // namespace

struct pm_batch {
    int device = 0, n_lanes = 0, max_n1 = 0, max_n2 = 0, dim = 0;
    int desc_u8 = 0;              // pm_batch_set_desc_type: the jobs' descriptor rows are bytes
    int host_threads = 0;         // 0: automatic (two with >= 4 lanes), 1: the calling thread only
    Lane* lanes = nullptr;
};

namespace {

int lane_free(Lane& L)
{
    if (L.ctx) (void)hipStreamSynchronize(L.ctx->stream);
    (void)hipFree(L.din);
    (void)hipFree(L.dxy1); (void)hipFree(L.dxy2); (void)hipFree(L.dknn); (void)hipFree(L.dgood);
    (void)hipFree(L.dmask); (void)hipFree(L.dres);
    if (L.hres) (void)hipHostFree(L.hres);
    if (L.hgood) (void)hipHostFree(L.hgood);
    if (L.hmask) (void)hipHostFree(L.hmask);
    if (L.done) (void)hipEventDestroy(L.done);
    if (L.ctx) (void)pm_ctx_destroy(L.ctx);
    L = Lane{};
    return PM_OK;
}

int lane_init(Lane& L, int device, int max_n1, int max_n2, int dim)
{
    int rc = pm_ctx_create(device, &L.ctx);
    if (rc != PM_OK) return rc;
    // a batch is a throughput job: 32 ids per workgroup (64 workgroups at config C5's 2048 hypotheses) leave CUs to the other
    // lanes' kernels instead of spreading 8 ids over each of the 256 CUs for the lowest latency of ONE run; the fp64 solve
    // costs a wave the same ~21k cycles either way.  Measured (C5, u8 rows, 6 lanes): 31.0k -> 33.1k image pairs/s, with 3
    // lanes 23.7k -> 28.7k; 64 ids lengthen each workgroup's scoring and measured slower (pm_batch_set_option overrides).
    L.ctx->opts[PM_OPT_RANSAC_WG_IDS] = 32;
    const size_t n1 = static_cast<size_t>(max_n1), n2 = static_cast<size_t>(max_n2);
    {
        const size_t a = pm::align_up(sizeof(float) * n1 * dim, 256), b = pm::align_up(sizeof(float) * n2 * dim, 256);
        const size_t c = pm::align_up(sizeof(float) * 2 * n1, 256), d = pm::align_up(sizeof(float) * 2 * n2, 256);
        L.din_cap = a + b + c + d + 1024;                     // (+ the alignment gaps a one-block job may carry)
        PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&L.din), L.din_cap));
        L.dq = reinterpret_cast<float*>(L.din);
        L.dt = reinterpret_cast<float*>(L.din + a);
        L.dkp1 = reinterpret_cast<float*>(L.din + a + b);
        L.dkp2 = reinterpret_cast<float*>(L.din + a + b + c);
    }
    PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&L.dxy1), sizeof(float) * 2 * n1));
    PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&L.dxy2), sizeof(float) * 2 * n1));
    PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&L.dknn), sizeof(pm_match) * 2 * n1));
    PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&L.dgood), sizeof(pm_match) * n1));
    PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&L.dmask), n1));
    PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&L.dres), sizeof(DevResult)));
    PM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&L.hres), sizeof(DevResult), hipHostMallocDefault));
    PM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&L.hgood), sizeof(pm_match) * n1, hipHostMallocDefault));
    PM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&L.hmask), n1, hipHostMallocDefault));
    PM_HIP_CHECK(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    return PM_OK;
}

// hand the lane's finished pair back to the caller's arrays
int lane_collect(pm_batch* b, Lane& L, pm_pair_result* results, pm_match* good, uint8_t* masks)
{
    if (L.pending < 0) return PM_OK;
    PM_HIP_CHECK(hipEventSynchronize(L.done));
    const int j = L.pending;
    pm_pair_result& r = results[j];
    memcpy(r.F, L.hres->F, sizeof(r.F));
    r.best_key = L.hres->key;
    r.n_good = L.hres->n_good;
    r.n_inliers = L.hres->n_inliers;
    r.status = r.n_good < 8 ? PM_E_TOO_FEW : (r.best_key == 0 ? PM_E_NO_MODEL : PM_OK);
    if (good) memcpy(good + static_cast<size_t>(j) * b->max_n1, L.hgood, sizeof(pm_match) * static_cast<size_t>(r.n_good));
    if (masks) memcpy(masks + static_cast<size_t>(j) * b->max_n1, L.hmask, static_cast<size_t>(L.pending_n1));
    L.pending = -1;
    return PM_OK;
}

// jobs j = t mod T of one pm_batch_run call, on lanes t, t + T, ... (one host thread)
int run_part(pm_batch* b, const pm_pair_job* jobs, int n_jobs, int t, int T, float ratio, int knn_flags, const pm_ransac_params* p,
             pm_pair_result* results, pm_match* good, uint8_t* masks)
{
    PM_HIP_CHECK(hipSetDevice(b->device));
    int rc = PM_OK;
    // A failing HIP call must not leave the loop by `return`: copies may still be in flight on the caller's buffers
    // and a lane would keep a stale `pending` job for the next call.  Every failure falls through to the drain below.
#define PM_BATCH_HIP(expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess && rc == PM_OK) {                                                 \
            pm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = PM_E_HIP;                                                                     \
        }                                                                                      \
    } while (0)
    const int lanes_mine = (b->n_lanes - t + T - 1) / T;          // lanes t, t + T, ...
    for (int j = t, c = 0; j < n_jobs && rc == PM_OK; j += T, ++c) {
        Lane& L = b->lanes[t + T * (c % lanes_mine)];
        rc = lane_collect(b, L, results, good, masks);
        if (rc != PM_OK) break;
        const pm_pair_job& jb = jobs[j];
        hipStream_t s = L.ctx->stream;
        const size_t n1 = static_cast<size_t>(jb.n1), n2 = static_cast<size_t>(jb.n2);
        const size_t esz = b->desc_u8 ? 1 : sizeof(float);
        const size_t qb = esz * n1 * b->dim, tb = esz * n2 * b->dim, k1b = sizeof(float) * 2 * n1, k2b = sizeof(float) * 2 * n2;
        // One block?  desc1 | desc2 | kp1 | kp2 in this order inside one host allocation, sections aligned for the kernels'
        // vector loads and (nearly) gap-free: the pair then travels in ONE copy (a 4k x 4k f32 pair: 4.3 MB instead of
        // 2 + 2 + 0.03 + 0.03 MB — the link's rate grows with the copy size), at the host's own offsets.
        const char* h0 = reinterpret_cast<const char*>(jb.desc1);
        const ptrdiff_t ot = reinterpret_cast<const char*>(jb.desc2) - h0, o1 = reinterpret_cast<const char*>(jb.kp1_xy) - h0,
                        o2 = reinterpret_cast<const char*>(jb.kp2_xy) - h0;
        const bool one_block = ot >= static_cast<ptrdiff_t>(qb) && o1 >= ot + static_cast<ptrdiff_t>(tb) &&
                               o2 >= o1 + static_cast<ptrdiff_t>(k1b) && (ot & 15) == 0 && (o1 & 7) == 0 && (o2 & 7) == 0 &&
                               static_cast<size_t>(o2) + k2b <= qb + tb + k1b + k2b + 768 && static_cast<size_t>(o2) + k2b <= L.din_cap;
        const void *dq = L.dq, *dt = L.dt;
        const float *dkp1 = L.dkp1, *dkp2 = L.dkp2;
        if (one_block) {
            PM_BATCH_HIP(hipMemcpyAsync(L.din, h0, static_cast<size_t>(o2) + k2b, hipMemcpyHostToDevice, s));
            dq = L.din; dt = L.din + ot;
            dkp1 = reinterpret_cast<const float*>(L.din + o1);
            dkp2 = reinterpret_cast<const float*>(L.din + o2);
        } else {
            PM_BATCH_HIP(hipMemcpyAsync(L.dq, jb.desc1, qb, hipMemcpyHostToDevice, s));
            PM_BATCH_HIP(hipMemcpyAsync(L.dt, jb.desc2, tb, hipMemcpyHostToDevice, s));
            PM_BATCH_HIP(hipMemcpyAsync(L.dkp1, jb.kp1_xy, k1b, hipMemcpyHostToDevice, s));
            PM_BATCH_HIP(hipMemcpyAsync(L.dkp2, jb.kp2_xy, k2b, hipMemcpyHostToDevice, s));
        }
        if (rc == PM_OK) {
            if (b->desc_u8)
                rc = pm_bf_knn_l2_u8_ratio_dev(L.ctx, static_cast<const uint8_t*>(dq), jb.n1, static_cast<const uint8_t*>(dt), jb.n2,
                                               b->dim, ratio, dkp1, dkp2, L.dknn, L.dgood, L.dxy1, L.dxy2, &L.dres->n_good);
            else
                rc = pm_bf_knn_l2_ratio_dev(L.ctx, static_cast<const float*>(dq), jb.n1, static_cast<const float*>(dt), jb.n2, b->dim,
                                            knn_flags, ratio, dkp1, dkp2, L.dknn, L.dgood, L.dxy1, L.dxy2, &L.dres->n_good);
        }
        if (rc == PM_OK)
            rc = pm_ransac_run_dev(L.ctx, L.dxy1, L.dxy2, jb.n1, &L.dres->n_good, p, &L.dres->key, L.dres->F, L.dmask,
                                   &L.dres->n_inliers);
        if (rc != PM_OK) break;
        PM_BATCH_HIP(hipMemcpyAsync(L.hres, L.dres, sizeof(DevResult), hipMemcpyDeviceToHost, s));
        if (good) PM_BATCH_HIP(hipMemcpyAsync(L.hgood, L.dgood, sizeof(pm_match) * n1, hipMemcpyDeviceToHost, s));
        if (masks) PM_BATCH_HIP(hipMemcpyAsync(L.hmask, L.dmask, n1, hipMemcpyDeviceToHost, s));
        PM_BATCH_HIP(hipEventRecord(L.done, s));
        if (rc != PM_OK) break;
        L.pending = j;
        L.pending_n1 = jb.n1;
    }
#undef PM_BATCH_HIP
    // drain in job order (also after an error: nothing may stay in flight on the caller's buffers)
    for (int i = t; i < b->n_lanes; i += T) {
        if (rc == PM_OK) rc = lane_collect(b, b->lanes[i], results, good, masks);
        else { (void)hipStreamSynchronize(b->lanes[i].ctx->stream); b->lanes[i].pending = -1; }
    }
    return rc;
}


}  // namespace

extern "C" int pm_batch_create(int device, int n_lanes, int max_n1, int max_n2, int dim, pm_batch** out)
{
    PM_REQUIRE(out != nullptr, PM_E_INVALID, "out is null");
    PM_REQUIRE(n_lanes >= 1 && n_lanes <= 16 && max_n1 >= 1 && max_n2 >= 1 && dim >= 1, PM_E_INVALID,
               "need 1 <= n_lanes <= 16, max_n1, max_n2, dim >= 1");
    PM_HIP_CHECK(hipSetDevice(device));
    pm_batch* b = new (std::nothrow) pm_batch;
    PM_REQUIRE(b != nullptr, PM_E_NOMEM, "out of host memory");
    b->device = device; b->n_lanes = n_lanes; b->max_n1 = max_n1; b->max_n2 = max_n2; b->dim = dim;
    b->lanes = new (std::nothrow) Lane[n_lanes];
    if (!b->lanes) { delete b; pm::set_error("out of host memory"); return PM_E_NOMEM; }
    for (int i = 0; i < n_lanes; ++i) {
        const int rc = lane_init(b->lanes[i], device, max_n1, max_n2, dim);
        if (rc != PM_OK) { (void)pm_batch_destroy(b); return rc; }
    }
    *out = b;
    return PM_OK;
}

extern "C" int pm_batch_set_option(pm_batch* b, int option, int value)
{
    PM_REQUIRE(b != nullptr, PM_E_INVALID, "batch is null");
    for (int i = 0; i < b->n_lanes; ++i) {
        const int rc = pm_ctx_set_option(b->lanes[i].ctx, option, value);
        if (rc != PM_OK) return rc;
    }
    return PM_OK;
}

extern "C" int pm_batch_set_host_threads(pm_batch* b, int n)
{
    PM_REQUIRE(b != nullptr && n >= 0 && n <= 2, PM_E_INVALID, "host threads: 0 = automatic, 1, 2");
    b->host_threads = n;
    return PM_OK;
}

extern "C" int pm_batch_set_desc_type(pm_batch* b, int desc_u8)
{
    PM_REQUIRE(b != nullptr && (desc_u8 == 0 || desc_u8 == 1), PM_E_INVALID, "descriptor type: 0 = float32 rows, 1 = uint8 rows");
    b->desc_u8 = desc_u8;
    return PM_OK;
}

extern "C" int pm_batch_destroy(pm_batch* b)
{
    if (!b) return PM_OK;
    (void)hipSetDevice(b->device);
    for (int i = 0; i < b->n_lanes; ++i) lane_free(b->lanes[i]);
    delete[] b->lanes;
    delete b;
    return PM_OK;
}

extern "C" int pm_batch_run(pm_batch* b, const pm_pair_job* jobs, int n_jobs, float ratio, int knn_flags,
                            const pm_ransac_params* p, pm_pair_result* results, pm_match* good, uint8_t* masks)
{
    PM_REQUIRE(b != nullptr && p != nullptr, PM_E_INVALID, "null batch / params");
    PM_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || (jobs && results)), PM_E_INVALID, "null jobs / results");
    for (int j = 0; j < n_jobs; ++j) {
        const pm_pair_job& jb = jobs[j];
        PM_REQUIRE(jb.n1 >= 1 && jb.n1 <= b->max_n1 && jb.n2 >= 1 && jb.n2 <= b->max_n2, PM_E_INVALID,
                   "a pair exceeds the batch's max_n1 / max_n2 (or is empty)");
        PM_REQUIRE(jb.desc1 && jb.desc2 && jb.kp1_xy && jb.kp2_xy, PM_E_INVALID, "null pointer in a pair job");
    }
    // Host threads: enqueueing a pair costs the host ~30 us (copies + five launches + the event), which at config C5's u8
    // rows is what bounds the batch once the GPU work per pair is ~20 us.  With >= 4 lanes the jobs are dealt to TWO host
    // threads, each with its own half of the lanes (thread t: jobs j = t mod T on lanes t, t + T, ...): nothing is shared
    // but the device.  Results land in job order either way.
    const int T = (b->n_lanes >= 4 && n_jobs >= 16 && b->host_threads != 1) ? 2 : 1;
    int rcs[2] = {PM_OK, PM_OK};
    char errs[2][256] = {"", ""};
    auto work = [&](int t) {
        rcs[t] = run_part(b, jobs, n_jobs, t, T, ratio, knn_flags, p, results, good, masks);
        if (rcs[t] != PM_OK) snprintf(errs[t], sizeof errs[t], "%s", pm_last_error());     // (the error text is per thread)
    };
    if (T == 1) {
        work(0);
    } else {
        bool started = false;
        std::thread other;
        try { other = std::thread(work, 1); started = true; } catch (...) { }     // (the C ABI never throws)
        work(0);
        if (started) other.join();
        else work(1);                                        // no second thread: the calling thread takes the other half too
    }
    for (int t = 0; t < T; ++t)
        if (rcs[t] != PM_OK) { pm::set_error("%s", errs[t]); return rcs[t]; }
    return PM_OK;
}

extern "C" int pm_host_register(void* ptr, size_t bytes)
{
    PM_REQUIRE(ptr != nullptr && bytes > 0, PM_E_INVALID, "null pointer / zero size");
    PM_HIP_CHECK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return PM_OK;
}

extern "C" int pm_host_unregister(void* ptr)
{
    PM_REQUIRE(ptr != nullptr, PM_E_INVALID, "null pointer");
    PM_HIP_CHECK(hipHostUnregister(ptr));
    return PM_OK;
}
