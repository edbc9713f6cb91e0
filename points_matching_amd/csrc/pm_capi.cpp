// pm_capi.cpp — context management and the host-side O(n) stages of the path:
// strong-match filter (main.cpp:49-69), match list / index vectors (main.cpp:71-79), point
// gather (main.cpp:89-91), residual report (main.cpp:103-123), epipolar lines (main.cpp:127-142).
// These are a few thousand scalar operations; they stay on the host like in the reference.
#include <cfloat>
#include <cmath>

#include "pm_common.hpp"

namespace pm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int arena_reserve(pm_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->arena_cap) return PM_OK;
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->arena) PM_HIP_CHECK(hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_cap = 0;
    size_t cap = align_up(bytes + bytes / 4, size_t(1) << 20);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->arena), cap);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        return PM_E_NOMEM;
    }
    ctx->arena_cap = cap;
    return PM_OK;
}

void arena_reset(pm_ctx* ctx) { ctx->arena_off = 0; }

void* arena_take(pm_ctx* ctx, size_t bytes)
{
    size_t off = align_up(ctx->arena_off, 256);
    if (off + bytes > ctx->arena_cap) return nullptr;
    ctx->arena_off = off + bytes;
    return ctx->arena + off;
}

int pinned_reserve(pm_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->pinned_cap) return PM_OK;
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->pinned) PM_HIP_CHECK(hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_cap = 0;
    size_t cap = align_up(bytes, 4096);
    PM_HIP_CHECK(hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault));
    ctx->pinned_cap = cap;
    return PM_OK;
}

ScopedKernelTime::ScopedKernelTime(pm_ctx* c, const char* n) : ctx(c), name(n)
{
    if (!ctx->timing) return;
    auto grab = [&]() -> hipEvent_t {
        if (!ctx->event_pool.empty()) {
            hipEvent_t e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    };
    a = grab();
    b = grab();
    if (a) (void)hipEventRecord(a, ctx->stream);
}

ScopedKernelTime::~ScopedKernelTime()
{
    if (!ctx->timing || !a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->timers[name].pending.emplace_back(a, b);
}

static void drain_timers(pm_ctx* ctx)
{
    for (auto& kv : ctx->timers) {
        for (auto& ab : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ab.first, ab.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.launches += 1;
            }
            ctx->event_pool.push_back(ab.first);
            ctx->event_pool.push_back(ab.second);
        }
        kv.second.pending.clear();
    }
}

}  // namespace pm

using namespace pm;

extern "C" {

const char* pm_last_error(void) { return g_err; }

const char* pm_status_string(int s)
{
    switch (s) {
        case PM_OK: return "ok";
        case PM_E_INVALID: return "invalid argument";
        case PM_E_TOO_FEW: return "fewer than 8 correspondences";
        case PM_E_NO_MODEL: return "no valid model in the hypothesis range";
        case PM_E_HIP: return "HIP runtime error";
        case PM_E_NOMEM: return "out of device memory";
        case PM_E_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

int pm_version(void) { return PM_VERSION_MAJOR * 100 + PM_VERSION_MINOR; }

int pm_ctx_create(int device, pm_ctx** out)
{
    PM_REQUIRE(out != nullptr, PM_E_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    PM_HIP_CHECK(hipGetDeviceCount(&n));
    PM_REQUIRE(device >= 0 && device < n && device < PM_MAX_DEVICES, PM_E_INVALID, "no such HIP device");
    PM_HIP_CHECK(hipSetDevice(device));
    pm_ctx* c = new (std::nothrow) pm_ctx();
    PM_REQUIRE(c != nullptr, PM_E_NOMEM, "host allocation failed");
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete c;
        return PM_E_HIP;
    }
    c->stream = c->own_stream;
    e = hipMalloc(reinterpret_cast<void**>(&c->knn_stats), 64);
    if (e == hipSuccess) e = hipMemset(c->knn_stats, 0, 64);
    if (e != hipSuccess) {
        set_error("hipMalloc failed: %s", hipGetErrorString(e));
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return PM_E_NOMEM;
    }
    c->knn_diag_words = reinterpret_cast<unsigned*>(c->knn_stats + 4);
    *out = c;
    return PM_OK;
}

int pm_ctx_destroy(pm_ctx* ctx)
{
    if (!ctx) return PM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    drain_timers(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->knn_stats) (void)hipFree(ctx->knn_stats);
    if (ctx->fg_counts) (void)hipFree(ctx->fg_counts);
    if (ctx->sync_words) (void)hipFree(ctx->sync_words);
    if (ctx->kf_tile) (void)hipFree(ctx->kf_tile);
    if (ctx->widen) (void)hipFree(ctx->widen);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return PM_OK;
}

int pm_ctx_set_stream(pm_ctx* ctx, void* hip_stream)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return PM_OK;
}

int pm_ctx_synchronize(pm_ctx* ctx)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

int pm_ctx_timing_enable(pm_ctx* ctx, int enable)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    ctx->timing = enable != 0;
    return PM_OK;
}

int pm_ctx_timing_reset(pm_ctx* ctx)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    for (auto& kv : ctx->timers) { kv.second.total_ms = 0.0; kv.second.launches = 0; }
    return PM_OK;
}

int pm_ctx_timing_get(pm_ctx* ctx, const char* kernel, double* mean_ms, int* launches)
{
    PM_REQUIRE(ctx != nullptr && kernel != nullptr, PM_E_INVALID, "null argument");
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    auto it = ctx->timers.find(kernel);
    double ms = 0.0;
    int n = 0;
    if (it != ctx->timers.end()) {
        n = it->second.launches;
        ms = n ? it->second.total_ms / n : 0.0;
    }
    if (mean_ms) *mean_ms = ms;
    if (launches) *launches = n;
    return PM_OK;
}

int pm_ctx_set_option(pm_ctx* ctx, int option, int value)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(option >= 1 && option < PM_OPT_COUNT_, PM_E_INVALID, "unknown option");
    const int vmax = option == PM_OPT_RANSAC_WG_IDS ? 128 : option == PM_OPT_KNN_RING_PROLOGUE ? 8 : option == PM_OPT_KNN_RING ? 6 : (option == PM_OPT_KNN_F16_WAVES || option == PM_OPT_KNN_U8_GROUP) ? 3 : 2;
    PM_REQUIRE(value >= 0 && value <= vmax, PM_E_INVALID,
               "option value out of range (0 = automatic, 1, 2; KNN_F16_WAVES, KNN_U8_GROUP: .. 3)");
    ctx->opts[option] = value;
    return PM_OK;
}

int pm_ctx_get_option(pm_ctx* ctx, int option, int* value)
{
    PM_REQUIRE(ctx != nullptr && value != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(option >= 1 && option < PM_OPT_COUNT_, PM_E_INVALID, "unknown option");
    *value = ctx->opts[option];
    return PM_OK;
}

int pm_ctx_knn_diag_enable(pm_ctx* ctx, int enable)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    ctx->knn_diag = enable != 0;
    return PM_OK;
}

// The compaction that rides the refinement launch (pm_bf_knn_l2_ratio_dev without a record buffer) bounds its look-back
// polls; a poll that ran out stores the call's epoch at byte 48 of the side-band block.  *gave_up != 0: the LAST such call
// on this context placed survivors with an incomplete prefix — its outputs must not be used (never observed).
int pm_ctx_filter_fusion_status(pm_ctx* ctx, int* gave_up)
{
    PM_REQUIRE(ctx != nullptr && gave_up != nullptr, PM_E_INVALID, "null argument");
    unsigned w = 0;
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PM_HIP_CHECK(hipMemcpy(&w, reinterpret_cast<const char*>(ctx->knn_stats) + 48, sizeof w, hipMemcpyDeviceToHost));
    *gave_up = (ctx->kf_epoch != 0u && w == ctx->kf_epoch) ? 1 : 0;
    return PM_OK;
}

int pm_ctx_knn_stats(pm_ctx* ctx, int* rescans, int* nonfinite)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    unsigned h[2] = {0, 0};
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PM_HIP_CHECK(hipMemcpy(h, ctx->knn_diag_words, sizeof h, hipMemcpyDeviceToHost));
    if (rescans) *rescans = static_cast<int>(h[0]);
    if (nonfinite) *nonfinite = static_cast<int>(h[1]);
    return PM_OK;
}

int pm_ctx_knn_route(pm_ctx* ctx, int* route)
{
    PM_REQUIRE(ctx != nullptr && route != nullptr, PM_E_INVALID, "null argument");
    unsigned h = 0;
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PM_HIP_CHECK(hipMemcpy(&h, ctx->knn_diag_words + 2, sizeof h, hipMemcpyDeviceToHost));
    *route = static_cast<int>(h);
    return PM_OK;
}

// ---- strong-match filters (slot of main.cpp:49-69) -----------------------------------------

int pm_filter_midpoint(const pm_match* m, int n, double* min_out, double* max_out,
                       pm_match* out, int* n_out)
{
    PM_REQUIRE(n >= 0 && (n == 0 || (m && out)) && n_out, PM_E_INVALID, "bad argument");
    // main.cpp:49-50 start the scan at 1 / 0, so distances that are all > 1 leave lo == 1.
    double lo = 1, hi = 0;
    for (int i = 0; i < n; ++i) {
        const double d = m[i].distance;
        if (lo > d) lo = d;
        if (hi < d) hi = d;
    }
    const double cut = lo + (hi - lo) / 2;     // main.cpp:65
    int kept = 0;
    for (int i = 0; i < n; ++i)
        if (static_cast<double>(m[i].distance) < cut) out[kept++] = m[i];
    if (min_out) *min_out = lo;
    if (max_out) *max_out = hi;
    *n_out = kept;
    return PM_OK;
}

int pm_filter_ratio(const pm_match* knn, int nq, int k, float ratio, pm_match* out, int* n_out)
{
    PM_REQUIRE(nq >= 0 && k >= 2 && (nq == 0 || (knn && out)) && n_out, PM_E_INVALID,
               "bad argument (ratio test needs k >= 2)");
    int kept = 0;
    for (int i = 0; i < nq; ++i) {
        const pm_match& best = knn[static_cast<size_t>(i) * k];
        const pm_match& second = knn[static_cast<size_t>(i) * k + 1];
        if (best.trainIdx < 0 || second.trainIdx < 0) continue;
        const float rhs = ratio * second.distance;
        if (best.distance < rhs) out[kept++] = best;
    }
    *n_out = kept;
    return PM_OK;
}

// ---- match list + gather (main.cpp:71-79, :89-91) ------------------------------------------

int pm_match_indices(const pm_match* m, int n, int32_t* query_idx, int32_t* train_idx)
{
    PM_REQUIRE(n >= 0 && (n == 0 || (m && query_idx && train_idx)), PM_E_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) {
        query_idx[i] = m[i].queryIdx;
        train_idx[i] = m[i].trainIdx;
    }
    return PM_OK;
}

int pm_gather_points(const float* kp_xy, int n_kp, const int32_t* idx, int n, float* out_xy)
{
    PM_REQUIRE(n >= 0 && n_kp >= 0 && (n == 0 || (kp_xy && idx && out_xy)), PM_E_INVALID,
               "bad argument");
    for (int i = 0; i < n; ++i) {
        const int32_t j = idx[i];
        PM_REQUIRE(j >= 0 && j < n_kp, PM_E_INVALID, "keypoint index out of range");
        out_xy[2 * i] = kp_xy[2 * static_cast<size_t>(j)];
        out_xy[2 * i + 1] = kp_xy[2 * static_cast<size_t>(j) + 1];
    }
    return PM_OK;
}

long pm_format_match_list(const pm_match* m, int n, char* buf, size_t cap)
{
    if (n < 0 || (n > 0 && !m)) return -1;
    std::string s = "Good Matches are:\n";          // main.cpp:73
    char line[128];
    for (int i = 0; i < n; ++i) {
        snprintf(line, sizeof line, "-- Good Match [%d] Keypoint 1: %d  -- Keypoint 2: %d  \n", i,
                 m[i].queryIdx, m[i].trainIdx);     // main.cpp:76 (two double spaces + trailing)
        s += line;
    }
    if (buf && cap) {
        size_t w = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), w);
        buf[w] = 0;
    }
    return static_cast<long>(s.size());
}

// ---- residual report (main.cpp:103-123) ----------------------------------------------------

int pm_epipolar_residuals(const float* xy1, const float* xy2, int n, const double F[9],
                          int transposed, double* r, double* mean_abs)
{
    PM_REQUIRE(n >= 0 && F && (n == 0 || (xy1 && xy2)), PM_E_INVALID, "bad argument");
    const float* pa = transposed ? xy1 : xy2;   // the 1x3 row vector  (main.cpp:110-112)
    const float* pb = transposed ? xy2 : xy1;   // the 3x1 column      (main.cpp:113-115)
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        const double xa = pa[2 * i], ya = pa[2 * i + 1];
        const double xb = pb[2 * i], yb = pb[2 * i + 1];
        // row * F, accumulated in k order, then (row*F) * column     (main.cpp:117)
        double v[3];
        for (int j = 0; j < 3; ++j) {
            double t = xa * F[j] + ya * F[3 + j];
            v[j] = t + F[6 + j];
        }
        double t = v[0] * xb + v[1] * yb;
        const double res = t + v[2];
        if (r) r[i] = res;
        acc += std::fabs(res);                   // main.cpp:120
    }
    if (mean_abs) *mean_abs = acc / n;           // main.cpp:123 (n == 0 prints nan there too)
    return PM_OK;
}

int pm_f_scale_f33(double F[9])
{
    PM_REQUIRE(F != nullptr, PM_E_INVALID, "F is null");
    if (std::fabs(F[8]) > DBL_EPSILON) {
        const double inv = 1.0 / F[8];
        for (int i = 0; i < 9; ++i) F[i] = F[i] * inv;
    }
    return PM_OK;
}

// ---- epipolar lines (main.cpp:127-142) -----------------------------------------------------

int pm_epilines(const float* xy, int n, int which_image, const double F[9], float* lines)
{
    PM_REQUIRE(n >= 0 && F && (which_image == 1 || which_image == 2) && (n == 0 || (xy && lines)),
               PM_E_INVALID, "bad argument");
    const int rs = which_image == 1 ? 3 : 1;    // l = F x  or  F^T x
    const int cs = which_image == 1 ? 1 : 3;
    for (int i = 0; i < n; ++i) {
        const double x = xy[2 * i], y = xy[2 * i + 1];
        double l[3];
        for (int j = 0; j < 3; ++j) {
            double t = F[j * rs] * x + F[j * rs + cs] * y;
            l[j] = t + F[j * rs + 2 * cs];
        }
        double nu = l[0] * l[0] + l[1] * l[1];
        nu = nu != 0.0 ? 1.0 / std::sqrt(nu) : 1.0;
        for (int j = 0; j < 3; ++j) lines[3 * i + j] = static_cast<float>(l[j] * nu);
    }
    return PM_OK;
}

static int32_t float_to_int_trunc(float v)
{
    // x86 cvttss2si semantics (what the implicit float->int conversions at main.cpp:138-140
    // compile to): out-of-range and NaN give INT_MIN.
    if (!(v == v) || v >= 2147483648.0f || v <= -2147483904.0f) return INT32_MIN;
    return static_cast<int32_t>(v);
}

int pm_epiline_endpoints(const float* lines, int n, int cols, int32_t* xyxy)
{
    PM_REQUIRE(n >= 0 && (n == 0 || (lines && xyxy)), PM_E_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) {
        const float a = lines[3 * i], b = lines[3 * i + 1], c = lines[3 * i + 2];
        xyxy[4 * i] = 0;
        xyxy[4 * i + 1] = float_to_int_trunc(-c / b);
        xyxy[4 * i + 2] = cols;
        xyxy[4 * i + 3] = float_to_int_trunc(-(c + a * static_cast<float>(cols)) / b);
    }
    return PM_OK;
}

}  // extern "C"

/* OpenCV 2.4 CvModelEstimator2::runLMeDS iteration count [recalled]: 300 for (0.99, 0.45). */
extern "C" int pm_lmeds_default_iters(double confidence, double outlier_ratio)
{
    if (!(confidence > 0.0) || !(confidence < 1.0) || !(outlier_ratio >= 0.0) || !(outlier_ratio < 1.0)) return -1;
    const double w7 = std::pow(1.0 - outlier_ratio, 7.0);
    const double den = std::log(1.0 - w7);
    if (!(den < 0.0)) return 1;
    const double it = std::log(1.0 - confidence) / den;
    if (!(it < 1e9)) return 1000000000;
    const long r = std::lround(it);
    return static_cast<int>(r < 1 ? 1 : r);
}
