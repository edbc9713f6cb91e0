// pm_common.hpp — context, error plumbing, scratch arena and per-kernel event timing shared by
// the translation units of libpm_hip.so.  gfx950 only; no CUDA-compat layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pm.h"

namespace pm {

void set_error(const char* fmt, ...);

#define PM_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ::pm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                            __FILE__, __LINE__);                                        \
            return PM_E_HIP;                                                            \
        }                                                                               \
    } while (0)

#define PM_REQUIRE(cond, status, msg)                                                   \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ::pm::set_error("%s: %s", __func__, msg);                                   \
            return (status);                                                            \
        }                                                                               \
    } while (0)

// The matcher and the compaction kernels tag their side-band words with a per-call epoch that the HOST increments (it
// stands in for a memset per call).  A captured graph freezes that argument: a replay could take the previous replay's
// look-back counts for its own.  So these calls refuse a capturing stream instead of recording something unreplayable.
inline bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
#define PM_REFUSE_CAPTURE(ctx_)                                                                                         \
    PM_REQUIRE(!::pm::stream_is_capturing((ctx_)->stream), PM_E_UNSUPPORTED,                                            \
               "the stream is capturing a graph: this call carries a per-call epoch argument and cannot be replayed")

struct KernelTimer {
    double total_ms = 0.0;
    int launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace pm

constexpr int PM_MAX_DEVICES = 64;      // per-device one-time state (kernel attributes) is kept in arrays of this size

struct pm_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // grow-only device scratch; carved by a bump pointer that every API call resets
    char* arena = nullptr;
    size_t arena_cap = 0;
    size_t arena_off = 0;
    // pinned host staging for small results
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    // timing
    bool timing = false;
    std::map<std::string, pm::KernelTimer> timers;
    std::vector<hipEvent_t> event_pool;
    int n_cu = 256;
    // kNN side-band: [0..1] epoch-tagged 64-bit maxima (never cleared), diag words (re-scan count,
    // non-finite flag) filled only while knn_diag is on
    unsigned long long* knn_stats = nullptr;
    unsigned* knn_diag_words = nullptr;
    unsigned knn_epoch = 0;
    bool knn_diag = false;
    // stable compaction: epoch-tagged per-block survivor counts
    unsigned* fg_counts = nullptr;
    unsigned fg_epoch = 0;
    // compaction fused into the L2 refinement: kf_cap arrival words (8 B) followed by kf_cap epoch-tagged counts (4 B)
    unsigned long long* kf_tile = nullptr;
    int kf_cap = 0;
    unsigned kf_epoch = 0;
    // arrival tickets / counters of the one-launch RANSAC kernels; every launch returns them to zero
    int* sync_words = nullptr;
    int opts[PM_OPT_COUNT_] = {};          // pm_ctx_set_option
    // f32 copies of u8 descriptor rows for the shapes pm_bf_knn_l2_u8 hands to the f32 matcher (grow-only)
    float* widen = nullptr;
    size_t widen_cap = 0;
};

namespace pm {

int arena_reserve(pm_ctx* ctx, size_t bytes);          // ensure capacity (may sync + realloc)
void arena_reset(pm_ctx* ctx);
void* arena_take(pm_ctx* ctx, size_t bytes);            // 256-B aligned carve; nullptr if over cap
int pinned_reserve(pm_ctx* ctx, size_t bytes);

// RAII event bracket: records start/stop on ctx->stream when timing is on.
struct ScopedKernelTime {
    pm_ctx* ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    ScopedKernelTime(pm_ctx* c, const char* n);
    ~ScopedKernelTime();
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#if defined(__HIPCC__)
// Wave-wide reductions on the DPP path (all 64 lanes must be active): four v_min/v_max with a
// row_ror operand reduce each row of 16 lanes, four v_readlane + scalar ops combine the rows; the
// result is wave-uniform.  A __shfl_xor butterfly costs six ds_bpermute round trips (two each for
// 64-bit values) — several hundred cycles per reduction in the one-wave-per-query refinements.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, dpp_u32<0x121>(v));      // row_ror:1
    v = min(v, dpp_u32<0x122>(v));      // row_ror:2
    v = min(v, dpp_u32<0x124>(v));      // row_ror:4
    v = min(v, dpp_u32<0x128>(v));      // row_ror:8
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) { return ~wave_min_u32(~v); }
// lexicographic (high word, low word): exact 64-bit minimum in two 32-bit reductions
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    const unsigned hi = static_cast<unsigned>(v >> 32), lo = static_cast<unsigned>(v);
    const unsigned mh = wave_min_u32(hi);
    const unsigned ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return (static_cast<unsigned long long>(mh) << 32) | ml;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) { return ~wave_min_u64(~v); }
// float minimum with fminf semantics on non-NaN data (the callers' values are finite)
__device__ __forceinline__ float wave_min_f32(float v)
{
#define PM_DPP_F(ctrl) __uint_as_float(dpp_u32<ctrl>(__float_as_uint(v)))
    v = fminf(v, PM_DPP_F(0x121));
    v = fminf(v, PM_DPP_F(0x122));
    v = fminf(v, PM_DPP_F(0x124));
    v = fminf(v, PM_DPP_F(0x128));
#undef PM_DPP_F
    const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
    const float b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
    const float c = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
    const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}
#endif

}  // namespace pm
