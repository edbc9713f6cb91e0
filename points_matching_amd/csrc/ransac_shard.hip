// ransac_shard.hip — the sharded RANSAC surface (docs/SPEC.md S18): geometry of the one-launch run, the finish
// kernel of a multi-GPU run and the two C-ABI entry points.  Replaces, for a hypothesis-sharded run, the tail of
// `cv::findFundamentalMat` (main.cpp:95-98): every rank scores its id range (pm_ransac_shard_parts_dev -> one
// 80-byte record), the records are all-gathered, every rank finishes (pm_ransac_finish_parts_dev).
#include "ransac_fused_kernels.hpp"

namespace pm_ransac {
namespace {

// ---- finish of a sharded run: winner among the gathered records, its mask over the viewed correspondences -------
// sync[0] = arrival ticket, sync[1] = inlier counter; both return to 0 at the end of the launch.
template <int KIND>
__global__ __launch_bounds__(RF_THREADS) void ransac_finish(pm_points_view v, const pm_ransac_record* __restrict__ recs, int nrec,
                                                            float thr2, unsigned long long* __restrict__ key_out,
                                                            double* __restrict__ F_out, uint8_t* __restrict__ mask, int mask_len,
                                                            int* __restrict__ n_inl_out, int* __restrict__ n_total_out,
                                                            int* __restrict__ sync)
{
    __shared__ int s_offs[PM_MAX_PARTS + 1];
    __shared__ unsigned long long s_wk[RF_THREADS / 64];
    __shared__ int s_wc[RF_THREADS / 64];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (v.parts > 1) view_offsets(v, s_offs, tid);
    unsigned long long kb = 0ull;
    for (int j = tid; j < nrec; j += RF_THREADS) { const unsigned long long kj = recs[j].key; kb = kj > kb ? kj : kb; }
    const unsigned long long kwin = wg_max_u64<RF_THREADS / 64>(kb, s_wk, tid);   // (its barriers also publish s_offs)
    const int n = v.parts > 1 ? s_offs[v.parts] : view_count1(v);
    const bool ok = kwin != 0ull && n >= 8;
    int owner = 0;
    for (int j = 0; j < nrec; ++j) owner = recs[j].key == kwin ? j : owner;   // keys of distinct ids differ
    float fw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fw[i] = ok ? static_cast<float>(recs[owner].F[i]) : 0.f;
    if (blockIdx.x == 0) {
        if (tid < 9 && F_out) F_out[tid] = ok ? recs[owner].F[tid] : 0.0;
        if (tid == 9 && key_out) *key_out = ok ? kwin : 0ull;
        if (tid == 10 && n_total_out) *n_total_out = n;
    }
    int mine = 0;
    for (int i = static_cast<int>(blockIdx.x) * RF_THREADS + tid; i < mask_len; i += static_cast<int>(gridDim.x) * RF_THREADS) {
        bool in = false;
        if (i < n && ok) {
            float2 a, b;
            view_point(v, s_offs, i, a, b);
            in = inlier32<KIND>(fw, a.x, a.y, b.x, b.y, thr2);
        }
        mask[i] = in ? 1 : 0;
        mine += in ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) s_wc[wave] = mine;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < RF_THREADS / 64; ++w) tot += s_wc[w];
        if (tot) __hip_atomic_fetch_add(&sync[1], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int tk = __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = tk == static_cast<int>(gridDim.x) - 1 ? 1 : 0;
        if (s_last) {
            const int all = __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n_inl_out) *n_inl_out = all;
            __hip_atomic_store(&sync[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

// Hypothesis ids per workgroup: whole rounds of one 512-thread workgroup per CU (two teams of four waves: a lone wave
// issues a VALU instruction every 4-8 cycles, two per SIMD fill the vector pipe, and keeping both in ONE workgroup keeps
// them in step — with two 256-thread workgroups per CU the second one's fp64 solve overlapped the first one's scoring
// and stretched the slowest workgroup's score phase from 17k to 25k cycles), at most RF_HB_MAX ids each.
int fused_hb(const pm_ctx* ctx, long long nh)
{
    if (ctx->opts[PM_OPT_RANSAC_WG_IDS] > 0)                  // pinned (throughput form: a pm_batch lane asks for 64)
        return ctx->opts[PM_OPT_RANSAC_WG_IDS] < RF_HB_MAX ? ctx->opts[PM_OPT_RANSAC_WG_IDS] : RF_HB_MAX;
    const long long wgs = ctx->n_cu;
    const long long per_round = wgs * RF_HB_MAX;
    const long long rounds = (nh + per_round - 1) / per_round;
    long long hb = (nh + wgs * rounds - 1) / (wgs * rounds);
    if (hb < 1) hb = 1;
    if (hb > RF_HB_MAX) hb = RF_HB_MAX;
    return static_cast<int>(hb);
}

size_t fused_scratch_bytes(const pm_ctx* ctx, const pm_ransac_params* p)
{
    const long long nh = p->hyp_end - p->hyp_begin;
    const int hb = fused_hb(ctx, nh);
    const long long nwg = (nh + hb - 1) / hb;
    return pm::align_up(sizeof(RfSlot) * static_cast<size_t>(nwg > 0 ? nwg : 1), 256) + pm::align_up(sizeof(FinalOut), 256) + 512;
}

}  // namespace pm_ransac

using namespace pm_ransac;


extern "C" int pm_ransac_shard_parts_dev(pm_ctx* ctx, const pm_points_view* view, const pm_ransac_params* p,
                                         pm_ransac_record* d_record)
{
    PM_REQUIRE(ctx != nullptr && d_record != nullptr && p != nullptr, PM_E_INVALID, "null argument");
    int rc = check_view(view);
    if (rc != PM_OK) return rc;
    PM_REQUIRE(p->hyp_begin >= 0 && p->hyp_end > p->hyp_begin && p->hyp_end <= 0x100000000LL, PM_E_INVALID,
               "hypothesis ids must satisfy 0 <= begin < end <= 2^32");
    PM_REQUIRE(p->error_kind == PM_ERR_SAMPSON || p->error_kind == PM_ERR_SYM_EPIPOLAR, PM_E_INVALID, "unknown error_kind");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    rc = pm::arena_reserve(ctx, fused_scratch_bytes(ctx, p) + 1024);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    return fused_launch(ctx, *view, p, 1, d_record, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
}

extern "C" int pm_ransac_finish_parts_dev(pm_ctx* ctx, const pm_points_view* view, const pm_ransac_params* p,
                                          const pm_ransac_record* d_records, int n_records, uint64_t* d_key, double* d_F,
                                          uint8_t* d_mask, int mask_len, int32_t* d_n_inliers, int32_t* d_n_total)
{
    PM_REQUIRE(ctx != nullptr && p != nullptr && d_records != nullptr && d_mask != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(n_records >= 1 && mask_len >= 0, PM_E_INVALID, "need n_records >= 1, mask_len >= 0");
    int rc = check_view(view);
    if (rc != PM_OK) return rc;
    PM_REQUIRE(p->error_kind == PM_ERR_SAMPSON || p->error_kind == PM_ERR_SYM_EPIPOLAR, PM_E_INVALID, "unknown error_kind");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    int* sync = nullptr;
    rc = sync_words(ctx, &sync);
    if (rc != PM_OK) return rc;
    const float thr2 = p->thresh_px * p->thresh_px;
    int blocks = (mask_len + 8 * RF_THREADS - 1) / (8 * RF_THREADS);
    blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
    pm::ScopedKernelTime t(ctx, "ransac_finish");
    unsigned long long* key = reinterpret_cast<unsigned long long*>(d_key);
    if (p->error_kind == PM_ERR_SAMPSON)
        hipLaunchKernelGGL(ransac_finish<PM_ERR_SAMPSON>, dim3(blocks), dim3(RF_THREADS), 0, ctx->stream, *view, d_records,
                           n_records, thr2, key, d_F, d_mask, mask_len, d_n_inliers, d_n_total, sync + 2);
    else
        hipLaunchKernelGGL(ransac_finish<PM_ERR_SYM_EPIPOLAR>, dim3(blocks), dim3(RF_THREADS), 0, ctx->stream, *view, d_records,
                           n_records, thr2, key, d_F, d_mask, mask_len, d_n_inliers, d_n_total, sync + 2);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}
