// knn_hamming.hip — brute-force Hamming k-NN over packed binary descriptors (ORB-256 = 32 B/row),
// BASELINE config C4; same slot as main.cpp:46.  distance = popcount(q XOR t) reported as float,
// order = (distance, trainIdx) (docs/SPEC.md S2/S3) — integer arithmetic, exact in any order.
//
// Mapping: a lane owns one query (its NW dwords live in VGPRs for the whole sweep); train rows
// are wave-uniform, so they arrive through the scalar cache as SGPR operands of v_xor_b32, and
// v_bcnt_u32_b32 accumulates the popcount in the same instruction.  The train rows are split
// over gridDim.y; each lane keeps its KL best (distance, index) pairs in registers and a small
// merge kernel combines the splits.  k > 4 runs ceil(k/4) passes, each admitting only pairs
// above the last pair the previous pass emitted.
#include "pm_common.hpp"

namespace {

constexpr float HM_INF = __builtin_inff();
constexpr int HM_BIG = 0x7FFFFFFF;

template <int KL>
struct PairList {
    int d[KL];
    int i[KL];
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int c = 0; c < KL; ++c) { d[c] = HM_BIG; i[c] = -1; }
    }
    // rows are scanned in ascending index, so strict < keeps the lower index on equal distance
    __device__ __forceinline__ void insert(int dist, int idx)
    {
        if (dist < d[KL - 1]) {
            d[KL - 1] = dist; i[KL - 1] = idx;
#pragma unroll
            for (int c = KL - 1; c > 0; --c)
                if (d[c] < d[c - 1]) {
                    int t = d[c]; d[c] = d[c - 1]; d[c - 1] = t;
                    int u = i[c]; i[c] = i[c - 1]; i[c - 1] = u;
                }
        }
    }
};

// NW = dwords per descriptor (compile-time) or 0 for the generic loop.
template <int NW, int KL>
__global__ __launch_bounds__(256) void knn_hamming_scan(
    const uint32_t* __restrict__ Q, const uint32_t* __restrict__ T, int nq, int nt, int nw_rt,
    int rows_per_split, const int2* __restrict__ floor_pair, int use_floor, int* __restrict__ part_d,
    int* __restrict__ part_i, int splits)
{
    const int nw = NW ? NW : nw_rt;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int ql = q < nq ? q : nq - 1;
    uint32_t qw[NW ? NW : 1];
    if (NW) {
#pragma unroll
        for (int w = 0; w < NW; ++w) qw[w] = Q[static_cast<size_t>(ql) * NW + w];
    }
    int fd = -1, fi = -1;
    if (use_floor) { const int2 f = floor_pair[ql]; fd = f.x; fi = f.y; }

    PairList<KL> best;
    best.reset();
    const int j0 = blockIdx.y * rows_per_split;
    int j1 = j0 + rows_per_split;
    if (j1 > nt) j1 = nt;
    for (int j = j0; j < j1; ++j) {
        const uint32_t* tr = T + static_cast<size_t>(j) * nw;      // wave-uniform -> s_load
        int dist = 0;
        if (NW) {
#pragma unroll
            for (int w = 0; w < NW; ++w) dist += __builtin_popcount(qw[w] ^ tr[w]);
        } else {
            const uint32_t* qr = Q + static_cast<size_t>(ql) * nw;
            for (int w = 0; w < nw; ++w) dist += __builtin_popcount(qr[w] ^ tr[w]);
        }
        const bool above = !use_floor || dist > fd || (dist == fd && j > fi);
        if (above) best.insert(dist, j);
    }
    if (q < nq) {
        const size_t o = (static_cast<size_t>(q) * splits + blockIdx.y) * KL;
#pragma unroll
        for (int c = 0; c < KL; ++c) { part_d[o + c] = best.d[c]; part_i[o + c] = best.i[c]; }
    }
}

// one thread per query: the KL smallest (d, idx) over splits*KL partial entries
template <int KL>
__global__ __launch_bounds__(256) void knn_hamming_merge(const int* __restrict__ part_d,
                                                         const int* __restrict__ part_i, int nq,
                                                         int splits, int k, int emitted,
                                                         int2* __restrict__ floor_pair,
                                                         pm_match* __restrict__ out)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int n = splits * KL;
    const int* pd = part_d + static_cast<size_t>(q) * n;
    const int* pi = part_i + static_cast<size_t>(q) * n;
    int ld = -1, li = -1;                     // last emitted pair of this pass
    int2 fl = floor_pair[q];
    for (int c = 0; c < KL && emitted + c < k; ++c) {
        int bd = HM_BIG, bi = -1;
        for (int e = 0; e < n; ++e) {
            const int d = pd[e], i = pi[e];
            if (i < 0) continue;
            const bool above = d > ld || (d == ld && i > li);
            const bool better = d < bd || (d == bd && i < bi);
            if (above && (bi < 0 || better)) { bd = d; bi = i; }
        }
        pm_match m;
        m.queryIdx = q;
        m.imgIdx = 0;
        if (bi < 0) { m.trainIdx = -1; m.distance = HM_INF; }
        else { m.trainIdx = bi; m.distance = static_cast<float>(bd); ld = bd; li = bi; fl = int2{bd, bi}; }
        out[static_cast<size_t>(q) * k + emitted + c] = m;
        if (bi < 0) { ld = HM_BIG; li = HM_BIG; }
    }
    floor_pair[q] = fl;
}

template <int KL>
int run_passes(pm_ctx* ctx, const uint32_t* dq, int nq, const uint32_t* dt, int nt, int nw, int k,
               pm_match* dout)
{
    const int qblocks = (nq + 255) / 256;
    int splits = nt > 0 ? (8 * ctx->n_cu + qblocks - 1) / qblocks : 1;   // ~8 waves per SIMD
    if (splits > (nt + 63) / 64) splits = (nt + 63) / 64;
    if (splits > 32) splits = 32;            // the merge walks splits*KL entries per query serially
    if (splits < 1) splits = 1;
    int rows_per_split = (nt + splits - 1) / splits;
    if (rows_per_split < 1) rows_per_split = 1;
    splits = nt > 0 ? (nt + rows_per_split - 1) / rows_per_split : 1;

    const size_t part = sizeof(int) * static_cast<size_t>(nq) * splits * KL;
    const size_t need = 2 * pm::align_up(part, 256) + pm::align_up(sizeof(int2) * nq, 256) + 1024;
    int rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    int* part_d = static_cast<int*>(pm::arena_take(ctx, part));
    int* part_i = static_cast<int*>(pm::arena_take(ctx, part));
    int2* floor_pair = static_cast<int2*>(pm::arena_take(ctx, sizeof(int2) * nq));
    PM_REQUIRE(part_d && part_i && floor_pair, PM_E_NOMEM, "scratch arena too small");
    PM_HIP_CHECK(hipMemsetAsync(floor_pair, 0xFF, sizeof(int2) * nq, ctx->stream));   // (-1,-1)

    dim3 grid(qblocks, splits);
    for (int emitted = 0; emitted < k; emitted += KL) {
        const int use_floor = emitted > 0;
        {
            pm::ScopedKernelTime t(ctx, "knn_hamming");
            switch (nw) {
                case 4:
                    hipLaunchKernelGGL((knn_hamming_scan<4, KL>), grid, dim3(256), 0, ctx->stream, dq, dt, nq, nt, nw,
                                       rows_per_split, floor_pair, use_floor, part_d, part_i, splits);
                    break;
                case 8:
                    hipLaunchKernelGGL((knn_hamming_scan<8, KL>), grid, dim3(256), 0, ctx->stream, dq, dt, nq, nt, nw,
                                       rows_per_split, floor_pair, use_floor, part_d, part_i, splits);
                    break;
                case 16:
                    hipLaunchKernelGGL((knn_hamming_scan<16, KL>), grid, dim3(256), 0, ctx->stream, dq, dt, nq, nt, nw,
                                       rows_per_split, floor_pair, use_floor, part_d, part_i, splits);
                    break;
                default:
                    hipLaunchKernelGGL((knn_hamming_scan<0, KL>), grid, dim3(256), 0, ctx->stream, dq, dt, nq, nt, nw,
                                       rows_per_split, floor_pair, use_floor, part_d, part_i, splits);
            }
            PM_HIP_CHECK(hipGetLastError());
        }
        {
            pm::ScopedKernelTime t(ctx, "knn_hamming_merge");
            hipLaunchKernelGGL(knn_hamming_merge<KL>, dim3(qblocks), dim3(256), 0, ctx->stream, part_d, part_i, nq,
                               splits, k, emitted, floor_pair, dout);
            PM_HIP_CHECK(hipGetLastError());
        }
    }
    return PM_OK;
}

}  // namespace

extern "C" int pm_bf_knn_hamming_u8_dev(pm_ctx* ctx, const uint8_t* dq, int nq, const uint8_t* dt, int nt,
                                        int bytes, int k, pm_match* dout)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(nq >= 0 && nt >= 0 && bytes >= 4 && (bytes % 4) == 0 && k >= 1 && k <= PM_MAX_K, PM_E_INVALID,
               "need nq,nt >= 0, bytes a positive multiple of 4, 1 <= k <= PM_MAX_K");
    PM_REQUIRE(nq == 0 || (dq && dout), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE(nt == 0 || dt, PM_E_INVALID, "null train pointer");
    PM_REQUIRE((reinterpret_cast<uintptr_t>(dq) & 3) == 0 && (reinterpret_cast<uintptr_t>(dt) & 3) == 0,
               PM_E_INVALID, "descriptor buffers must be 4-byte aligned");
    if (nq == 0) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const uint32_t* q32 = reinterpret_cast<const uint32_t*>(dq);
    const uint32_t* t32 = reinterpret_cast<const uint32_t*>(dt);
    const int nw = bytes / 4;
    if (k == 1) return run_passes<1>(ctx, q32, nq, t32, nt, nw, k, dout);
    if (k == 2) return run_passes<2>(ctx, q32, nq, t32, nt, nw, k, dout);
    return run_passes<4>(ctx, q32, nq, t32, nt, nw, k, dout);
}

extern "C" int pm_bf_knn_hamming_u8(pm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes,
                                    int k, pm_match* out)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(nq >= 0 && nt >= 0 && bytes >= 4 && (bytes % 4) == 0 && k >= 1 && k <= PM_MAX_K, PM_E_INVALID,
               "need nq,nt >= 0, bytes a positive multiple of 4, 1 <= k <= PM_MAX_K");
    PM_REQUIRE(nq == 0 || (q && out), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE(nt == 0 || t, PM_E_INVALID, "null train pointer");
    if (nq == 0) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t qb = static_cast<size_t>(nq) * bytes, tb = static_cast<size_t>(nt) * bytes;
    const size_t ob = sizeof(pm_match) * static_cast<size_t>(nq) * k;
    uint8_t *dq = nullptr, *dt = nullptr;
    pm_match* dout = nullptr;
    PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dq), qb));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dt), tb ? tb : 16);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), ob);
    int rc = PM_OK;
    if (e != hipSuccess) { pm::set_error("hipMalloc failed: %s", hipGetErrorString(e)); rc = PM_E_NOMEM; }
    if (rc == PM_OK) {
        e = hipMemcpyAsync(dq, q, qb, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && tb) e = hipMemcpyAsync(dt, t, tb, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { pm::set_error("H2D copy failed: %s", hipGetErrorString(e)); rc = PM_E_HIP; }
    }
    if (rc == PM_OK) rc = pm_bf_knn_hamming_u8_dev(ctx, dq, nq, dt, nt, bytes, k, dout);
    if (rc == PM_OK) {
        e = hipMemcpyAsync(out, dout, ob, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { pm::set_error("D2H copy failed: %s", hipGetErrorString(e)); rc = PM_E_HIP; }
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(dq);
    (void)hipFree(dt);
    (void)hipFree(dout);
    return rc;
}
