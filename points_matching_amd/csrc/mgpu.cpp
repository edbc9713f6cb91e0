// mgpu.cpp — the path over the GPUs of one node behind the C ABI (SURVEY.md 8b `pm_ransac_reduce`, 8e): one host
// thread, one pm_ctx per device, ONE RCCL communicator set (ncclCommInitAll), collectives enqueued on the contexts'
// streams inside ncclGroupStart/End.  Slot in the reference: main.cpp:46 (the query rows of the matcher shard) and
// main.cpp:95-98 (the hypothesis ids of the robust estimator shard).
//
//   matcher      query rows cut into n_dev contiguous blocks, train set replicated; each device matches and filters
//                its block straight into ITS slot of the gathered buffer, then ncclAllGather #1 (in place) hands every
//                device every block: [count | xy1 | xy2 | match records].
//   RANSAC       device g samples/solves/scores ids [H*g/G, H*(g+1)/G) over ALL gathered correspondences (read through
//                a pm_points_view: no concatenation pass), leaves the shard's 80-byte (key, F) record in its slot;
//                ncclAllGather #2 (in place) = the arg-max all-reduce of SURVEY 8e carried with its payload; every
//                device then picks the winner and writes the inlier mask (pm_ransac_finish_parts_dev).  Nobody
//                re-solves, nothing is broadcast.
// RCCL is bound at run time (dlopen "librccl.so.1": the copy already in the process when there is one, e.g. under
// PyTorch-ROCm, else /opt/rocm's), so single-GPU users of libpm_hip.so never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <new>
#include <vector>

#include "pm_common.hpp"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int rccl_load(Rccl& r)
{
    if (r.lib) return PM_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { pm::set_error("cannot load librccl.so.1: %s", dlerror()); return PM_E_UNSUPPORTED; }
#define PM_SYM(field, name)                                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));                          \
    if (!r.field) { pm::set_error("librccl lacks %s", name); return PM_E_UNSUPPORTED; }
    PM_SYM(CommInitAll, "ncclCommInitAll")
    PM_SYM(CommDestroy, "ncclCommDestroy")
    PM_SYM(AllGather, "ncclAllGather")
    PM_SYM(GroupStart, "ncclGroupStart")
    PM_SYM(GroupEnd, "ncclGroupEnd")
    PM_SYM(GetErrorString, "ncclGetErrorString")
#undef PM_SYM
    return PM_OK;
}

Rccl g_rccl;

#define PM_NCCL_CHECK(expr)                                                                     \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            ::pm::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return PM_E_HIP;                                                                    \
        }                                                                                       \
    } while (0)

struct Dev {
    int device = 0;
    pm_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    char* buf = nullptr;          // grow-only: descriptors, keypoints, k-NN records, gathered blocks, records, outputs
    size_t cap = 0;
};

int dev_reserve(Dev& d, size_t bytes)
{
    if (bytes <= d.cap) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(d.device));
    PM_HIP_CHECK(hipStreamSynchronize(d.ctx->stream));
    if (d.buf) PM_HIP_CHECK(hipFree(d.buf));
    d.buf = nullptr;
    d.cap = 0;
    const size_t cap = pm::align_up(bytes + bytes / 8, size_t(1) << 20);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d.buf), cap);
    if (e != hipSuccess) { pm::set_error("hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e)); return PM_E_NOMEM; }
    d.cap = cap;
    return PM_OK;
}

struct Carve {                    // 256-byte aligned bump carving of a Dev buffer
    char* base;
    size_t off = 0;
    explicit Carve(char* b) : base(b) {}
    template <typename T>
    T* take(size_t count)
    {
        off = pm::align_up(off, 256);
        T* p = reinterpret_cast<T*>(base + off);
        off += sizeof(T) * count;
        return p;
    }
};

}  // namespace

struct pm_mgpu {
    int n = 0;
    Dev* dev = nullptr;
};

extern "C" int pm_mgpu_create(int n_dev, const int* devices, pm_mgpu** out)
{
    PM_REQUIRE(out != nullptr, PM_E_INVALID, "out is null");
    *out = nullptr;
    int have = 0;
    PM_HIP_CHECK(hipGetDeviceCount(&have));
    PM_REQUIRE(n_dev >= 1 && n_dev <= have && n_dev <= PM_MAX_PARTS, PM_E_INVALID, "need 1 <= n_dev <= visible HIP devices (<= 64)");
    int rc = rccl_load(g_rccl);
    if (rc != PM_OK) return rc;
    pm_mgpu* mg = new (std::nothrow) pm_mgpu;
    PM_REQUIRE(mg != nullptr, PM_E_NOMEM, "out of host memory");
    mg->n = n_dev;
    mg->dev = new (std::nothrow) Dev[n_dev];
    if (!mg->dev) { delete mg; pm::set_error("out of host memory"); return PM_E_NOMEM; }
    std::vector<int> list(n_dev);
    for (int i = 0; i < n_dev; ++i) {
        list[i] = devices ? devices[i] : i;
        mg->dev[i].device = list[i];
        rc = pm_ctx_create(list[i], &mg->dev[i].ctx);
        if (rc != PM_OK) { (void)pm_mgpu_destroy(mg); return rc; }
    }
    std::vector<ncclComm_t> comms(n_dev, nullptr);
    // RCCL prints a version banner on fd 1 when its first communicator is made; the drop-in's stdout is the
    // reference's surface (main.cpp:58-59, :73-76, :119, :123), so fd 1 points at stderr for the duration of the call.
    fflush(stdout);
    const int saved_out = dup(1);
    if (saved_out >= 0) (void)dup2(2, 1);
    ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_dev, list.data());
    fflush(stdout);
    if (saved_out >= 0) { (void)dup2(saved_out, 1); (void)close(saved_out); }
    if (r != ncclSuccess) {
        pm::set_error("ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
        (void)pm_mgpu_destroy(mg);
        return PM_E_HIP;
    }
    for (int i = 0; i < n_dev; ++i) mg->dev[i].comm = comms[i];
    *out = mg;
    return PM_OK;
}

extern "C" int pm_mgpu_destroy(pm_mgpu* mg)
{
    if (!mg) return PM_OK;
    for (int i = 0; i < mg->n; ++i) {
        Dev& d = mg->dev[i];
        (void)hipSetDevice(d.device);
        if (d.ctx) (void)hipStreamSynchronize(d.ctx->stream);
        if (d.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d.comm);
        if (d.buf) (void)hipFree(d.buf);
        if (d.ctx) (void)pm_ctx_destroy(d.ctx);
    }
    delete[] mg->dev;
    delete mg;
    return PM_OK;
}

extern "C" int pm_mgpu_size(const pm_mgpu* mg) { return mg ? mg->n : 0; }

extern "C" pm_ctx* pm_mgpu_ctx(pm_mgpu* mg, int i) { return (mg && i >= 0 && i < mg->n) ? mg->dev[i].ctx : nullptr; }

namespace {

// hypothesis range of device g: contiguous, disjoint, union = [hb, he)
void hyp_range(const pm_ransac_params* p, int g, int G, pm_ransac_params& q)
{
    q = *p;
    const int64_t H = p->hyp_end - p->hyp_begin;
    q.hyp_begin = p->hyp_begin + H * g / G;
    q.hyp_end = p->hyp_begin + H * (g + 1) / G;
}

int gather_in_place(pm_mgpu* mg, std::vector<char*>& bufs, size_t bytes_per_rank)
{
    PM_NCCL_CHECK(g_rccl.GroupStart());
    for (int g = 0; g < mg->n; ++g) {
        Dev& d = mg->dev[g];
        ncclResult_t r = g_rccl.AllGather(bufs[g] + static_cast<size_t>(g) * bytes_per_rank, bufs[g], bytes_per_rank, ncclChar, d.comm,
                                          d.ctx->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            pm::set_error("ncclAllGather failed: %s", g_rccl.GetErrorString(r));
            return PM_E_HIP;
        }
    }
    PM_NCCL_CHECK(g_rccl.GroupEnd());
    return PM_OK;
}

int sync_all(pm_mgpu* mg)
{
    int rc = PM_OK;
    for (int g = 0; g < mg->n; ++g) {
        if (hipSetDevice(mg->dev[g].device) != hipSuccess || hipStreamSynchronize(mg->dev[g].ctx->stream) != hipSuccess) {
            if (rc == PM_OK) pm::set_error("stream synchronisation failed on device %d", mg->dev[g].device);
            rc = PM_E_HIP;
        }
    }
    return rc;
}

}  // namespace

// ---- main.cpp:95-98 over the devices: correspondences replicated, hypothesis ids sharded ---------------------------
extern "C" int pm_mgpu_ransac_fundamental(pm_mgpu* mg, const float* xy1, const float* xy2, int n, const pm_ransac_params* p,
                                          double F[9], uint8_t* mask, int* n_inliers, uint64_t* best_key)
{
    if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (mask && n > 0) memset(mask, 0, static_cast<size_t>(n));
    if (n_inliers) *n_inliers = 0;
    if (best_key) *best_key = 0;
    PM_REQUIRE(mg != nullptr && p != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(n >= 0 && (n == 0 || (xy1 && xy2)), PM_E_INVALID, "bad point arrays");
    PM_REQUIRE(p->hyp_begin >= 0 && p->hyp_end >= p->hyp_begin && p->hyp_end <= 0x100000000LL, PM_E_INVALID,
               "hypothesis ids must satisfy 0 <= begin <= end <= 2^32");
    PM_REQUIRE(p->hyp_end - p->hyp_begin >= mg->n, PM_E_INVALID, "fewer hypotheses than devices");
    if (n < 8) { pm::set_error("need at least 8 correspondences, got %d", n); return PM_E_TOO_FEW; }
    const int G = mg->n;
    const size_t xyb = sizeof(float) * 2 * static_cast<size_t>(n);
    struct Out { uint64_t key; double F[9]; int32_t ninl; int32_t ntot; };
    std::vector<char*> recs(G);
    std::vector<uint8_t*> dmask(G);
    std::vector<Out*> dout(G);
    for (int g = 0; g < G; ++g) {
        Dev& d = mg->dev[g];
        int rc = dev_reserve(d, 2 * pm::align_up(xyb, 256) + pm::align_up(sizeof(pm_ransac_record) * G, 256) +
                                    pm::align_up(static_cast<size_t>(n), 256) + 1024);
        if (rc != PM_OK) return rc;
        PM_HIP_CHECK(hipSetDevice(d.device));
        Carve c(d.buf);
        float* dxy1 = c.take<float>(2 * static_cast<size_t>(n));
        float* dxy2 = c.take<float>(2 * static_cast<size_t>(n));
        recs[g] = reinterpret_cast<char*>(c.take<pm_ransac_record>(G));
        dmask[g] = c.take<uint8_t>(static_cast<size_t>(n));
        dout[g] = c.take<Out>(1);
        PM_HIP_CHECK(hipMemcpyAsync(dxy1, xy1, xyb, hipMemcpyHostToDevice, d.ctx->stream));
        PM_HIP_CHECK(hipMemcpyAsync(dxy2, xy2, xyb, hipMemcpyHostToDevice, d.ctx->stream));
        pm_ransac_params q;
        hyp_range(p, g, G, q);
        const pm_points_view v{dxy1, dxy2, nullptr, 1, n, 0, 1, 0};
        rc = pm_ransac_shard_parts_dev(d.ctx, &v, &q, reinterpret_cast<pm_ransac_record*>(recs[g]) + g);
        if (rc != PM_OK) { (void)sync_all(mg); return rc; }
    }
    int rc = gather_in_place(mg, recs, sizeof(pm_ransac_record));          // the one exchange: 80 bytes per device
    if (rc != PM_OK) { (void)sync_all(mg); return rc; }
    for (int g = 0; g < G; ++g) {
        Dev& d = mg->dev[g];
        PM_HIP_CHECK(hipSetDevice(d.device));
        Carve c(d.buf);
        float* dxy1 = c.take<float>(2 * static_cast<size_t>(n));
        float* dxy2 = c.take<float>(2 * static_cast<size_t>(n));
        const pm_points_view v{dxy1, dxy2, nullptr, 1, n, 0, 1, 0};
        rc = pm_ransac_finish_parts_dev(d.ctx, &v, p, reinterpret_cast<pm_ransac_record*>(recs[g]), G, &dout[g]->key, dout[g]->F,
                                        dmask[g], n, &dout[g]->ninl, &dout[g]->ntot);
        if (rc != PM_OK) { (void)sync_all(mg); return rc; }
    }
    // every device holds the same answer; device 0's is returned
    Out h{};
    std::vector<uint8_t> hmask(static_cast<size_t>(n));
    PM_HIP_CHECK(hipSetDevice(mg->dev[0].device));
    PM_HIP_CHECK(hipMemcpyAsync(&h, dout[0], sizeof(Out), hipMemcpyDeviceToHost, mg->dev[0].ctx->stream));
    PM_HIP_CHECK(hipMemcpyAsync(hmask.data(), dmask[0], static_cast<size_t>(n), hipMemcpyDeviceToHost, mg->dev[0].ctx->stream));
    rc = sync_all(mg);
    if (rc != PM_OK) return rc;
    if (best_key) *best_key = h.key;
    if (h.key == 0) { pm::set_error("no valid model (all hypotheses degenerate)"); return PM_E_NO_MODEL; }
    if (F) memcpy(F, h.F, sizeof(h.F));
    if (mask) memcpy(mask, hmask.data(), static_cast<size_t>(n));
    if (n_inliers) *n_inliers = h.ninl;
    return PM_OK;
}

// ---- main.cpp:46 -> :49-69 (ratio form) -> :89-91 -> :95-98 over the devices (BASELINE config C4) -------------------
extern "C" int pm_mgpu_match_ransac(pm_mgpu* mg, const void* desc1, int n1, const void* desc2, int n2, int dim, int binary,
                                    const float* kp1_xy, const float* kp2_xy, float ratio, int knn_flags,
                                    const pm_ransac_params* p, pm_match* good, int* n_good, double F[9], uint8_t* mask,
                                    int* n_inliers, uint64_t* best_key)
{
    if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (n_good) *n_good = 0;
    if (n_inliers) *n_inliers = 0;
    if (best_key) *best_key = 0;
    PM_REQUIRE(mg != nullptr && p != nullptr && n_good != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(n1 >= 1 && n2 >= 1 && dim >= 1 && desc1 && desc2 && kp1_xy && kp2_xy, PM_E_INVALID, "bad descriptor / keypoint arrays");
    PM_REQUIRE(!binary || dim % 4 == 0, PM_E_INVALID, "binary descriptors: bytes per row must be a multiple of 4");
    PM_REQUIRE(p->hyp_begin >= 0 && p->hyp_end >= p->hyp_begin && p->hyp_end <= 0x100000000LL, PM_E_INVALID,
               "hypothesis ids must satisfy 0 <= begin <= end <= 2^32");
    PM_REQUIRE(p->hyp_end - p->hyp_begin >= mg->n, PM_E_INVALID, "fewer hypotheses than devices");
    const int G = mg->n;
    const int cap = (n1 + G - 1) / G;                          // query rows per device (the last block may be short)
    const size_t esz = binary ? 1 : sizeof(float);
    const size_t row = static_cast<size_t>(dim) * esz;
    // survivor block of one device: [count + pad (16 B) | xy1 cap x 8 B | xy2 cap x 8 B | records cap x 16 B]
    const size_t blk = 16 + static_cast<size_t>(cap) * (8 + 8 + 16);
    const size_t blk_al = pm::align_up(blk, 16);
    struct Out { uint64_t key; double F[9]; int32_t ninl; int32_t ntot; };
    std::vector<char*> gblk(G), recs(G);
    std::vector<uint8_t*> dmask(G);
    std::vector<Out*> dout(G);
    const int mask_len = G * cap;
    for (int g = 0; g < G; ++g) {
        Dev& d = mg->dev[g];
        const int r0 = g * cap < n1 ? g * cap : n1, r1 = (g + 1) * cap < n1 ? (g + 1) * cap : n1;
        const int rows = r1 - r0;
        size_t need = pm::align_up(static_cast<size_t>(cap) * row, 256) + pm::align_up(static_cast<size_t>(n2) * row, 256) +
                      pm::align_up(sizeof(float) * 2 * cap, 256) + pm::align_up(sizeof(float) * 2 * static_cast<size_t>(n2), 256) +
                      pm::align_up(sizeof(pm_match) * 2 * cap, 256) + pm::align_up(blk_al * G, 256) +
                      pm::align_up(sizeof(pm_ransac_record) * G, 256) + pm::align_up(static_cast<size_t>(mask_len), 256) + 2048;
        int rc = dev_reserve(d, need);
        if (rc != PM_OK) return rc;
        PM_HIP_CHECK(hipSetDevice(d.device));
        hipStream_t s = d.ctx->stream;
        Carve c(d.buf);
        char* dq = c.take<char>(static_cast<size_t>(cap) * row);
        char* dt = c.take<char>(static_cast<size_t>(n2) * row);
        float* dkp1 = c.take<float>(2 * static_cast<size_t>(cap));
        float* dkp2 = c.take<float>(2 * static_cast<size_t>(n2));
        pm_match* dknn = c.take<pm_match>(2 * static_cast<size_t>(cap));
        gblk[g] = c.take<char>(blk_al * G);
        recs[g] = reinterpret_cast<char*>(c.take<pm_ransac_record>(G));
        dmask[g] = c.take<uint8_t>(static_cast<size_t>(mask_len));
        dout[g] = c.take<Out>(1);
        char* mine = gblk[g] + static_cast<size_t>(g) * blk_al;
        int32_t* dcount = reinterpret_cast<int32_t*>(mine);
        float* dxy1 = reinterpret_cast<float*>(mine + 16);
        float* dxy2 = dxy1 + 2 * static_cast<size_t>(cap);
        pm_match* dgood = reinterpret_cast<pm_match*>(dxy2 + 2 * static_cast<size_t>(cap));
        PM_HIP_CHECK(hipMemcpyAsync(dt, desc2, static_cast<size_t>(n2) * row, hipMemcpyHostToDevice, s));
        PM_HIP_CHECK(hipMemcpyAsync(dkp2, kp2_xy, sizeof(float) * 2 * static_cast<size_t>(n2), hipMemcpyHostToDevice, s));
        if (rows > 0) {
            PM_HIP_CHECK(hipMemcpyAsync(dq, static_cast<const char*>(desc1) + static_cast<size_t>(r0) * row,
                                        static_cast<size_t>(rows) * row, hipMemcpyHostToDevice, s));
            PM_HIP_CHECK(hipMemcpyAsync(dkp1, kp1_xy + 2 * static_cast<size_t>(r0), sizeof(float) * 2 * static_cast<size_t>(rows),
                                        hipMemcpyHostToDevice, s));
            if (binary)
                rc = pm_bf_knn_hamming_u8_dev(d.ctx, reinterpret_cast<const uint8_t*>(dq), rows, reinterpret_cast<const uint8_t*>(dt),
                                              n2, dim, 2, dknn);
        }
        if (rc == PM_OK && !binary)
            rc = pm_bf_knn_l2_ratio_dev(d.ctx, reinterpret_cast<const float*>(dq), rows, reinterpret_cast<const float*>(dt), n2, dim,
                                        knn_flags, ratio, dkp1, dkp2, dknn, dgood, dxy1, dxy2, dcount);
        else if (rc == PM_OK)
            rc = pm_filter_ratio_gather_dev(d.ctx, dknn, rows, 2, ratio, dkp1, dkp2, dgood, dxy1, dxy2, dcount);
        if (rc != PM_OK) { (void)sync_all(mg); return rc; }
    }
    int rc = gather_in_place(mg, gblk, blk_al);                 // exchange 1: the survivor blocks
    if (rc != PM_OK) { (void)sync_all(mg); return rc; }
    auto view_of = [&](int g) {
        pm_points_view v{};
        v.xy1 = reinterpret_cast<const float*>(gblk[g] + 16);
        v.xy2 = v.xy1 + 2 * static_cast<size_t>(cap);
        v.counts = reinterpret_cast<const int32_t*>(gblk[g]);
        v.parts = G;
        v.cap = cap;
        v.pitch_xy = static_cast<int64_t>(blk_al / sizeof(float));
        v.pitch_cnt = static_cast<int32_t>(blk_al / sizeof(int32_t));
        return v;
    };
    for (int g = 0; g < G; ++g) {
        Dev& d = mg->dev[g];
        PM_HIP_CHECK(hipSetDevice(d.device));
        pm_ransac_params q;
        hyp_range(p, g, G, q);
        const pm_points_view v = view_of(g);
        rc = pm_ransac_shard_parts_dev(d.ctx, &v, &q, reinterpret_cast<pm_ransac_record*>(recs[g]) + g);
        if (rc != PM_OK) { (void)sync_all(mg); return rc; }
    }
    rc = gather_in_place(mg, recs, sizeof(pm_ransac_record));  // exchange 2: 80 bytes per device
    if (rc != PM_OK) { (void)sync_all(mg); return rc; }
    for (int g = 0; g < G; ++g) {
        Dev& d = mg->dev[g];
        PM_HIP_CHECK(hipSetDevice(d.device));
        const pm_points_view v = view_of(g);
        rc = pm_ransac_finish_parts_dev(d.ctx, &v, p, reinterpret_cast<pm_ransac_record*>(recs[g]), G, &dout[g]->key, dout[g]->F,
                                        dmask[g], mask_len, &dout[g]->ninl, &dout[g]->ntot);
        if (rc != PM_OK) { (void)sync_all(mg); return rc; }
    }
    // device 0's copy of everything goes back to the host
    Out h{};
    std::vector<char> hblk(blk_al * G);
    std::vector<uint8_t> hmask(static_cast<size_t>(mask_len));
    PM_HIP_CHECK(hipSetDevice(mg->dev[0].device));
    hipStream_t s0 = mg->dev[0].ctx->stream;
    PM_HIP_CHECK(hipMemcpyAsync(&h, dout[0], sizeof(Out), hipMemcpyDeviceToHost, s0));
    PM_HIP_CHECK(hipMemcpyAsync(hblk.data(), gblk[0], blk_al * G, hipMemcpyDeviceToHost, s0));
    PM_HIP_CHECK(hipMemcpyAsync(hmask.data(), dmask[0], static_cast<size_t>(mask_len), hipMemcpyDeviceToHost, s0));
    rc = sync_all(mg);
    if (rc != PM_OK) return rc;
    int total = 0;
    for (int g = 0; g < G; ++g) {
        const char* b = hblk.data() + static_cast<size_t>(g) * blk_al;
        int c = *reinterpret_cast<const int32_t*>(b);
        c = c < 0 ? 0 : (c > cap ? cap : c);
        const pm_match* rec = reinterpret_cast<const pm_match*>(b + 16 + static_cast<size_t>(cap) * 16);
        for (int i = 0; i < c; ++i) {
            pm_match m = rec[i];
            m.queryIdx += g * cap;                             // block-local row -> row of desc1
            if (good) good[total + i] = m;
        }
        total += c;
    }
    *n_good = total;
    if (best_key) *best_key = h.key;
    if (total < 8) { pm::set_error("need at least 8 correspondences, got %d", total); return PM_E_TOO_FEW; }
    if (h.key == 0) { pm::set_error("no valid model (all hypotheses degenerate)"); return PM_E_NO_MODEL; }
    if (F) memcpy(F, h.F, sizeof(h.F));
    if (mask) memcpy(mask, hmask.data(), static_cast<size_t>(total));
    if (n_inliers) *n_inliers = h.ninl;
    return PM_OK;
}
