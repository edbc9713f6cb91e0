// mgpu.cpp — the path over the GPUs of one node behind the C ABI (SURVEY.md 8b `pm_ransac_reduce`, 8e).  Slot in the
// reference: main.cpp:46 (the query rows of the matcher shard) and main.cpp:95-98 (the hypothesis ids of the robust
// estimator shard); the batch form runs the whole of main() once per image pair (main.cpp:9-147), pair p on device p mod N.
//
//   matcher      query rows cut into n_dev contiguous blocks, train set replicated AND RESIDENT (pm_mgpu_set_train[_dev]:
//                uploaded once, not once per pair); each device matches and filters its block straight into ITS slot of
//                the gathered buffer, then ncclAllGather #1 (in place) hands every device every block:
//                [count | xy1 | xy2 | match records].
//   RANSAC       device g samples/solves/scores ids [H*g/G, H*(g+1)/G) over ALL gathered correspondences (read through
//                a pm_points_view: no concatenation pass), leaves the shard's 80-byte (key, F) record in its slot;
//                ncclAllGather #2 (in place) = the arg-max all-reduce of SURVEY 8e carried with its payload; every
//                device then picks the winner and writes the inlier mask (pm_ransac_finish_parts_dev).  Nobody
//                re-solves, nothing is broadcast.
//
// Round 3: the step is throughput-shaped.  One HOST THREAD PER DEVICE enqueues that device's launches and collectives
// (a single thread walking 8 devices x ~8 launches cannot feed them: ~250 us of launch calls per pair against ~100 us of
// GPU work), and every device has up to PM_MGPU_MAX_LANES lanes — context + stream + RCCL communicator + buffers each —
// with pair j on lane j mod L, so pair j+1's matcher runs while pair j waits in its two all-gathers.  pm_mgpu_submit_dev
// takes device pointers and returns a ticket at once; pm_mgpu_collect blocks for one ticket.  Host inputs
// (pm_mgpu_match_ransac, pm_mgpu_set_train) are staged through ONE pinned copy that all devices' H2D engines read
// concurrently (round 2 issued eight serialised pageable copies of the replicated train set).
// RCCL is bound at run time (dlopen "librccl.so.1": the copy already in the process when there is one, e.g. under
// PyTorch-ROCm, else /opt/rocm's), so single-GPU users of libpm_hip.so never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <exception>
#include <thread>
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

constexpr int MAX_LANES = 4;
constexpr int TICKET_RING = 64;

struct Out { uint64_t key; double F[9]; int32_t ninl; int32_t ntot; };

// one (device, lane): its own stream (context), communicator, scratch
struct Lane {
    pm_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    char* buf = nullptr;          // grow-only device scratch: query block, keypoints, k-NN records, gathered blocks, records, mask, out
    size_t cap = 0;
    char* hbuf = nullptr;         // pinned (device 0 only): out + gathered blocks + mask of the lane's last pair
    size_t hcap = 0;
    hipEvent_t done = nullptr;
};

struct Job {
    int kind = 0;                 // 0 pair, 1 train upload, 2 batch, 3 quit, 4 exchange probe
    int ticket = 0, lane = 0;
    // pair
    const void* d_desc1 = nullptr;    // this device's query rows (device memory), or null: copy from h_desc1
    const float* d_kp1 = nullptr;
    const char* h_desc1 = nullptr;    // pinned staging (host-pointer form)
    const float* h_kp1 = nullptr;
    int rows = 0, cap = 0, r0 = 0;
    float ratio = 0.f;
    int flags = 0;
    pm_ransac_params p{};
    // train upload
    const char* h_desc2 = nullptr;
    const float* h_kp2 = nullptr;
    // batch
    pm_batch* batch = nullptr;
    const pm_pair_job* jobs = nullptr;
    int n_jobs = 0, stride = 1, first = 0, knn_flags = 0;
    pm_pair_result* results = nullptr;
    pm_match* good = nullptr;
    uint8_t* masks = nullptr;
    int max_n1 = 0;
    // probe
    int probe_bytes = 0, probe_reps = 0;
    double* probe_us = nullptr;
};

struct Dev {
    int device = 0;
    int index = 0;
    Lane lane[MAX_LANES];
    // resident train side
    char* dt = nullptr;
    float* dkp2 = nullptr;
    size_t t_cap = 0, kp2_cap = 0;
    bool t_borrowed = false;      // pm_mgpu_set_train_dev: the caller's buffers
    pm_batch* batch = nullptr;    // pm_mgpu_batch_run
    int batch_sig[4] = {0, 0, 0, 0};
    // worker
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> queue;
};

struct Ticket {
    std::atomic<int> done{0};     // workers that have ENQUEUED this ticket's work
    int rc[PM_MAX_PARTS];
    int lane = 0, cap = 0, n1 = 0, G = 0;
    bool open = false;
    bool pair = false;            // an image pair (holds its lane until collected); else train upload / probe / batch
};

int lane_reserve(Dev& d, Lane& L, size_t bytes)
{
    if (bytes <= L.cap) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(d.device));
    PM_HIP_CHECK(hipStreamSynchronize(L.ctx->stream));
    if (L.buf) PM_HIP_CHECK(hipFree(L.buf));
    L.buf = nullptr;
    L.cap = 0;
    const size_t cap = pm::align_up(bytes + bytes / 8, size_t(1) << 20);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&L.buf), cap);
    if (e != hipSuccess) { pm::set_error("hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e)); return PM_E_NOMEM; }
    L.cap = cap;
    return PM_OK;
}

int pinned_reserve(char*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap) return PM_OK;
    if (p) PM_HIP_CHECK(hipHostFree(p));
    p = nullptr;
    cap = 0;
    const size_t want = pm::align_up(bytes + bytes / 8, size_t(1) << 16);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault);
    if (e != hipSuccess) { pm::set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return PM_E_NOMEM; }
    cap = want;
    return PM_OK;
}

struct Carve {                    // 256-byte aligned bump carving of a buffer
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

// layout of a lane's device scratch for one pair (the same carving on every device and in pm_mgpu_collect)
struct PairLayout {
    size_t row, blk_al;
    int cap, G, mask_len;
    char *dq, *gblk, *recs;
    float* dkp1;
    pm_match* dknn;
    uint8_t* dmask;
    Out* dout;
    size_t bytes;
    PairLayout(char* base, int cap_, int G_, int dim, int binary) : cap(cap_), G(G_)
    {
        row = static_cast<size_t>(dim) * (binary ? 1 : sizeof(float));
        // survivor block of one device: [count + pad (16 B) | xy1 cap x 8 B | xy2 cap x 8 B | records cap x 16 B]
        blk_al = pm::align_up(16 + static_cast<size_t>(cap) * (8 + 8 + 16), 16);
        mask_len = G * cap;
        Carve c(base);
        dq = c.take<char>(static_cast<size_t>(cap) * row);
        dkp1 = c.take<float>(2 * static_cast<size_t>(cap));
        dknn = c.take<pm_match>(2 * static_cast<size_t>(cap));
        gblk = c.take<char>(blk_al * G);
        recs = reinterpret_cast<char*>(c.take<pm_ransac_record>(G));
        dmask = c.take<uint8_t>(static_cast<size_t>(mask_len));
        dout = c.take<Out>(1);
        bytes = c.off + 256;
    }
    pm_points_view view() const
    {
        pm_points_view v{};
        v.xy1 = reinterpret_cast<const float*>(gblk + 16);
        v.xy2 = v.xy1 + 2 * static_cast<size_t>(cap);
        v.counts = reinterpret_cast<const int32_t*>(gblk);
        v.parts = G;
        v.cap = cap;
        v.pitch_xy = static_cast<int64_t>(blk_al / sizeof(float));
        v.pitch_cnt = static_cast<int32_t>(blk_al / sizeof(int32_t));
        return v;
    }
};

}  // namespace

struct pm_mgpu {
    int n = 0;
    int lanes = 1;
    Dev* dev = nullptr;
    // resident train side
    int n2 = 0, dim = 0, binary = 0;
    bool have_train = false;
    char* hstage = nullptr;       // pinned staging of host inputs (train set; per-lane query blocks behind it)
    size_t hstage_cap = 0;
    char* hq[MAX_LANES] = {};     // pinned staging of a pair's host query side, one per lane
    size_t hq_cap[MAX_LANES] = {};
    Ticket tk[TICKET_RING];
    int next_ticket = 0;
    std::mutex tmu;
    std::condition_variable tcv;
};

namespace {

void ticket_mark(pm_mgpu* mg, int ticket, int g, int rc)
{
    Ticket& t = mg->tk[ticket % TICKET_RING];
    t.rc[g] = rc;
    {
        std::lock_guard<std::mutex> lk(mg->tmu);
        t.done.fetch_add(1);
    }
    mg->tcv.notify_all();
}

// hypothesis range of device g: contiguous, disjoint, union = [hb, he)
void hyp_range(const pm_ransac_params* p, int g, int G, pm_ransac_params& q)
{
    q = *p;
    const int64_t H = p->hyp_end - p->hyp_begin;
    q.hyp_begin = p->hyp_begin + H * g / G;
    q.hyp_end = p->hyp_begin + H * (g + 1) / G;
}

#define PM_W_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess && rc == PM_OK) {                                                  \
            ::pm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = PM_E_HIP;                                                                      \
        }                                                                                       \
    } while (0)
#define PM_W_NCCL(expr)                                                                         \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess && rc == PM_OK) {                                                 \
            ::pm::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            rc = PM_E_HIP;                                                                      \
        }                                                                                       \
    } while (0)

// One image pair on one device: everything this device contributes, enqueued on the lane's stream.  The two collectives
// are issued unconditionally (a device that failed earlier still takes part: its peers must not hang in the all-gather;
// the failure is reported through the ticket).
int run_pair(pm_mgpu* mg, Dev& d, const Job& j)
{
    Lane& L = d.lane[j.lane];
    const int G = mg->n, g = d.index;
    int rc = PM_OK;
    PM_W_HIP(hipSetDevice(d.device));
    hipStream_t s = L.ctx->stream;
    PairLayout lay(L.buf, j.cap, G, mg->dim, mg->binary);
    char* mine = lay.gblk + static_cast<size_t>(g) * lay.blk_al;
    int32_t* dcount = reinterpret_cast<int32_t*>(mine);
    float* dxy1 = reinterpret_cast<float*>(mine + 16);
    float* dxy2 = dxy1 + 2 * static_cast<size_t>(j.cap);
    pm_match* dgood = reinterpret_cast<pm_match*>(dxy2 + 2 * static_cast<size_t>(j.cap));
    const void* dq = j.d_desc1;
    const float* dkp1 = j.d_kp1;
    if (!dq && j.rows > 0) {      // host-pointer form: this device's slice of the pinned staging copy
        PM_W_HIP(hipMemcpyAsync(lay.dq, j.h_desc1 + static_cast<size_t>(j.r0) * lay.row, static_cast<size_t>(j.rows) * lay.row,
                                hipMemcpyHostToDevice, s));
        PM_W_HIP(hipMemcpyAsync(lay.dkp1, j.h_kp1 + 2 * static_cast<size_t>(j.r0), sizeof(float) * 2 * static_cast<size_t>(j.rows),
                                hipMemcpyHostToDevice, s));
        dq = lay.dq;
        dkp1 = lay.dkp1;
    }
    if (rc == PM_OK) {
        if (mg->binary) {
            if (j.rows > 0)
                rc = pm_bf_knn_hamming_u8_dev(L.ctx, static_cast<const uint8_t*>(dq), j.rows, reinterpret_cast<const uint8_t*>(d.dt),
                                              mg->n2, mg->dim, 2, lay.dknn);
            if (rc == PM_OK)
                rc = pm_filter_ratio_gather_dev(L.ctx, lay.dknn, j.rows, 2, j.ratio, dkp1, d.dkp2, dgood, dxy1, dxy2, dcount);
        } else {
            rc = pm_bf_knn_l2_ratio_dev(L.ctx, static_cast<const float*>(dq), j.rows, reinterpret_cast<const float*>(d.dt), mg->n2,
                                        mg->dim, j.flags, j.ratio, dkp1, d.dkp2, lay.dknn, dgood, dxy1, dxy2, dcount);
        }
    }
    // exchange 1: the survivor blocks (in place: this device's block already sits in its slot)
    PM_W_NCCL(g_rccl.AllGather(mine, lay.gblk, lay.blk_al, ncclChar, L.comm, s));
    const pm_points_view v = lay.view();
    pm_ransac_record* rec = reinterpret_cast<pm_ransac_record*>(lay.recs);
    if (rc == PM_OK) {
        pm_ransac_params q;
        hyp_range(&j.p, g, G, q);
        rc = pm_ransac_shard_parts_dev(L.ctx, &v, &q, rec + g);
    }
    // exchange 2: 80 bytes per device
    PM_W_NCCL(g_rccl.AllGather(rec + g, rec, sizeof(pm_ransac_record), ncclChar, L.comm, s));
    if (rc == PM_OK)
        rc = pm_ransac_finish_parts_dev(L.ctx, &v, &j.p, rec, G, &lay.dout->key, lay.dout->F, lay.dmask, lay.mask_len,
                                        &lay.dout->ninl, &lay.dout->ntot);
    if (g == 0 && rc == PM_OK) {      // device 0's copy of everything goes to the lane's pinned slot
        Carve h(L.hbuf);
        Out* ho = h.take<Out>(1);
        char* hb = h.take<char>(lay.blk_al * G);
        uint8_t* hm = h.take<uint8_t>(static_cast<size_t>(lay.mask_len));
        PM_W_HIP(hipMemcpyAsync(ho, lay.dout, sizeof(Out), hipMemcpyDeviceToHost, s));
        PM_W_HIP(hipMemcpyAsync(hb, lay.gblk, lay.blk_al * G, hipMemcpyDeviceToHost, s));
        PM_W_HIP(hipMemcpyAsync(hm, lay.dmask, static_cast<size_t>(lay.mask_len), hipMemcpyDeviceToHost, s));
    }
    PM_W_HIP(hipEventRecord(L.done, s));
    return rc;
}

int run_train_upload(pm_mgpu* mg, Dev& d, const Job& j)
{
    int rc = PM_OK;
    PM_W_HIP(hipSetDevice(d.device));
    hipStream_t s = d.lane[0].ctx->stream;
    const size_t row = static_cast<size_t>(mg->dim) * (mg->binary ? 1 : sizeof(float));
    PM_W_HIP(hipMemcpyAsync(d.dt, j.h_desc2, static_cast<size_t>(mg->n2) * row, hipMemcpyHostToDevice, s));
    PM_W_HIP(hipMemcpyAsync(d.dkp2, j.h_kp2, sizeof(float) * 2 * static_cast<size_t>(mg->n2), hipMemcpyHostToDevice, s));
    PM_W_HIP(hipStreamSynchronize(s));          // the staging copy is the caller's to re-use after pm_mgpu_set_train
    return rc;
}

// latency of one all-gather of `probe_bytes` per device, by itself: reps back-to-back collectives between two events
int run_probe(pm_mgpu* mg, Dev& d, const Job& j)
{
    Lane& L = d.lane[0];
    int rc = PM_OK;
    PM_W_HIP(hipSetDevice(d.device));
    hipStream_t s = L.ctx->stream;
    char* buf = L.buf;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    PM_W_HIP(hipEventCreate(&e0));
    PM_W_HIP(hipEventCreate(&e1));
    const size_t b = static_cast<size_t>(j.probe_bytes);
    for (int i = 0; i < 3; ++i) PM_W_NCCL(g_rccl.AllGather(buf + d.index * b, buf, b, ncclChar, L.comm, s));
    PM_W_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < j.probe_reps; ++i) PM_W_NCCL(g_rccl.AllGather(buf + d.index * b, buf, b, ncclChar, L.comm, s));
    PM_W_HIP(hipEventRecord(e1, s));
    PM_W_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    if (rc == PM_OK) PM_W_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (j.probe_us) j.probe_us[d.index] = rc == PM_OK ? static_cast<double>(ms) * 1e3 / j.probe_reps : -1.0;
    return rc;
}

int run_batch(pm_mgpu*, Dev& d, const Job& j)
{
    // this device's share of the batch: jobs first, first + stride, ... (pair p -> device p mod N), results scattered back
    std::vector<pm_pair_job> mine;
    std::vector<int> where;
    for (int p = j.first; p < j.n_jobs; p += j.stride) { mine.push_back(j.jobs[p]); where.push_back(p); }
    if (mine.empty()) return PM_OK;
    const int m = static_cast<int>(mine.size());
    std::vector<pm_pair_result> res(mine.size());
    std::vector<pm_match> good;
    std::vector<uint8_t> masks;
    if (j.good) good.resize(static_cast<size_t>(m) * j.max_n1);
    if (j.masks) masks.resize(static_cast<size_t>(m) * j.max_n1);
    const int rc = pm_batch_run(d.batch, mine.data(), m, j.ratio, j.knn_flags, &j.p, res.data(), j.good ? good.data() : nullptr,
                                j.masks ? masks.data() : nullptr);
    if (rc != PM_OK) return rc;
    for (int i = 0; i < m; ++i) {
        j.results[where[i]] = res[i];
        if (j.good) memcpy(j.good + static_cast<size_t>(where[i]) * j.max_n1, good.data() + static_cast<size_t>(i) * j.max_n1,
                           sizeof(pm_match) * static_cast<size_t>(res[i].n_good > 0 ? res[i].n_good : 0));
        if (j.masks) memcpy(j.masks + static_cast<size_t>(where[i]) * j.max_n1, masks.data() + static_cast<size_t>(i) * j.max_n1,
                            static_cast<size_t>(mine[i].n1));
    }
    return PM_OK;
}

void worker_main(pm_mgpu* mg, Dev* d)
{
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> lk(d->mu);
            d->cv.wait(lk, [&] { return !d->queue.empty(); });
            j = d->queue.front();
            d->queue.pop_front();
        }
        if (j.kind == 3) return;
        int rc = PM_OK;
        if (j.kind == 0) rc = run_pair(mg, *d, j);
        else if (j.kind == 1) rc = run_train_upload(mg, *d, j);
        else if (j.kind == 2) rc = run_batch(mg, *d, j);
        else if (j.kind == 4) rc = run_probe(mg, *d, j);
        ticket_mark(mg, j.ticket, d->index, rc);
    }
}

void post(Dev& d, const Job& j)
{
    {
        std::lock_guard<std::mutex> lk(d.mu);
        d.queue.push_back(j);
    }
    d.cv.notify_one();
}

int ticket_open(pm_mgpu* mg)
{
    const int t = mg->next_ticket++;
    Ticket& k = mg->tk[t % TICKET_RING];
    k.done.store(0);
    for (int g = 0; g < mg->n; ++g) k.rc[g] = PM_OK;
    k.open = true;
    k.pair = false;
    return t;
}

// wait until every worker has enqueued the ticket's work; first failure (if any)
int ticket_wait(pm_mgpu* mg, int ticket)
{
    Ticket& t = mg->tk[ticket % TICKET_RING];
    {
        std::unique_lock<std::mutex> lk(mg->tmu);
        mg->tcv.wait(lk, [&] { return t.done.load() >= mg->n; });
    }
    t.open = false;
    for (int g = 0; g < mg->n; ++g)
        if (t.rc[g] != PM_OK) {           // (the worker's own message is thread-local to it)
            pm::set_error("device %d failed while enqueueing its share: %s", mg->dev[g].device, pm_status_string(t.rc[g]));
            return t.rc[g];
        }
    return PM_OK;
}

// entry points that re-use lane 0's buffers or issue collectives from the calling thread need the workers idle
bool tickets_outstanding(const pm_mgpu* mg)
{
    for (int i = 0; i < TICKET_RING; ++i)
        if (mg->tk[i].open) return true;
    return false;
}

int sync_all(pm_mgpu* mg)
{
    int rc = PM_OK;
    for (int g = 0; g < mg->n; ++g)
        for (int l = 0; l < mg->lanes; ++l) {
            Lane& L = mg->dev[g].lane[l];
            if (!L.ctx) continue;
            if (hipSetDevice(mg->dev[g].device) != hipSuccess || hipStreamSynchronize(L.ctx->stream) != hipSuccess) {
                if (rc == PM_OK) pm::set_error("stream synchronisation failed on device %d", mg->dev[g].device);
                rc = PM_E_HIP;
            }
        }
    return rc;
}

// contexts, events and one communicator set for lane l
int lane_create(pm_mgpu* mg, int l)
{
    const int G = mg->n;
    std::vector<int> list(G);
    for (int g = 0; g < G; ++g) {
        Dev& d = mg->dev[g];
        list[g] = d.device;
        int rc = pm_ctx_create(d.device, &d.lane[l].ctx);
        if (rc != PM_OK) return rc;
        PM_HIP_CHECK(hipSetDevice(d.device));
        PM_HIP_CHECK(hipEventCreateWithFlags(&d.lane[l].done, hipEventDisableTiming));
    }
    std::vector<ncclComm_t> comms(G, nullptr);
    // RCCL prints a version banner on fd 1 when its first communicator is made; the drop-in's stdout is the
    // reference's surface (main.cpp:58-59, :73-76, :119, :123), so fd 1 points at stderr for the duration of the call.
    fflush(stdout);
    const int saved_out = dup(1);
    if (saved_out >= 0) (void)dup2(2, 1);
    ncclResult_t r = g_rccl.CommInitAll(comms.data(), G, list.data());
    fflush(stdout);
    if (saved_out >= 0) { (void)dup2(saved_out, 1); (void)close(saved_out); }
    if (r != ncclSuccess) {
        pm::set_error("ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
        return PM_E_HIP;
    }
    for (int g = 0; g < G; ++g) mg->dev[g].lane[l].comm = comms[g];
    return PM_OK;
}

}  // namespace

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
    for (int i = 0; i < n_dev; ++i) {
        mg->dev[i].device = devices ? devices[i] : i;
        mg->dev[i].index = i;
    }
    rc = lane_create(mg, 0);
    if (rc != PM_OK) { (void)pm_mgpu_destroy(mg); return rc; }
    try {                                                    // (the C ABI never throws: a host that cannot start a thread gets a status)
        for (int i = 0; i < n_dev; ++i) mg->dev[i].worker = std::thread(worker_main, mg, &mg->dev[i]);
    } catch (const std::exception& e) {
        pm::set_error("cannot start a worker thread: %s", e.what());
        (void)pm_mgpu_destroy(mg);                           // (joins the workers that did start)
        return PM_E_NOMEM;
    }
    *out = mg;
    return PM_OK;
}

extern "C" int pm_mgpu_set_lanes(pm_mgpu* mg, int n_lanes)
{
    PM_REQUIRE(mg != nullptr && n_lanes >= 1 && n_lanes <= MAX_LANES, PM_E_INVALID, "need 1 <= n_lanes <= 4");
    for (int l = mg->lanes; l < n_lanes; ++l) {
        const int rc = lane_create(mg, l);
        if (rc != PM_OK) return rc;
        mg->lanes = l + 1;
    }
    if (n_lanes < mg->lanes) {              // shrinking only stops the use of the upper lanes
        const int rc = sync_all(mg);
        if (rc != PM_OK) return rc;
        mg->lanes = n_lanes;
    }
    return PM_OK;
}

extern "C" int pm_mgpu_destroy(pm_mgpu* mg)
{
    if (!mg) return PM_OK;
    for (int i = 0; i < mg->n; ++i) {
        Dev& d = mg->dev[i];
        if (d.worker.joinable()) {
            Job q;
            q.kind = 3;
            post(d, q);
            d.worker.join();
        }
    }
    for (int i = 0; i < mg->n; ++i) {
        Dev& d = mg->dev[i];
        (void)hipSetDevice(d.device);
        for (int l = 0; l < MAX_LANES; ++l) {
            Lane& L = d.lane[l];
            if (L.ctx) (void)hipStreamSynchronize(L.ctx->stream);
            if (L.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(L.comm);
            if (L.buf) (void)hipFree(L.buf);
            if (L.hbuf) (void)hipHostFree(L.hbuf);
            if (L.done) (void)hipEventDestroy(L.done);
            if (L.ctx) (void)pm_ctx_destroy(L.ctx);
        }
        if (!d.t_borrowed) {
            if (d.dt) (void)hipFree(d.dt);
            if (d.dkp2) (void)hipFree(d.dkp2);
        }
        if (d.batch) (void)pm_batch_destroy(d.batch);
    }
    if (mg->hstage) (void)hipHostFree(mg->hstage);
    for (int l = 0; l < MAX_LANES; ++l)
        if (mg->hq[l]) (void)hipHostFree(mg->hq[l]);
    delete[] mg->dev;
    delete mg;
    return PM_OK;
}

extern "C" int pm_mgpu_size(const pm_mgpu* mg) { return mg ? mg->n : 0; }

extern "C" pm_ctx* pm_mgpu_ctx(pm_mgpu* mg, int i) { return (mg && i >= 0 && i < mg->n) ? mg->dev[i].lane[0].ctx : nullptr; }

extern "C" pm_ctx* pm_mgpu_lane_ctx(pm_mgpu* mg, int i, int lane)
{
    return (mg && i >= 0 && i < mg->n && lane >= 0 && lane < mg->lanes) ? mg->dev[i].lane[lane].ctx : nullptr;
}

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
    PM_REQUIRE(!tickets_outstanding(mg), PM_E_INVALID, "collect the submitted pairs first");
    if (n < 8) { pm::set_error("need at least 8 correspondences, got %d", n); return PM_E_TOO_FEW; }
    const int G = mg->n;
    const size_t xyb = sizeof(float) * 2 * static_cast<size_t>(n);
    std::vector<char*> recs(G);
    std::vector<uint8_t*> dmask(G);
    std::vector<Out*> dout(G);
    std::vector<float*> dx1(G), dx2(G);
    const size_t need = 2 * pm::align_up(xyb, 256) + pm::align_up(sizeof(pm_ransac_record) * G, 256) +
                        pm::align_up(static_cast<size_t>(n), 256) + 1024;
    // one exit for every failure: nothing may stay in flight on the caller's arrays (copies come from pageable memory)
    int rc = sync_all(mg);
    for (int g = 0; g < G && rc == PM_OK; ++g) rc = lane_reserve(mg->dev[g], mg->dev[g].lane[0], need);
    for (int g = 0; g < G && rc == PM_OK; ++g) {
        Dev& d = mg->dev[g];
        Lane& L = d.lane[0];
        PM_W_HIP(hipSetDevice(d.device));
        Carve c(L.buf);
        dx1[g] = c.take<float>(2 * static_cast<size_t>(n));
        dx2[g] = c.take<float>(2 * static_cast<size_t>(n));
        recs[g] = reinterpret_cast<char*>(c.take<pm_ransac_record>(G));
        dmask[g] = c.take<uint8_t>(static_cast<size_t>(n));
        dout[g] = c.take<Out>(1);
        PM_W_HIP(hipMemcpyAsync(dx1[g], xy1, xyb, hipMemcpyHostToDevice, L.ctx->stream));
        PM_W_HIP(hipMemcpyAsync(dx2[g], xy2, xyb, hipMemcpyHostToDevice, L.ctx->stream));
        if (rc != PM_OK) break;
        pm_ransac_params q;
        hyp_range(p, g, G, q);
        const pm_points_view v{dx1[g], dx2[g], nullptr, 1, n, 0, 1, 0};
        rc = pm_ransac_shard_parts_dev(L.ctx, &v, &q, reinterpret_cast<pm_ransac_record*>(recs[g]) + g);
    }
    if (rc == PM_OK) {                                       // the one exchange: 80 bytes per device
        PM_W_NCCL(g_rccl.GroupStart());
        for (int g = 0; g < G && rc == PM_OK; ++g)
            PM_W_NCCL(g_rccl.AllGather(recs[g] + static_cast<size_t>(g) * sizeof(pm_ransac_record), recs[g], sizeof(pm_ransac_record),
                                       ncclChar, mg->dev[g].lane[0].comm, mg->dev[g].lane[0].ctx->stream));
        PM_W_NCCL(g_rccl.GroupEnd());
    }
    for (int g = 0; g < G && rc == PM_OK; ++g) {
        Dev& d = mg->dev[g];
        PM_W_HIP(hipSetDevice(d.device));
        const pm_points_view v{dx1[g], dx2[g], nullptr, 1, n, 0, 1, 0};
        if (rc == PM_OK)
            rc = pm_ransac_finish_parts_dev(d.lane[0].ctx, &v, p, reinterpret_cast<pm_ransac_record*>(recs[g]), G, &dout[g]->key,
                                            dout[g]->F, dmask[g], n, &dout[g]->ninl, &dout[g]->ntot);
    }
    // every device holds the same answer; device 0's is returned
    Out h{};
    std::vector<uint8_t> hmask(static_cast<size_t>(n));
    if (rc == PM_OK) {
        PM_W_HIP(hipSetDevice(mg->dev[0].device));
        PM_W_HIP(hipMemcpyAsync(&h, dout[0], sizeof(Out), hipMemcpyDeviceToHost, mg->dev[0].lane[0].ctx->stream));
        PM_W_HIP(hipMemcpyAsync(hmask.data(), dmask[0], static_cast<size_t>(n), hipMemcpyDeviceToHost, mg->dev[0].lane[0].ctx->stream));
    }
    const int rs = sync_all(mg);
    if (rc == PM_OK) rc = rs;
    if (rc != PM_OK) return rc;
    if (best_key) *best_key = h.key;
    if (h.key == 0) { pm::set_error("no valid model (all hypotheses degenerate)"); return PM_E_NO_MODEL; }
    if (F) memcpy(F, h.F, sizeof(h.F));
    if (mask) memcpy(mask, hmask.data(), static_cast<size_t>(n));
    if (n_inliers) *n_inliers = h.ninl;
    return PM_OK;
}

// ---- the resident train side ---------------------------------------------------------------------------------------
namespace {
int train_shape(pm_mgpu* mg, int n2, int dim, int binary)
{
    PM_REQUIRE(mg != nullptr && n2 >= 1 && dim >= 1, PM_E_INVALID, "bad train set");
    PM_REQUIRE(!binary || dim % 4 == 0, PM_E_INVALID, "binary descriptors: bytes per row must be a multiple of 4");
    PM_REQUIRE(!tickets_outstanding(mg), PM_E_INVALID, "collect the submitted pairs first");
    const int rc = sync_all(mg);             // pairs in flight still read the old train set
    if (rc != PM_OK) return rc;
    mg->n2 = n2; mg->dim = dim; mg->binary = binary ? 1 : 0;
    mg->have_train = false;
    return PM_OK;
}
}  // namespace

extern "C" int pm_mgpu_set_train_dev(pm_mgpu* mg, const void* const* d_desc2, int n2, int dim, int binary,
                                     const float* const* d_kp2_xy)
{
    PM_REQUIRE(mg != nullptr && d_desc2 != nullptr && d_kp2_xy != nullptr, PM_E_INVALID, "null argument");
    int rc = train_shape(mg, n2, dim, binary);
    if (rc != PM_OK) return rc;
    for (int g = 0; g < mg->n; ++g) {
        Dev& d = mg->dev[g];
        PM_REQUIRE(d_desc2[g] && d_kp2_xy[g], PM_E_INVALID, "null per-device train pointer");
        if (!d.t_borrowed) {
            PM_HIP_CHECK(hipSetDevice(d.device));
            if (d.dt) PM_HIP_CHECK(hipFree(d.dt));
            if (d.dkp2) PM_HIP_CHECK(hipFree(d.dkp2));
            d.t_cap = d.kp2_cap = 0;
        }
        d.dt = static_cast<char*>(const_cast<void*>(d_desc2[g]));
        d.dkp2 = const_cast<float*>(d_kp2_xy[g]);
        d.t_borrowed = true;
    }
    mg->have_train = true;
    return PM_OK;
}

extern "C" int pm_mgpu_set_train(pm_mgpu* mg, const void* desc2, int n2, int dim, int binary, const float* kp2_xy)
{
    PM_REQUIRE(mg != nullptr && desc2 != nullptr && kp2_xy != nullptr, PM_E_INVALID, "null argument");
    int rc = train_shape(mg, n2, dim, binary);
    if (rc != PM_OK) return rc;
    const size_t row = static_cast<size_t>(dim) * (binary ? 1 : sizeof(float));
    const size_t tb = static_cast<size_t>(n2) * row, kb = sizeof(float) * 2 * static_cast<size_t>(n2);
    rc = pinned_reserve(mg->hstage, mg->hstage_cap, pm::align_up(tb, 256) + kb);
    if (rc != PM_OK) return rc;
    memcpy(mg->hstage, desc2, tb);                            // ONE pinned copy; every device's copy engine reads it
    memcpy(mg->hstage + pm::align_up(tb, 256), kp2_xy, kb);
    for (int g = 0; g < mg->n; ++g) {
        Dev& d = mg->dev[g];
        PM_HIP_CHECK(hipSetDevice(d.device));
        if (d.t_borrowed) { d.dt = nullptr; d.dkp2 = nullptr; d.t_cap = d.kp2_cap = 0; d.t_borrowed = false; }
        if (tb > d.t_cap) {
            if (d.dt) PM_HIP_CHECK(hipFree(d.dt));
            d.dt = nullptr; d.t_cap = 0;
            PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d.dt), tb));
            d.t_cap = tb;
        }
        if (kb > d.kp2_cap) {
            if (d.dkp2) PM_HIP_CHECK(hipFree(d.dkp2));
            d.dkp2 = nullptr; d.kp2_cap = 0;
            PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d.dkp2), kb));
            d.kp2_cap = kb;
        }
    }
    const int t = ticket_open(mg);
    for (int g = 0; g < mg->n; ++g) {
        Job j;
        j.kind = 1; j.ticket = t;
        j.h_desc2 = mg->hstage;
        j.h_kp2 = reinterpret_cast<const float*>(mg->hstage + pm::align_up(tb, 256));
        post(mg->dev[g], j);
    }
    rc = ticket_wait(mg, t);
    if (rc != PM_OK) return rc;
    mg->have_train = true;
    return PM_OK;
}

// ---- one image pair, streamed: main.cpp:46 -> :49-69 (ratio form) -> :89-91 -> :95-98 over the devices ----------------
namespace {
int submit_common(pm_mgpu* mg, const void* const* d_desc1, const float* const* d_kp1, const char* h_desc1, const float* h_kp1,
                  const int32_t* rows, int cap, float ratio, int knn_flags, const pm_ransac_params* p, int* ticket)
{
    const int G = mg->n;
    PM_REQUIRE(mg->have_train, PM_E_INVALID, "no train set: call pm_mgpu_set_train[_dev] first");
    PM_REQUIRE(p->hyp_begin >= 0 && p->hyp_end >= p->hyp_begin && p->hyp_end <= 0x100000000LL, PM_E_INVALID,
               "hypothesis ids must satisfy 0 <= begin <= end <= 2^32");
    PM_REQUIRE(p->hyp_end - p->hyp_begin >= G, PM_E_INVALID, "fewer hypotheses than devices");
    PM_REQUIRE(static_cast<long long>(G) * cap <= 0x7FFFFFFFLL, PM_E_INVALID, "too many query rows");
    const int t = mg->next_ticket;
    const int lane = t % mg->lanes;
    // a lane holds ONE pair's results (device blocks + device 0's pinned copy) until they are collected: at most `lanes`
    // pairs are outstanding, and the pair that last used this lane must have been collected
    for (int i = 0; i < TICKET_RING; ++i)
        PM_REQUIRE(!(mg->tk[i].open && mg->tk[i].pair && mg->tk[i].lane == lane), PM_E_INVALID,
                   "this lane still holds an uncollected pair: collect it first (at most pm_mgpu_set_lanes pairs in flight)");
    PM_REQUIRE(!mg->tk[t % TICKET_RING].open, PM_E_INVALID, "too many tickets in flight");
    // everything that can fail for lack of memory happens HERE, on every device, before any worker enqueues anything
    // (a device that dropped out of a collective would leave its peers waiting)
    {
        PairLayout lay(nullptr, cap, G, mg->dim, mg->binary);
        const size_t hneed = 256 + pm::align_up(sizeof(Out), 256) + pm::align_up(lay.blk_al * G, 256) + static_cast<size_t>(lay.mask_len);
        bool grow = mg->dev[0].lane[lane].hcap < hneed;
        for (int g = 0; g < G; ++g) grow = grow || mg->dev[g].lane[lane].cap < lay.bytes;
        (void)grow;      // (the lane is idle — checked above — so its buffers may be re-allocated)
        for (int g = 0; g < G; ++g) {
            Dev& d = mg->dev[g];
            int rc = lane_reserve(d, d.lane[lane], lay.bytes);
            if (rc != PM_OK) return rc;
            if (g == 0) {
                rc = pinned_reserve(d.lane[lane].hbuf, d.lane[lane].hcap, hneed);
                if (rc != PM_OK) return rc;
            }
        }
    }
    const int tk = ticket_open(mg);
    Ticket& T = mg->tk[tk % TICKET_RING];
    T.lane = lane; T.cap = cap; T.G = G; T.pair = true;
    int r0 = 0;
    for (int g = 0; g < G; ++g) {
        Job j;
        j.kind = 0; j.ticket = tk; j.lane = lane;
        j.d_desc1 = d_desc1 ? d_desc1[g] : nullptr;
        j.d_kp1 = d_kp1 ? d_kp1[g] : nullptr;
        j.h_desc1 = h_desc1; j.h_kp1 = h_kp1;
        j.rows = rows[g]; j.cap = cap; j.r0 = r0;
        j.ratio = ratio; j.flags = knn_flags; j.p = *p;
        post(mg->dev[g], j);
        r0 += rows[g];
    }
    T.n1 = r0;
    *ticket = tk;
    return PM_OK;
}
}  // namespace

extern "C" int pm_mgpu_submit_dev(pm_mgpu* mg, const void* const* d_desc1, const int32_t* rows, const float* const* d_kp1_xy,
                                  float ratio, int knn_flags, const pm_ransac_params* p, int* ticket)
{
    PM_REQUIRE(mg != nullptr && d_desc1 != nullptr && rows != nullptr && d_kp1_xy != nullptr && p != nullptr && ticket != nullptr,
               PM_E_INVALID, "null argument");
    int cap = 1;
    for (int g = 0; g < mg->n; ++g) {
        PM_REQUIRE(rows[g] >= 0 && (rows[g] == 0 || (d_desc1[g] && d_kp1_xy[g])), PM_E_INVALID, "bad per-device query block");
        cap = rows[g] > cap ? rows[g] : cap;
    }
    return submit_common(mg, d_desc1, d_kp1_xy, nullptr, nullptr, rows, cap, ratio, knn_flags, p, ticket);
}

extern "C" int pm_mgpu_collect(pm_mgpu* mg, int ticket, pm_pair_result* result, pm_match* good, uint8_t* mask)
{
    PM_REQUIRE(mg != nullptr && result != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(ticket >= 0 && ticket < mg->next_ticket && ticket + TICKET_RING > mg->next_ticket, PM_E_INVALID, "unknown ticket");
    Ticket& T = mg->tk[ticket % TICKET_RING];
    PM_REQUIRE(T.open, PM_E_INVALID, "ticket already collected");
    memset(result, 0, sizeof(*result));
    int rc = ticket_wait(mg, ticket);
    Dev& d0 = mg->dev[0];
    Lane& L = d0.lane[T.lane];
    if (rc != PM_OK) {                      // some device failed to enqueue: drain the lane everywhere before reporting
        (void)sync_all(mg);
        result->status = rc;
        return rc;
    }
    PM_HIP_CHECK(hipSetDevice(d0.device));
    PM_HIP_CHECK(hipEventSynchronize(L.done));
    PairLayout lay(nullptr, T.cap, T.G, mg->dim, mg->binary);
    Carve h(L.hbuf);
    const Out* ho = h.take<Out>(1);
    const char* hb = h.take<char>(lay.blk_al * T.G);
    const uint8_t* hm = h.take<uint8_t>(static_cast<size_t>(lay.mask_len));
    int total = 0, r0 = 0;
    for (int g = 0; g < T.G; ++g) {
        const char* b = hb + static_cast<size_t>(g) * lay.blk_al;
        int c = *reinterpret_cast<const int32_t*>(b);
        c = c < 0 ? 0 : (c > T.cap ? T.cap : c);
        if (good) {
            const pm_match* rec = reinterpret_cast<const pm_match*>(b + 16 + static_cast<size_t>(T.cap) * 16);
            for (int i = 0; i < c; ++i) {
                pm_match m = rec[i];
                m.queryIdx += r0;                            // block-local row -> row of the pair's query set
                good[total + i] = m;
            }
        }
        total += c;
        r0 += T.cap;
    }
    // (the block-local -> global row mapping above assumes equal blocks of T.cap rows, the last one possibly short: what
    // pm_mgpu_match_ransac and bench.py submit)
    result->n_good = total;
    result->best_key = ho->key;
    result->n_inliers = ho->ninl;
    memcpy(result->F, ho->F, sizeof(ho->F));
    if (mask && total > 0) memcpy(mask, hm, static_cast<size_t>(total));
    result->status = total < 8 ? PM_E_TOO_FEW : (ho->key == 0 ? PM_E_NO_MODEL : PM_OK);
    if (result->status != PM_OK) { memset(result->F, 0, sizeof(result->F)); result->n_inliers = 0; }
    return PM_OK;
}

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
    int rc = pm_mgpu_set_train(mg, desc2, n2, dim, binary, kp2_xy);
    if (rc != PM_OK) return rc;
    const int G = mg->n;
    const int cap = (n1 + G - 1) / G;                          // query rows per device (the last block may be short)
    const int lane = mg->next_ticket % mg->lanes;
    const size_t row = static_cast<size_t>(dim) * (binary ? 1 : sizeof(float));
    const size_t qb = static_cast<size_t>(n1) * row, kb = sizeof(float) * 2 * static_cast<size_t>(n1);
    rc = pinned_reserve(mg->hq[lane], mg->hq_cap[lane], pm::align_up(qb, 256) + kb);
    if (rc != PM_OK) return rc;
    memcpy(mg->hq[lane], desc1, qb);
    memcpy(mg->hq[lane] + pm::align_up(qb, 256), kp1_xy, kb);
    std::vector<int32_t> rows(G);
    for (int g = 0; g < G; ++g) {
        const int a = g * cap < n1 ? g * cap : n1, b = (g + 1) * cap < n1 ? (g + 1) * cap : n1;
        rows[g] = b - a;
    }
    int ticket = -1;
    rc = submit_common(mg, nullptr, nullptr, mg->hq[lane], reinterpret_cast<const float*>(mg->hq[lane] + pm::align_up(qb, 256)),
                       rows.data(), cap, ratio, knn_flags, p, &ticket);
    if (rc != PM_OK) return rc;
    pm_pair_result res;
    std::vector<uint8_t> hmask(static_cast<size_t>(G) * cap);
    rc = pm_mgpu_collect(mg, ticket, &res, good, hmask.data());
    if (rc != PM_OK) return rc;
    *n_good = res.n_good;
    if (best_key) *best_key = res.best_key;
    if (res.status == PM_E_TOO_FEW) { pm::set_error("need at least 8 correspondences, got %d", res.n_good); return PM_E_TOO_FEW; }
    if (res.status == PM_E_NO_MODEL) { pm::set_error("no valid model (all hypotheses degenerate)"); return PM_E_NO_MODEL; }
    if (F) memcpy(F, res.F, sizeof(res.F));
    if (mask) memcpy(mask, hmask.data(), static_cast<size_t>(res.n_good));
    if (n_inliers) *n_inliers = res.n_inliers;
    return PM_OK;
}

// ---- latency of the collective by itself ----------------------------------------------------------------
extern "C" int pm_mgpu_allgather_latency(pm_mgpu* mg, int bytes_per_device, int reps, double* us_per_collective)
{
    PM_REQUIRE(mg != nullptr && bytes_per_device >= 1 && reps >= 1 && us_per_collective != nullptr, PM_E_INVALID, "bad argument");
    PM_REQUIRE(!tickets_outstanding(mg), PM_E_INVALID, "collect the submitted pairs first");
    int rc = sync_all(mg);
    if (rc != PM_OK) return rc;
    for (int g = 0; g < mg->n; ++g) {
        rc = lane_reserve(mg->dev[g], mg->dev[g].lane[0], static_cast<size_t>(bytes_per_device) * mg->n + 256);
        if (rc != PM_OK) return rc;
    }
    std::vector<double> us(mg->n, 0.0);
    const int t = ticket_open(mg);
    for (int g = 0; g < mg->n; ++g) {
        Job j;
        j.kind = 4; j.ticket = t;
        j.probe_bytes = bytes_per_device; j.probe_reps = reps; j.probe_us = us.data();
        post(mg->dev[g], j);
    }
    rc = ticket_wait(mg, t);
    if (rc != PM_OK) return rc;
    double worst = 0.0;
    for (int g = 0; g < mg->n; ++g) worst = us[g] > worst ? us[g] : worst;
    *us_per_collective = worst;
    return PM_OK;
}

// ---- BASELINE config C5 behind the ABI: a batch of independent image pairs, pair p -> device p mod N ----------------
extern "C" int pm_mgpu_batch_run(pm_mgpu* mg, int n_lanes, int max_n1, int max_n2, int dim, const pm_pair_job* jobs, int n_jobs,
                                 float ratio, int knn_flags, const pm_ransac_params* p, pm_pair_result* results, pm_match* good,
                                 uint8_t* masks)
{
    PM_REQUIRE(mg != nullptr && p != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || (jobs && results)), PM_E_INVALID, "null jobs / results");
    if (n_jobs == 0) return PM_OK;
    for (int g = 0; g < mg->n; ++g) {                         // one streamed batch object per device, kept across calls
        Dev& d = mg->dev[g];
        const int sig[4] = {n_lanes, max_n1, max_n2, dim};
        if (d.batch && memcmp(sig, d.batch_sig, sizeof sig) != 0) { (void)pm_batch_destroy(d.batch); d.batch = nullptr; }
        if (!d.batch) {
            const int rc = pm_batch_create(d.device, n_lanes, max_n1, max_n2, dim, &d.batch);
            if (rc != PM_OK) return rc;
            memcpy(d.batch_sig, sig, sizeof sig);
        }
    }
    const int t = ticket_open(mg);
    for (int g = 0; g < mg->n; ++g) {
        Job j;
        j.kind = 2; j.ticket = t;
        j.batch = mg->dev[g].batch; j.jobs = jobs; j.n_jobs = n_jobs; j.stride = mg->n; j.first = g;
        j.ratio = ratio; j.knn_flags = knn_flags; j.p = *p;
        j.results = results; j.good = good; j.masks = masks; j.max_n1 = max_n1;
        post(mg->dev[g], j);
    }
    return ticket_wait(mg, t);
}

extern "C" int pm_mgpu_batch_set_option(pm_mgpu* mg, int option, int value)
{
    PM_REQUIRE(mg != nullptr, PM_E_INVALID, "null argument");
    for (int g = 0; g < mg->n; ++g) {
        for (int l = 0; l < MAX_LANES; ++l)
            if (mg->dev[g].lane[l].ctx) {
                const int rc = pm_ctx_set_option(mg->dev[g].lane[l].ctx, option, value);
                if (rc != PM_OK) return rc;
            }
        if (mg->dev[g].batch) {
            const int rc = pm_batch_set_option(mg->dev[g].batch, option, value);
            if (rc != PM_OK) return rc;
        }
    }
    return PM_OK;
}
