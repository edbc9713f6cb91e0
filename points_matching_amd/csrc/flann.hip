// flann.hip — approximate matcher compatible with `FlannBasedMatcher matcher;` (main.cpp:44, the reference's ACTIVE
// matcher object, called at main.cpp:46): cv::flann defaults = KDTreeIndexParams(4) + SearchParams(checks = 32,
// eps = 0, sorted), i.e. FLANN's forest of randomised kd-trees with a best-bin-first search shared by the trees.
// SURVEY.md 8f-4.  docs/SPEC.md S17 freezes what is implemented here (FLANN 1.6.11 as shipped with OpenCV 2.4.13,
// restated from its published algorithm [recalled: the library is not in /root/reference]):
//
//   build (host, C++, once per train image)   per tree: shuffle the point ids; divide recursively: mean and variance
//       per dimension over the first <= 101 ids of the node, cut dimension = one of the 5 highest-variance dimensions
//       picked at random, cut value = that dimension's mean, Hoare partition with FLANN's lim1/lim2 balancing rule;
//       a leaf is one point.  FLANN draws from C rand(); here every draw is a counter-based mix64 stream keyed by
//       (seed, tree) — reproducible on every machine, which rand() is not (and why no bit-level target exists).
//   search (HIP, one lane per query)   descend every tree to a leaf pushing the far child with its lower bound
//       mindist + (q[d] - cut)^2 on ONE binary heap per query; then pop the closest branch and descend again, while
//       fewer than `checks` distinct points have been examined (or the result is not full); a point is examined once
//       (per-query bit set); branches whose bound exceeds the current k-th distance are dropped.  Distances are the
//       canonical squared L2 of docs/SPEC.md S1, so a neighbour FLANN finds carries the exact matcher's bits.
//
// Data layout in HBM: nodes as 16-byte records {child1, child2, divfeat | point id, divval}, 2n-1 per tree; per query
// a 1024-entry heap (distance and node arrays interleaved across queries, so lanes of a wave touch neighbouring words)
// and an n-bit "examined" set.  The search is latency-bound pointer chasing by nature (4 x ~13 dependent node reads
// plus 32 row reads per query); what the GPU adds is 8192+ queries in flight.  At the reference's own sizes
// (10^2..10^3 descriptors) the exact matrix-core matcher is faster than any tree search — this path exists for
// behavioural compatibility with main.cpp:44, and pm_cli --matcher flann selects it.
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "pm_common.hpp"

namespace {

constexpr int FL_SAMPLE_MEAN = 100;   // FLANN: mean / variance from the first SAMPLE_MEAN + 1 ids of a node
constexpr int FL_RAND_DIM = 5;        // FLANN: cut dimension drawn among the 5 highest variances
constexpr int FL_HEAP_CAP = 1024;     // branches kept per query (FLANN sizes its heap by the point count; 32 checks
                                      // push ~(4 + 40) x tree depth entries: < 1024 up to 2^20 points; a full heap drops)
constexpr int FL_MAX_K = 4;

struct FlNode {
    int32_t child1, child2;           // node indices; -1 / -1 for a leaf
    int32_t divfeat;                  // cut dimension, or the point id in a leaf
    float divval;
};
static_assert(sizeof(FlNode) == 16, "node record");

inline uint64_t fl_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct FlRng {                        // counter-based: draw i of stream (seed, tree) is a pure function of (seed, tree, i)
    uint64_t key, ctr = 0;
    FlRng(uint64_t seed, int tree) : key(fl_mix64(seed ^ 0x9E3779B97F4A7C15ULL) ^ fl_mix64(static_cast<uint64_t>(tree) + 0xD1B54A32D192ED03ULL)) {}
    uint32_t below(uint32_t n)        // uniform in [0, n)
    {
        const uint64_t r = fl_mix64(key + (++ctr) * 0x9E3779B97F4A7C15ULL);
        return static_cast<uint32_t>(((r >> 32) * static_cast<uint64_t>(n)) >> 32);
    }
};

struct Builder {
    const float* data;
    int dim;
    std::vector<FlNode>* nodes;
    FlRng* rng;
    std::vector<float> mean, var;

    int select_division()
    {
        int num = 0;
        int topind[FL_RAND_DIM];
        for (int i = 0; i < dim; ++i) {                       // keep the FL_RAND_DIM largest variances, sorted
            if (num < FL_RAND_DIM || var[i] > var[topind[num - 1]]) {
                if (num < FL_RAND_DIM) topind[num++] = i;
                else topind[num - 1] = i;
                int j = num - 1;
                while (j > 0 && var[topind[j]] > var[topind[j - 1]]) { std::swap(topind[j], topind[j - 1]); --j; }
            }
        }
        return topind[rng->below(static_cast<uint32_t>(num))];
    }

    void plane_split(int* ind, int count, int cutfeat, float cutval, int& lim1, int& lim2)
    {
        int left = 0, right = count - 1;
        for (;;) {                                            // ids with value < cutval to the front
            while (left <= right && data[static_cast<size_t>(ind[left]) * dim + cutfeat] < cutval) ++left;
            while (left <= right && data[static_cast<size_t>(ind[right]) * dim + cutfeat] >= cutval) --right;
            if (left > right) break;
            std::swap(ind[left], ind[right]); ++left; --right;
        }
        lim1 = left;
        right = count - 1;
        for (;;) {                                            // then those <= cutval
            while (left <= right && data[static_cast<size_t>(ind[left]) * dim + cutfeat] <= cutval) ++left;
            while (left <= right && data[static_cast<size_t>(ind[right]) * dim + cutfeat] > cutval) --right;
            if (left > right) break;
            std::swap(ind[left], ind[right]); ++left; --right;
        }
        lim2 = left;
    }

    int divide(int* ind, int count)
    {
        const int me = static_cast<int>(nodes->size());
        nodes->push_back(FlNode{-1, -1, 0, 0.f});
        if (count == 1) {
            (*nodes)[me].divfeat = ind[0];
            return me;
        }
        const int cnt = std::min(FL_SAMPLE_MEAN + 1, count);
        std::fill(mean.begin(), mean.end(), 0.f);
        std::fill(var.begin(), var.end(), 0.f);
        for (int j = 0; j < cnt; ++j) {
            const float* v = data + static_cast<size_t>(ind[j]) * dim;
            for (int k = 0; k < dim; ++k) mean[k] += v[k];
        }
        for (int k = 0; k < dim; ++k) mean[k] /= static_cast<float>(cnt);
        for (int j = 0; j < cnt; ++j) {
            const float* v = data + static_cast<size_t>(ind[j]) * dim;
            for (int k = 0; k < dim; ++k) { const float d = v[k] - mean[k]; var[k] += d * d; }
        }
        const int cutfeat = select_division();
        const float cutval = mean[cutfeat];
        int lim1, lim2;
        plane_split(ind, count, cutfeat, cutval, lim1, lim2);
        int index;
        if (lim1 > count / 2) index = lim1;
        else if (lim2 < count / 2) index = lim2;
        else index = count / 2;
        if (lim1 == count || lim2 == 0) index = count / 2;    // all remaining values identical: split in the middle
        const int c1 = divide(ind, index);
        const int c2 = divide(ind + index, count - index);
        FlNode& nd = (*nodes)[me];
        nd.child1 = c1; nd.child2 = c2; nd.divfeat = cutfeat; nd.divval = cutval;
        return me;
    }
};

// canonical squared L2 of docs/SPEC.md S1 (the op order of knn_l2.hip / the oracle: not contracted, TU is -ffp-contract=off)
__device__ __forceinline__ float fl_l2sqr(const float* __restrict__ a, const float* __restrict__ b, int dim)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 8 <= dim; j += 8) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const float t = a[j + l] - b[j + l];
            const float p = t * t;
            acc[l] = acc[l] + p;
        }
    }
    const float s0 = acc[0] + acc[4], s1 = acc[1] + acc[5], s2 = acc[2] + acc[6], s3 = acc[3] + acc[7];
    float d = ((s0 + s1) + s2) + s3;
    for (; j < dim; ++j) {
        const float t = a[j] - b[j];
        const float p = t * t;
        d = d + p;
    }
    return d;
}

struct FlSearch {
    const FlNode* nodes;
    int roots[16];
    int trees;
    const float* T;
    int nt, dim, k, checks;
    unsigned* checked;                // [nq_pad][words]: examined-point bit set of every query
    int nq_pad, words;
};

// SPEC S1 by 8 consecutive lanes (lane l = accumulator l of the canonical form, canonical combine): one row costs one
// memory round trip instead of dim/8 dependent ones.  Valid in every lane of the wave (the 8 partial sums are broadcast).
__device__ __forceinline__ float fl_l2sqr_coop(const float* __restrict__ a, const float* __restrict__ b, int dim, int lane)
{
    const int l = lane & 7;
    float acc = 0.f;
    const int full8 = dim & ~7;
    for (int j0 = 0; j0 < full8; j0 += 64) {                 // up to 8 column groups' loads in flight together
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = j0 + 8 * u + l;
            const int cc = c < full8 ? c : l;                // (clamped: the load stays unconditional)
            av[u] = a[cc]; bv[u] = b[cc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + 8 * u < full8) { const float t = av[u] - bv[u]; const float p = t * t; acc = acc + p; }
    }
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __shfl(acc, i, 64);   // accumulator i (lanes 0..7 hold them)
    const float s0 = e[0] + e[4], s1 = e[1] + e[5], s2 = e[2] + e[6], s3 = e[3] + e[7];
    float d = ((s0 + s1) + s2) + s3;
    for (int j = full8; j < dim; ++j) {
        const float t = a[j] - b[j];
        const float p = t * t;
        d = d + p;
    }
    return d;
}

// ONE WAVE PER QUERY (round 3; round 2: one lane per query — 300 queries were five waves on the whole chip, each lane
// chasing its own pointers: 1.3 ms).  The best-bin-first search is a dependent chain by nature; here every value of the
// chain is wave-uniform, so the branches cost nothing, the heap lives in LDS (lane 0 writes, every lane reads the same
// word), the row distances are evaluated by 8 lanes in the canonical order, and 300 queries are 300 waves on 256 CUs.
// Same visiting order and same bits as pmo_flann_search (tests/test_flann_gpu.py).
constexpr int FL_QPW = 4;             // queries (waves) per workgroup
__global__ __launch_bounds__(64 * FL_QPW) void flann_search(FlSearch s, const float* __restrict__ Q, int nq, pm_match* __restrict__ out)
{
    __shared__ float hd[FL_QPW][FL_HEAP_CAP];
    __shared__ int hn[FL_QPW][FL_HEAP_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * FL_QPW + wave);
    if (q >= nq) return;                                       // wave-uniform; no workgroup barriers below
    float* heap_d = hd[wave];
    int* heap_n = hn[wave];
    const float* qv = Q + static_cast<size_t>(q) * s.dim;
    unsigned* chk = s.checked + static_cast<size_t>(q) * s.words;
    float rd[FL_MAX_K];                                        // result set, ascending squared distances
    int ri[FL_MAX_K];
#pragma unroll
    for (int i = 0; i < FL_MAX_K; ++i) { rd[i] = __builtin_inff(); ri[i] = -1; }
    int rcount = 0, hcount = 0, check_count = 0;
    for (int w = 0; w < s.words; ++w) chk[w] = 0u;           // (every lane, same words: see the leaf test below)

    auto worst = [&]() -> float {
        float v = rd[0];
#pragma unroll
        for (int i = 1; i < FL_MAX_K; ++i) v = (i == s.k - 1) ? rd[i] : v;
        return s.k == 1 ? rd[0] : v;                            // rd[k-1]: +inf until the set is full
    };
    // binary min-heap on (bound, node id): the id breaks ties between equal bounds, so the order in which branches are
    // taken up is a property of the forest and the query, not of a heap implementation (SPEC S17)
    auto before = [](float da, int na, float db, int nb) -> bool { return da < db || (da == db && na < nb); };
    auto put = [&](int i, float d, int n) { if (lane == 0) { heap_d[i] = d; heap_n[i] = n; } };
    auto heap_push = [&](float d, int node) {
        if (hcount >= FL_HEAP_CAP) return;                      // FLANN's Heap::insert also drops when full
        int i = hcount++;
        while (i > 0) {                                         // sift up
            const int p = (i - 1) >> 1;
            const float pd = heap_d[p];
            const int pn = heap_n[p];
            if (!before(d, node, pd, pn)) break;
            put(i, pd, pn);
            i = p;
        }
        put(i, d, node);
    };
    auto heap_pop = [&](float& d, int& node) -> bool {
        if (hcount == 0) return false;
        d = heap_d[0];
        node = heap_n[0];
        --hcount;
        if (hcount > 0) {
            const float ld = heap_d[hcount];
            const int ln = heap_n[hcount];
            int i = 0;
            for (;;) {                                          // sift down
                int c = 2 * i + 1;
                if (c >= hcount) break;
                float cd = heap_d[c];
                int cn = heap_n[c];
                if (c + 1 < hcount) {
                    const float cd2 = heap_d[c + 1];
                    const int cn2 = heap_n[c + 1];
                    if (before(cd2, cn2, cd, cn)) { cd = cd2; cn = cn2; ++c; }
                }
                if (!before(cd, cn, ld, ln)) break;
                put(i, cd, cn);
                i = c;
            }
            put(i, ld, ln);
        }
        return true;
    };
    // searchLevel of FLANN's KDTreeIndex, iteratively: walk to the leaf, far children to the heap
    auto descend = [&](int node, float mindist) {
        for (;;) {
            if (worst() < mindist) return;
            const FlNode nd = s.nodes[node];
            if (nd.child1 < 0) {                                // leaf
                const int idx = nd.divfeat;
                const unsigned bit = 1u << (idx & 31);
                const unsigned wv = chk[idx >> 5];
                if ((wv & bit) || (check_count >= s.checks && rcount >= s.k)) return;
                chk[idx >> 5] = wv | bit;                    // every lane stores the same word: each later reads its own store
                ++check_count;
                const float d = fl_l2sqr_coop(qv, s.T + static_cast<size_t>(idx) * s.dim, s.dim, lane);
                // KNNResultSet::addPoint: insert in ascending order (an equal distance goes behind the earlier one)
                if (rcount < s.k || d < worst()) {
                    float cd = d;
                    int ci = idx;
#pragma unroll
                    for (int i = 0; i < FL_MAX_K; ++i) {
                        if (i < s.k && cd < rd[i]) {
                            const float td = rd[i]; const int ti = ri[i];
                            rd[i] = cd; ri[i] = ci;
                            cd = td; ci = ti;
                        }
                    }
                    if (rcount < s.k) ++rcount;
                }
                return;
            }
            const float val = qv[nd.divfeat];
            const float diff = val - nd.divval;
            const int best = diff < 0.f ? nd.child1 : nd.child2;
            const int other = diff < 0.f ? nd.child2 : nd.child1;
            const float p = diff * diff;
            const float new_d = mindist + p;
            if (new_d < worst() || rcount < s.k) heap_push(new_d, other);
            node = best;
        }
    };
    for (int t = 0; t < s.trees; ++t) descend(s.roots[t], 0.f);
    float bd;
    int bn;
    while (heap_pop(bd, bn) && (check_count < s.checks || rcount < s.k)) descend(bn, bd);

    for (int c = 0; c < s.k; ++c) {
        pm_match m;
        m.queryIdx = q;
        m.imgIdx = 0;
        float d = rd[0];
        int id = ri[0];
#pragma unroll
        for (int i = 1; i < FL_MAX_K; ++i) { if (i == c) { d = rd[i]; id = ri[i]; } }
        m.trainIdx = id;
        m.distance = id >= 0 ? __builtin_sqrtf(d) : __builtin_inff();
        if (lane == 0) out[static_cast<size_t>(q) * s.k + c] = m;
    }
}

}  // namespace

struct pm_flann_index {
    int device = 0;
    int nt = 0, dim = 0, trees = 0, checks = 0;
    uint64_t seed = 0;
    int roots[16] = {};
    std::vector<FlNode> h_nodes;      // kept for pm_flann_export (tests)
    FlNode* d_nodes = nullptr;
    float* d_train = nullptr;
    // per-call scratch, grow-only
    unsigned* d_checked = nullptr;
    int scratch_nq = 0;
};

extern "C" int pm_flann_destroy(pm_flann_index* ix)
{
    if (!ix) return PM_OK;
    (void)hipSetDevice(ix->device);
    (void)hipFree(ix->d_nodes); (void)hipFree(ix->d_train);
    (void)hipFree(ix->d_checked);
    delete ix;
    return PM_OK;
}

extern "C" int pm_flann_build(pm_ctx* ctx, const float* train, int nt, int dim, const pm_flann_params* prm, pm_flann_index** out)
{
    PM_REQUIRE(ctx != nullptr && out != nullptr, PM_E_INVALID, "null argument");
    *out = nullptr;
    PM_REQUIRE(train != nullptr && nt >= 1 && dim >= 1, PM_E_INVALID, "need a non-empty train set");
    const int trees = prm && prm->trees > 0 ? prm->trees : 4;           // KDTreeIndexParams(4)
    const int checks = prm && prm->checks > 0 ? prm->checks : 32;       // SearchParams(32)
    PM_REQUIRE(trees <= 16, PM_E_INVALID, "at most 16 trees");
    pm_flann_index* ix = new (std::nothrow) pm_flann_index;
    PM_REQUIRE(ix != nullptr, PM_E_NOMEM, "out of host memory");
    ix->device = ctx->device; ix->nt = nt; ix->dim = dim; ix->trees = trees; ix->checks = checks;
    ix->seed = prm ? prm->seed : 0;
    try {
        ix->h_nodes.reserve(static_cast<size_t>(trees) * (2 * static_cast<size_t>(nt) - 1));
        std::vector<int> ind(static_cast<size_t>(nt));
        for (int t = 0; t < trees; ++t) {
            FlRng rng(ix->seed, t);
            for (int i = 0; i < nt; ++i) ind[i] = i;
            for (int i = nt - 1; i > 0; --i) std::swap(ind[i], ind[rng.below(static_cast<uint32_t>(i + 1))]);   // Fisher-Yates
            Builder b{train, dim, &ix->h_nodes, &rng, std::vector<float>(dim), std::vector<float>(dim)};
            ix->roots[t] = b.divide(ind.data(), nt);
        }
    } catch (const std::bad_alloc&) {
        delete ix;
        pm::set_error("out of host memory while building the kd-forest");
        return PM_E_NOMEM;
    }
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t nb = sizeof(FlNode) * ix->h_nodes.size(), tb = sizeof(float) * static_cast<size_t>(nt) * dim;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ix->d_nodes), nb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ix->d_train), tb);
    if (e == hipSuccess) e = hipMemcpy(ix->d_nodes, ix->h_nodes.data(), nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ix->d_train, train, tb, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        pm::set_error("kd-forest upload failed: %s", hipGetErrorString(e));
        (void)pm_flann_destroy(ix);
        return PM_E_HIP;
    }
    *out = ix;
    return PM_OK;
}

extern "C" int pm_flann_knn_l2_f32_dev(pm_ctx* ctx, pm_flann_index* ix, const float* d_q, int nq, int k, pm_match* d_out)
{
    PM_REQUIRE(ctx != nullptr && ix != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(nq >= 0 && k >= 1 && k <= FL_MAX_K, PM_E_INVALID, "need nq >= 0, 1 <= k <= 4");
    PM_REQUIRE(nq == 0 || (d_q && d_out), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE(ctx->device == ix->device, PM_E_INVALID, "index lives on another device");
    if (nq == 0) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const int nq_pad = (nq + 63) / 64 * 64;
    const int words = (ix->nt + 31) / 32;
    if (nq_pad > ix->scratch_nq) {
        PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ix->d_checked);
        ix->d_checked = nullptr; ix->scratch_nq = 0;
        PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ix->d_checked), sizeof(unsigned) * static_cast<size_t>(words) * nq_pad));
        ix->scratch_nq = nq_pad;
    }
    FlSearch s{};
    s.nodes = ix->d_nodes;
    for (int t = 0; t < ix->trees; ++t) s.roots[t] = ix->roots[t];
    s.trees = ix->trees; s.T = ix->d_train; s.nt = ix->nt; s.dim = ix->dim; s.k = k; s.checks = ix->checks;
    s.checked = ix->d_checked;
    s.nq_pad = ix->scratch_nq; s.words = words;
    pm::ScopedKernelTime t(ctx, "flann_search");
    hipLaunchKernelGGL(flann_search, dim3((nq + FL_QPW - 1) / FL_QPW), dim3(64 * FL_QPW), 0, ctx->stream, s, d_q, nq, d_out);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

extern "C" int pm_flann_knn_l2_f32(pm_ctx* ctx, pm_flann_index* ix, const float* q, int nq, int k, pm_match* out)
{
    PM_REQUIRE(ctx != nullptr && ix != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(nq >= 0 && k >= 1 && k <= FL_MAX_K, PM_E_INVALID, "need nq >= 0, 1 <= k <= 4");
    PM_REQUIRE(nq == 0 || (q && out), PM_E_INVALID, "null query/output pointer");
    if (nq == 0) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t qb = sizeof(float) * static_cast<size_t>(nq) * ix->dim, ob = sizeof(pm_match) * static_cast<size_t>(nq) * k;
    int rc = pm::arena_reserve(ctx, pm::align_up(qb, 256) + pm::align_up(ob, 256) + 1024);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    float* dq = static_cast<float*>(pm::arena_take(ctx, qb));
    pm_match* dout = static_cast<pm_match*>(pm::arena_take(ctx, ob));
    PM_REQUIRE(dq && dout, PM_E_NOMEM, "scratch arena too small");
    PM_HIP_CHECK(hipMemcpyAsync(dq, q, qb, hipMemcpyHostToDevice, ctx->stream));
    rc = pm_flann_knn_l2_f32_dev(ctx, ix, dq, nq, k, dout);
    if (rc != PM_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    PM_HIP_CHECK(hipMemcpyAsync(out, dout, ob, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

// Test / inspection access to the forest: node records (16 bytes each: child1, child2, divfeat | point id, divval) and roots.
extern "C" int pm_flann_export(const pm_flann_index* ix, int32_t* n_nodes, int32_t* roots, void* nodes, int32_t cap_nodes)
{
    PM_REQUIRE(ix != nullptr && n_nodes != nullptr, PM_E_INVALID, "null argument");
    *n_nodes = static_cast<int32_t>(ix->h_nodes.size());
    if (roots) for (int t = 0; t < ix->trees; ++t) roots[t] = ix->roots[t];
    if (nodes) {
        PM_REQUIRE(cap_nodes >= *n_nodes, PM_E_INVALID, "node buffer too small");
        memcpy(nodes, ix->h_nodes.data(), sizeof(FlNode) * ix->h_nodes.size());
    }
    return PM_OK;
}
