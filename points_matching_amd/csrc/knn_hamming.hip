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
//
// 256-bit descriptors with k <= 2 (ORB + ratio test: config C4) take the matrix-core route
// instead: the bits are expanded to +-1 bytes, so that v_mfma_i32_32x32x32_i8 yields
// dot = 256 - 2*hamming for 32x32 pairs per instruction (16x the VALU rate); the coarse kernel is
// knn_coarse.hip's row-streaming kernel (shared with the f16 L2 route) keeping, per lane stream,
// the 4 best 8-row groups, and knn_hamming_refine re-evaluates the candidate rows with popcounts.
// The integers are exact on both sides, so the window logic has no epsilon: a row can only be
// missed if its sub-list overflowed, which the 4th entry reveals (then that sub-list's rows are
// scanned).  Rows padding the last 128-row tile are all-zero (dot = 0): groups that contain such a
// row do not take part in the choice of tau, but are expanded like any other candidate.
#include <cstdlib>

#include "knn_shared.hpp"

namespace {
using namespace pm_knn;

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

// ---------------------------------------------------------------------------------------------
// i8 route
// ---------------------------------------------------------------------------------------------
// 4 descriptor bits -> 4 bytes: 0x01 where the bit is set, 0xFF (-1) where it is clear
__device__ __forceinline__ uint32_t expand_nibble(uint32_t nib)
{
    const uint32_t ones = (nib * 0x00204081u) & 0x01010101u;
    return 0xFFFFFFFFu ^ (ones * 0xFEu);
}

// one thread per (row, descriptor word): 32 bits -> 32 bytes.  Rows >= n of the padded copy are zero.
__global__ __launch_bounds__(256) void knn_hamming_expand(const uint32_t* __restrict__ Q, int nq, int nq_pad,
                                                          const uint32_t* __restrict__ T, int nt, int nt_pad,
                                                          uint4* __restrict__ Qe, uint4* __restrict__ Te)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int total_q = nq_pad * I8_NCH;
    const bool is_q = gid < total_q;
    const int e = is_q ? gid : gid - total_q;
    if (!is_q && e >= nt_pad * I8_NCH) return;
    const int row = e / I8_NCH, c = e % I8_NCH;
    const int n = is_q ? nq : nt;
    uint4 lo = uint4{0u, 0u, 0u, 0u}, hi = uint4{0u, 0u, 0u, 0u};
    if (row < n) {
        const uint32_t w = (is_q ? Q : T)[static_cast<size_t>(row) * 8 + c];
        lo = uint4{expand_nibble(w & 15u), expand_nibble((w >> 4) & 15u), expand_nibble((w >> 8) & 15u),
                   expand_nibble((w >> 12) & 15u)};
        hi = uint4{expand_nibble((w >> 16) & 15u), expand_nibble((w >> 20) & 15u), expand_nibble((w >> 24) & 15u),
                   expand_nibble(w >> 28)};
    }
    uint4* dst = (is_q ? Qe : Te) + static_cast<size_t>(row) * I8_ROW16 + 2 * c;
    dst[0] = lo;
    dst[1] = hi;
}

using pm::wave_min_u64;
using pm::wave_min_u32;

// (distance, row) keys: 32-bit (distance << 23 | row) while the train set has fewer than 2^23 rows
// (distance <= 256), 64-bit otherwise.  Half the VALU work of the one-wave-per-query refinement is
// key handling, so the narrow form matters.
struct Key32 {
    typedef uint32_t type;
    static constexpr type NONE = 0xFFFFFFFFu;
    static __device__ __forceinline__ type make(int d, int row) { return (static_cast<uint32_t>(d) << 23) | static_cast<uint32_t>(row); }
    static __device__ __forceinline__ int dist(type k) { return static_cast<int>(k >> 23); }
    static __device__ __forceinline__ int row(type k) { return static_cast<int>(k & 0x7FFFFFu); }
    static __device__ __forceinline__ type wave_min(type k) { return wave_min_u32(k); }
};
struct Key64 {
    typedef unsigned long long type;
    static constexpr type NONE = ~0ull;
    static __device__ __forceinline__ type make(int d, int row) { return (static_cast<type>(d) << 32) | static_cast<unsigned>(row); }
    static __device__ __forceinline__ int dist(type k) { return static_cast<int>(k >> 32); }
    static __device__ __forceinline__ int row(type k) { return static_cast<int>(k & 0xFFFFFFFFull); }
    static __device__ __forceinline__ type wave_min(type k) { return wave_min_u64(k); }
};

template <typename KT>
struct Best2 {
    KT a, b;       // a <= b
    __device__ __forceinline__ void insert(KT key)
    {
        if (key < a) { b = a; a = key; }
        else if (key < b) b = key;
    }
};

__device__ __forceinline__ int hamming256(const uint4 q0, const uint4 q1, const uint32_t* __restrict__ T, int row)
{
    const uint4* tr = reinterpret_cast<const uint4*>(T + static_cast<size_t>(row) * 8);
    const uint4 t0 = tr[0], t1 = tr[1];
    const int d = __builtin_popcount(q0.x ^ t0.x) + __builtin_popcount(q0.y ^ t0.y) + __builtin_popcount(q0.z ^ t0.z) +
                  __builtin_popcount(q0.w ^ t0.w) + __builtin_popcount(q1.x ^ t1.x) + __builtin_popcount(q1.y ^ t1.y) +
                  __builtin_popcount(q1.z ^ t1.z) + __builtin_popcount(q1.w ^ t1.w);
    return d;
}

// All rows of one sub-list's lane stream (split, lane half hh): LANES lanes, U rows per lane and pass with the 2 x U loads
// of a pass issued together (clamped addresses, no condition in front of a load).  Round 3: the first form of this scan
// took one row per lane and pass, i.e. tiles_per_split serial memory round trips of ~1k cycles each — at config C4 (64
// tiles per split) ~25 us for the handful of queries per launch that need it, which was most of the kernel's 31 us: a
// kernel lasts as long as its slowest wave.
template <int U, int LANES, typename K>
__device__ __forceinline__ void scan_sublist(Best2<typename K::type>& best, const uint4 q0, const uint4 q1,
                                             const uint32_t* __restrict__ T, int nt, int split, int hh, int tiles_per_split, int l)
{
    const int n = tiles_per_split * 64;
    for (int base = 0; base < n; base += LANES * U) {
        int row[U];
        bool ok[U];
        uint4 t0[U], t1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LANES + l;
            const int tile = idx >> 6, rem = idx & 63;
            row[u] = (split * tiles_per_split + tile) * H_TT + 32 * (rem >> 4) + 8 * ((rem >> 2) & 3) + 4 * hh + (rem & 3);
            ok[u] = idx < n && row[u] < nt;
            const uint4* tr = reinterpret_cast<const uint4*>(T + static_cast<size_t>(ok[u] ? row[u] : 0) * 8);
            t0[u] = tr[0];
            t1[u] = tr[1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int d = __builtin_popcount(q0.x ^ t0[u].x) + __builtin_popcount(q0.y ^ t0[u].y) + __builtin_popcount(q0.z ^ t0[u].z) +
                          __builtin_popcount(q0.w ^ t0[u].w) + __builtin_popcount(q1.x ^ t1[u].x) + __builtin_popcount(q1.y ^ t1[u].y) +
                          __builtin_popcount(q1.z ^ t1[u].z) + __builtin_popcount(q1.w ^ t1[u].w);
            if (ok[u]) best.insert(K::make(d, row[u]));
        }
    }
}

// The rare whole-sub-list scans of a workgroup's queries, done by ALL of its 256 threads (round 3).  A launch lasts as
// long as its slowest wave: scanning a 4096-row lane stream with the 64 (or 16) lanes that own the query was 16 (64)
// dependent passes — most of the kernel's time at config C4 although < 0.3 % of the queries need it.  Phase A of the
// kernels only RECORDS the sub-lists (HmScan in LDS); here every thread scans its share for one owner at a time and leaves
// its two best keys in LDS, from where the owner's lanes merge them.  NOWN owners per workgroup, LPO lanes per owner.
template <int NOWN, int MAXSUB, typename K>
struct HmScan {
    uint4 qw[NOWN][2];                                       // the owner's descriptor (first member: 16-byte aligned)
    typename K::type keys[256][2];
    int nsub[NOWN];
    int sub[NOWN][MAXSUB];
};
template <int NOWN, int MAXSUB, int LPO, typename K>
__device__ __forceinline__ void wg_scan_phase(HmScan<NOWN, MAXSUB, K>& hs, Best2<typename K::type>& best, int owner, int l,
                                              const uint32_t* __restrict__ T, int nt, int tiles_per_split)
{
    __syncthreads();                                         // every owner's record is complete
    for (int w = 0; w < NOWN; ++w) {
        const int n = hs.nsub[w];                            // workgroup-uniform
        if (n == 0) continue;
        const uint4 q0 = hs.qw[w][0], q1 = hs.qw[w][1];
        Best2<typename K::type> tb{K::NONE, K::NONE};
        for (int j = 0; j < n; ++j) {
            const int sb = hs.sub[w][j];
            scan_sublist<4, 256, K>(tb, q0, q1, T, nt, sb >> 1, sb & 1, tiles_per_split, static_cast<int>(threadIdx.x));
        }
        hs.keys[threadIdx.x][0] = tb.a;
        hs.keys[threadIdx.x][1] = tb.b;
        __syncthreads();
        if (owner == w) {
#pragma unroll
            for (int u = 0; u < 256 / LPO; ++u) {
                best.insert(hs.keys[l + LPO * u][0]);
                best.insert(hs.keys[l + LPO * u][1]);
            }
        }
        __syncthreads();                                     // keys[] is free again
    }
}

// One wave per query.  cand: [nq][slots] ints, sub-list s = entries 4s..4s+3 in descending order,
// s = split*2 + lane half; entry = (dot << shift) | id, id = (tile_in_split*8 + block*2 + group) * 2 + lane half;
// group (block, g, half hh) = rows 32*block + 16*g + 4*hh + {0,1,2,3, 8,9,10,11} of the tile.
constexpr int HR_MAXE = 8;          // entries per lane: slots <= 512
// NE = entries per lane actually needed (ceil(slots / 64) rounded up to 1, 2, 4 or 8): the per-entry
// work below is unrolled NE times, and with few splits (C4: 32 slots) NE = 1 instead of 8
template <int NE, typename K>
__global__ __launch_bounds__(256) void knn_hamming_refine(const uint32_t* __restrict__ Q, const uint32_t* __restrict__ T,
                                                          int nq, int nt, int k, const int* __restrict__ cand,
                                                          int slots, int tiles_per_split, int shift,
                                                          pm_match* __restrict__ out)
{
    __shared__ int clist[4][64 * NE];
    __shared__ __attribute__((aligned(16))) HmScan<4, 16 * NE, K> hs;      // (a query has slots / 4 <= 16 * NE sub-lists)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    const bool live = q < nq;                                // (dead waves shadow the last query: the workgroup has barriers)
    const int qc = live ? q : nq - 1;
    const uint4* qr = reinterpret_cast<const uint4*>(Q + static_cast<size_t>(qc) * 8);
    const uint4 q0 = qr[0], q1 = qr[1];
    const int gmask = (1 << shift) - 1;

    int v[NE], dc[NE];
    bool whole[NE];                                     // every row of the group is a real row
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = lane + 64 * i;
        v[i] = e < slots ? cand[static_cast<size_t>(qc) * slots + e] : I8_EMPTY;
        dc[i] = v[i] == I8_EMPTY ? 0x7FFFFFF0 : ((I8_BITS - (v[i] >> shift)) >> 1);      // coarse (= exact) distance
        const int gid2 = v[i] & gmask, gid = gid2 >> 1, split = e >> 3;
        const int last = (split * tiles_per_split + (gid >> 3)) * H_TT + 32 * ((gid >> 1) & 3) + 16 * (gid & 1) +
                         4 * (gid2 & 1) + 11;
        whole[i] = v[i] != I8_EMPTY && last < nt;
    }
    // tau = k-th smallest distance over the entries whose rows are all real (k distinct rows lie within
    // tau); fewer than k such entries (tiny train sets): tau = "everything"
    // (keys: distance << 16 | entry index, distances <= 256 or the 0x7FFF sentinel, entries < 512)
    unsigned lastk = 0u;
    int tau = 0;
    for (int c = 0; c < k; ++c) {
        unsigned m = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const unsigned dt = whole[i] ? static_cast<unsigned>(dc[i]) : 0x7FFFu;
            const unsigned key = (dt << 16) | static_cast<unsigned>(lane + 64 * i);
            if ((c == 0 || key > lastk) && key < m) m = key;
        }
        m = wave_min_u32(m);
        lastk = m;
        tau = static_cast<int>(m >> 16);
    }
    if (tau == 0x7FFF) tau = 0x7FFFFFFF;                     // fewer than k whole groups: everything is a candidate
    // sub-lists whose 4th entry is within tau may have dropped a row within tau: scan them whole
    Best2<typename K::type> best{K::NONE, K::NONE};
    int total = 0, nfull = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        if (64 * i >= slots) break;                          // wave-uniform
        const bool within = v[i] != I8_EMPTY && dc[i] <= tau;
        const unsigned long long full = __ballot(within && (lane & 3) == 3);
        const bool my_full = (full >> (lane | 3)) & 1ull;
        const unsigned long long cm = __ballot(within && !my_full);
        if (within && !my_full)
            clist[wave][total + __popcll(cm & ((1ull << lane) - 1ull))] = (v[i] & gmask) | (((lane + 64 * i) >> 2) << 16);
        total += __popcll(cm);
        // rare: a sub-list whose 4th entry is within tau is scanned whole — by the workgroup, below; recorded here
        if (full) {
            const unsigned long long fb = (1ull << lane) & full;
            if (fb && live) hs.sub[wave][nfull + __popcll(full & ((1ull << lane) - 1ull))] = (lane + 64 * i) >> 2;
            nfull += __popcll(full);
        }
    }
    if (lane == 0) hs.nsub[wave] = live ? nfull : 0;
    if (lane == 0) { hs.qw[wave][0] = q0; hs.qw[wave][1] = q1; }
    // the list entries were written by other lanes of this wave: LDS stores before the loads, explicitly
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // candidate groups: 8 rows each, one row per lane, 8 groups per round
    for (int base = 0; base < 8 * total; base += 64) {
        const int t = base + lane;
        if (t < 8 * total) {
            const int ent = clist[wave][t >> 3];
            const int gid2 = ent & 0xFFFF, gid = gid2 >> 1, hh = gid2 & 1, split = ent >> 17;      // ent >> 16 = split*2 + half
            const int s8 = t & 7;
            const int row = (split * tiles_per_split + (gid >> 3)) * H_TT + 32 * ((gid >> 1) & 3) + 16 * (gid & 1) + 4 * hh +
                            8 * (s8 >> 2) + (s8 & 3);
            if (row < nt) best.insert(K::make(hamming256(q0, q1, T, row), row));
        }
    }
    wg_scan_phase<4, 16 * NE, 64, K>(hs, best, wave, lane, T, nt, tiles_per_split);
    for (int c = 0; c < k; ++c) {
        const typename K::type m = K::wave_min(best.a);
        if (best.a == m && m != K::NONE) { best.a = best.b; best.b = K::NONE; }      // keys are unique rows
        if (lane == 0 && live) {
            pm_match mm;
            mm.queryIdx = q;
            mm.imgIdx = 0;
            if (m == K::NONE) { mm.trainIdx = -1; mm.distance = HM_INF; }
            else { mm.trainIdx = K::row(m); mm.distance = static_cast<float>(K::dist(m)); }
            out[static_cast<size_t>(q) * k + c] = mm;
        }
    }
}

// Round 3: FOUR queries per wave, one 16-lane row each (the form of knn_l2_refine8): at config C4 a query has 32 candidate
// entries, half a wave; the k-th-smallest search and the final top-k become 4-step DPP reductions inside a row (row_ror,
// no v_readlane), the instruction stream of a wave serves four queries.  Same entry layout, same tau / full-sub-list /
// candidate rules and therefore the same rows evaluated as knn_hamming_refine above; for slots <= 64 (NE = 1, 2 or 4
// entries per lane: entry e = l + 16*i).
__device__ __forceinline__ unsigned row16_min_u32(unsigned v)
{
    v = min(v, pm::dpp_u32<0x121>(v));      // row_ror:1
    v = min(v, pm::dpp_u32<0x122>(v));
    v = min(v, pm::dpp_u32<0x124>(v));
    v = min(v, pm::dpp_u32<0x128>(v));
    return v;
}
__device__ __forceinline__ unsigned long long row16_min_u64(unsigned long long v)
{
    const unsigned hi = static_cast<unsigned>(v >> 32), lo = static_cast<unsigned>(v);
    const unsigned mh = row16_min_u32(hi);
    const unsigned ml = row16_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return (static_cast<unsigned long long>(mh) << 32) | ml;
}
__device__ __forceinline__ unsigned row16_min(unsigned v) { return row16_min_u32(v); }
__device__ __forceinline__ unsigned long long row16_min(unsigned long long v) { return row16_min_u64(v); }

template <int NE, typename K>
__global__ __launch_bounds__(256) void knn_hamming_refine4(const uint32_t* __restrict__ Q, const uint32_t* __restrict__ T,
                                                           int nq, int nt, int k, const int* __restrict__ cand,
                                                           int slots, int tiles_per_split, int shift,
                                                           pm_match* __restrict__ out)
{
    __shared__ int clist[4][4][16 * NE];
    __shared__ __attribute__((aligned(16))) HmScan<16, 4 * NE, K> hs;      // (a query has slots / 4 <= 4 * NE sub-lists)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = lane >> 4, l = lane & 15;
    const int own = wave * 4 + qi;                           // this row's query inside the workgroup
    const int q = blockIdx.x * 16 + wave * 4 + qi;
    const bool live = q < nq;
    const int qc = live ? q : nq - 1;                        // rows past the last query shadow it (DPP rows stay whole)
    const uint4* qr = reinterpret_cast<const uint4*>(Q + static_cast<size_t>(qc) * 8);
    const uint4 q0 = qr[0], q1 = qr[1];
    const int gmask = (1 << shift) - 1;
    auto row_bits = [&](bool p) { return static_cast<unsigned>(__ballot(p) >> (16 * qi)) & 0xFFFFu; };

    int v[NE], dc[NE];
    bool whole[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = l + 16 * i;
        v[i] = e < slots ? cand[static_cast<size_t>(qc) * slots + e] : I8_EMPTY;
        dc[i] = v[i] == I8_EMPTY ? 0x7FFFFFF0 : ((I8_BITS - (v[i] >> shift)) >> 1);
        const int gid2 = v[i] & gmask, gid = gid2 >> 1, split = e >> 3;
        const int last = (split * tiles_per_split + (gid >> 3)) * H_TT + 32 * ((gid >> 1) & 3) + 16 * (gid & 1) +
                         4 * (gid2 & 1) + 11;
        whole[i] = v[i] != I8_EMPTY && last < nt;
    }
    unsigned lastk = 0u;
    int tau = 0;
    for (int c = 0; c < k; ++c) {
        unsigned m = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const unsigned dt = whole[i] ? static_cast<unsigned>(dc[i]) : 0x7FFFu;
            const unsigned key = (dt << 16) | static_cast<unsigned>(l + 16 * i);
            if ((c == 0 || key > lastk) && key < m) m = key;
        }
        m = row16_min_u32(m);
        lastk = m;
        tau = static_cast<int>(m >> 16);
    }
    if (tau == 0x7FFF) tau = 0x7FFFFFFF;
    Best2<typename K::type> best{K::NONE, K::NONE};
    int total = 0, nfull = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        if (16 * i >= slots) break;                          // wave-uniform
        const bool within = v[i] != I8_EMPTY && dc[i] <= tau;
        const unsigned full = row_bits(within && (l & 3) == 3);
        const bool my_full = (full >> (l | 3)) & 1u;
        const unsigned cm = row_bits(within && !my_full);
        if (within && !my_full) clist[wave][qi][total + __popc(cm & ((1u << l) - 1u))] = (v[i] & gmask) | (((l + 16 * i) >> 2) << 16);
        total += __popc(cm);
        // rare: a sub-list whose 4th entry is within tau is scanned whole — by the workgroup, below; recorded here
        if ((full >> l) & 1u) { if (live) hs.sub[own][nfull + __popc(full & ((1u << l) - 1u))] = (l + 16 * i) >> 2; }
        nfull += __popc(full);
    }
    if (l == 0) hs.nsub[own] = live ? nfull : 0;
    if (l == 0) { hs.qw[own][0] = q0; hs.qw[own][1] = q1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // candidate groups: 8 rows each, one row per lane, two groups per pass and query
    for (int base = 0; __any(base < 8 * total); base += 16) {
        const int t = base + l;
        if (t < 8 * total) {
            const int ent = clist[wave][qi][t >> 3];
            const int gid2 = ent & 0xFFFF, gid = gid2 >> 1, hh = gid2 & 1, split = ent >> 17;
            const int s8 = t & 7;
            const int row = (split * tiles_per_split + (gid >> 3)) * H_TT + 32 * ((gid >> 1) & 3) + 16 * (gid & 1) + 4 * hh +
                            8 * (s8 >> 2) + (s8 & 3);
            if (row < nt) best.insert(K::make(hamming256(q0, q1, T, row), row));
        }
    }
    wg_scan_phase<16, 4 * NE, 16, K>(hs, best, own, l, T, nt, tiles_per_split);
    for (int c = 0; c < k; ++c) {
        const typename K::type m = row16_min(best.a);
        if (best.a == m && m != K::NONE) { best.a = best.b; best.b = K::NONE; }      // keys are unique rows
        if (l == 0 && live) {
            pm_match mm;
            mm.queryIdx = q;
            mm.imgIdx = 0;
            if (m == K::NONE) { mm.trainIdx = -1; mm.distance = HM_INF; }
            else { mm.trainIdx = K::row(m); mm.distance = static_cast<float>(K::dist(m)); }
            out[static_cast<size_t>(q) * k + c] = mm;
        }
    }
}

// 256-bit descriptors, k <= 2
int run_mfma(pm_ctx* ctx, const uint32_t* dq, int nq, const uint32_t* dt, int nt, int k, pm_match* dout, bool wide_keys,
             bool* done)
{
    *done = false;
    const int nq_pad = (nq + H_QB - 1) / H_QB * H_QB, nt_pad = (nt + H_TT - 1) / H_TT * H_TT;
    const int nqb = nq_pad / H_QB, ntiles = nt_pad / H_TT;
    int splits = (2 * ctx->n_cu + nqb - 1) / nqb;
    if (splits > ntiles) splits = ntiles;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    const int tiles_per_split = (ntiles + splits - 1) / splits;
    splits = (ntiles + tiles_per_split - 1) / tiles_per_split;
    const int slots = splits * 2 * KNN_C;
    constexpr int shift = I8_SHIFT;
    if (tiles_per_split * (H_TT / 16) * 2 > (1 << shift)) return PM_OK; // > 64k (group, half) ids per split: VALU route

    const size_t cb = sizeof(int) * static_cast<size_t>(nq) * slots;
    const size_t qe = sizeof(uint4) * static_cast<size_t>(nq_pad) * I8_ROW16, te = sizeof(uint4) * static_cast<size_t>(nt_pad) * I8_ROW16;
    const size_t need = pm::align_up(cb, 256) + pm::align_up(qe, 256) + pm::align_up(te, 256) + 1024;
    int rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    int* cval = static_cast<int*>(pm::arena_take(ctx, cb));
    uint4* Qe = static_cast<uint4*>(pm::arena_take(ctx, qe));
    uint4* Te = static_cast<uint4*>(pm::arena_take(ctx, te));
    PM_REQUIRE(cval && Qe && Te, PM_E_NOMEM, "scratch arena too small");
    {
        pm::ScopedKernelTime t(ctx, "knn_hamming_expand");
        const int total = (nq_pad + nt_pad) * I8_NCH;
        hipLaunchKernelGGL(knn_hamming_expand, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, dq, nq, nq_pad, dt, nt,
                           nt_pad, Qe, Te);
        PM_HIP_CHECK(hipGetLastError());
    }
    rc = launch_coarse_i8(ctx, Qe, Te, nq, nq_pad, nt, splits, tiles_per_split, cval, slots);
    if (rc != PM_OK) return rc;
    {
        pm::ScopedKernelTime t(ctx, "knn_hamming_refine");
#define PM_HREFINE(NE_, K_)                                                                                      \
    hipLaunchKernelGGL((knn_hamming_refine<NE_, K_>), dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, dq, dt, nq, nt, k, \
                       cval, slots, tiles_per_split, shift, dout)
#define PM_HREFINE4(NE_, K_)                                                                                      \
    hipLaunchKernelGGL((knn_hamming_refine4<NE_, K_>), dim3((nq + 15) / 16), dim3(256), 0, ctx->stream, dq, dt, nq, nt, k, \
                       cval, slots, tiles_per_split, shift, dout)
    // Four queries per wave (round 3) where a query has <= 64 candidate entries; PM_OPT_HAMMING_REFINE = 1 keeps one wave
    // per query.  Config C4, refinement kernel: 31 us (round 2) -> 27 (batched scan loads) -> 22.3 (whole-sub-list scans
    // by the workgroup, one wave per query) / 20.8 us (four queries per wave); before the scans went to the workgroup the
    // four-queries form took 94-128 us there — a launch lasts as long as its slowest wave.
    const bool rows16 = slots <= 64 && ctx->opts[PM_OPT_HAMMING_REFINE] != 1;
#define PM_HREFINE_K(K_)                    \
    do {                                    \
        if (rows16 && slots <= 16) PM_HREFINE4(1, K_); \
        else if (rows16 && slots <= 32) PM_HREFINE4(2, K_); \
        else if (rows16) PM_HREFINE4(4, K_); \
        else if (slots <= 64) PM_HREFINE(1, K_); \
        else if (slots <= 128) PM_HREFINE(2, K_); \
        else if (slots <= 256) PM_HREFINE(4, K_); \
        else PM_HREFINE(HR_MAXE, K_);       \
    } while (0)
        if (nt < (1 << 23) && !wide_keys) PM_HREFINE_K(Key32);
        else PM_HREFINE_K(Key64);
#undef PM_HREFINE_K
#undef PM_HREFINE4
#undef PM_HREFINE
        PM_HIP_CHECK(hipGetLastError());
    }
    *done = true;
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
    // PM_OPT_HAMMING_ROUTE (tests / A-B timing): 1 pins the VALU scan, 2 the 64-bit-key refinement of the matrix-core route
    const bool force_valu = ctx->opts[PM_OPT_HAMMING_ROUTE] == 1;
    const bool wide_keys = ctx->opts[PM_OPT_HAMMING_ROUTE] == 2;
    if (bytes * 8 == I8_BITS && k <= 2 && nt >= 1 && !force_valu &&
        (reinterpret_cast<uintptr_t>(dq) & 15) == 0 && (reinterpret_cast<uintptr_t>(dt) & 15) == 0) {
        bool done = false;
        const int rc = run_mfma(ctx, q32, nq, t32, nt, k, dout, wide_keys, &done);
        if (rc != PM_OK || done) return rc;
    }
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
