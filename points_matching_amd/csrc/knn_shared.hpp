// knn_shared.hpp — constants and entry points shared by the two translation units of the L2
// matcher: knn_l2.hip (prep, refinement, exact kernel, C ABI) and knn_coarse.hip (the two MFMA
// coarse kernels, built with -ffinite-math-only so that the selection can use the plain
// max/med3 builtins: no canonicalising v_max per operand, and MFMA->VALU read hazards stay under
// the compiler's hazard recogniser instead of hand-placed inline asm).
#pragma once
#include "pm_common.hpp"

namespace pm_knn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int KNN_C = 4;          // coarse candidates kept per (query, split, lane-half)
constexpr int QB = 128;           // queries per workgroup, f32 route (4 waves x 32)
constexpr int TT32 = 64;          // train rows per LDS tile, f32 route
constexpr float KNN_BIG = 3.0e38f;   // finite sentinel: stays finite under the id insert

// f16 route (integer-valued descriptors)
constexpr int H_DP = 128;               // data columns
constexpr int H_ROW = H_DP + 16;        // halfs per global row (288 B): data + seed chunk
constexpr int H_LDS_ROW = H_ROW + 8;    // halfs per LDS row (304 B: 16 rows of a lane group hit 16 slots)
constexpr int H_NCH = H_ROW / 16;       // 9 k-chunks of 16
constexpr int H_TT = 128;               // train rows per tile
constexpr int H_QB = 256;               // queries per workgroup (4 waves x 64)
constexpr float H_MAXABS = 361.f;       // 128 * 361^2 < 2^24
constexpr int H_ROW16 = H_ROW / 8;      // 16-byte units per global row (18)
constexpr int H_LDS_ROW16 = H_LDS_ROW / 8;   // 16-byte units per LDS row (19)

// i8 route (256-bit binary descriptors expanded to +-1 bytes): 256-byte rows = 8 chunks of 32 bytes,
// no seed chunk (rows padding the last tile are all-zero: dot = 0, the refinement knows them by
// their index); row groups of 8 rows
constexpr int I8_BITS = 256;
constexpr int I8_NCH = 8;
constexpr int I8_ROW16 = 16;
constexpr int I8_LDS_ROW16 = 17;        // 272-byte LDS rows: 16 rows of a lane group hit 16 different slots
constexpr int I8_GROUP_ROWS = 8;
constexpr int I8_SHIFT = 16;             // candidate = (dot << 16) | group id, |dot| <= 256
constexpr int I8_EMPTY = static_cast<int>(0x80000000u);   // an unfilled list entry

// Seeded routes (round 3): the per-row term -||t||^2/2 no longer rides a k-chunk of its own through the matrix pipe; it
// STARTS the accumulators.  A 32-row block's 32 seeds are kept in the order of the 32x32 C/D layout ("seed order":
// position 16*h + reg <-> row (reg&3) + 8*(reg>>2) + 4*h of the block), so a lane's 16 C-in registers are four
// ds_read_b128 of a 512-byte per-tile array that is staged by one more LDS-DMA piece.  Every issued MFMA is then
// algorithmic work (2*D flop per pair).
//   u8 route   u8-valued descriptors (OpenCV SIFT: 0..255) centred to x - 128 and ranked on v_mfma_i32_32x32x32_i8:
//              128-byte rows = 4 k-chunks, exact integers; seed = -(||t - 128||^2 >> 1), so the coarse squared
//              distance ||q'||^2 - 2w is d2 or d2 - 1 (the refinement's window carries the unit);
//   f16s route integer-valued descriptors with |x| <= 361 on v_mfma_f32_32x32x16_f16: 256-byte rows = 8 chunks.
constexpr int U8_DP = 128;              // data columns (bytes) per row
constexpr int U8_NCH = 4;               // k-chunks of 32 bytes
constexpr int U8_ROW16 = 8;             // 16-byte units per global row
constexpr int U8_LDS_ROW16 = 9;         // 144-byte LDS rows: 16 rows of a lane group hit 16 different 16-byte slots
constexpr int U8_SHIFT = 9;             // candidate = (w << 9) | (group id << 1 | lane half); |w| < 2^22
constexpr int U8_PAD_SEED = -(1 << 22); // seed of the rows padding the last tile: below every real w (>= -3.13e6)
constexpr int F16S_NCH = 8;
constexpr int F16S_ROW16 = 16;
constexpr int F16S_LDS_ROW16 = 17;
constexpr int SEED_TILE_BYTES = H_TT * 4;       // one tile's seeds (128 x 4 B), seed order inside each 32-row block

// position of train row `row` (global index) in the seed array
__host__ __device__ inline int seed_pos(int row)
{
    const int i = row & 31;
    return (row & ~31) + 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3);
}

// "wide" train rows of the register-operand coarse form with a split per wave (knn_u8_rega<.., WSPLIT>): 144 bytes = the
// LDS image of a row, so a 32-row block is 4.5 KiB of contiguous memory that LDS-DMA copies as it is; the 16-byte pad slot
// of row j < 8 of a block holds the block's seeds at seed-order positions 4j .. 4j+3 (a lane half's 16 C-in values are
// the pad slots of rows 4h .. 4h+3: four ds_read_b128).  Pad slots of rows 8 .. 31 are unused.
constexpr int U8_WIDE_ROW16 = 9;
__host__ __device__ inline size_t u8_wide_seed_index(int row)           // index, in ints, into the wide train copy
{
    const int p = seed_pos(row) - (row & ~31);
    return (static_cast<size_t>(row & ~31) + (p >> 2)) * (U8_WIDE_ROW16 * 4) + U8_ROW16 * 4 + (p & 3);
}

// Enqueue the seeded coarse passes.  Q8/T8: nq_pad x 128 / nt_pad x 128 centred bytes; Qh/Th: n_pad x 128 halfs;
// seeds: nt_pad (+ H_TT slack) 4-byte seeds in seed order.  Candidates: u8 route int (w << U8_SHIFT) | id, f16s
// route float with the id in the low mantissa bits (keep_mask as on the f16 route).
// group_rows: rows per candidate group (4, 8 or 16); form: 0 / 1 two LDS tile buffers, 2 / 3 ring of 8 LDS tile buffers
// with counted waits, 4 / 5 / 6 the register-operand forms (grid and splits sized for 128 queries per workgroup; 6: one
// split per WAVE, grid.y = ceil(splits / 8)).
int launch_coarse_u8(pm_ctx* ctx, const void* Q8, const void* T8, const int* seeds, int nq, int nq_pad, int nt, int splits,
                     int tiles_per_split, int* cval, int slots, int group_rows, int form);
int launch_coarse_f16s(pm_ctx* ctx, const _Float16* Qh, const _Float16* Th, const float* seeds, int nq, int nq_pad, int nt,
                       int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots);

// Enqueue the f32-MFMA coarse pass (dim % 4 == 0, dim <= 128).  only_if_ineligible != 0: the
// kernel runs only when prep16 flagged the data as not f16-eligible (auto route).
int launch_coarse_f32(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim, const float* tnorm,
                      int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots,
                      const unsigned long long* stats, unsigned epoch, int only_if_ineligible);
// Enqueue the exact f16-MFMA coarse pass on the padded f16 copies.  mode 1: run only if eligible.
// dp: padded data columns of the copies, 128 or 256.
int launch_coarse_f16(pm_ctx* ctx, const _Float16* Qh, const _Float16* Th, int nq, int nq_pad, int nt, int splits,
                      int tiles_per_split, unsigned keep_mask, float* cval, int slots,
                      const unsigned long long* stats, unsigned epoch, int mode, int dp = 128);

// Enqueue the i8-MFMA coarse pass of the Hamming matcher on the expanded +-1 copies.  A candidate
// is (dot << I8_SHIFT) | group id, dot = 256 - 2*hamming.
int launch_coarse_i8(pm_ctx* ctx, const void* Qe, const void* Te, int nq, int nq_pad, int nt, int splits,
                     int tiles_per_split, int* cval, int slots);

}  // namespace pm_knn
