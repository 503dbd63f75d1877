// Epoch refresh of the resident training set in ONE kernel: the row gather (reshuffle) fused with the Gram
// blocks of consecutive mini-batches.
//
//   dst[r][:]        = src[perm[r]][:]                       r < nb * B        (mnist.train.next_batch's
//                                                                               reshuffle, tfdist_between.py:92)
//   gram[j][m][i]    = 1 + dst[(j+1) B + m] . dst[j B + i]   j < nb - 1, m, i < B; zero for B <= i < ld
//
// The Gram blocks are what lets a fused training step run as one launch (step_pipe_kernel: the forward of
// batch j+1 is computed from the parameters before the update of batch j plus a correction through block j).
// They cost as many FLOPs as the epoch's forward passes (549 x 100 x 100 x 784 x 2 = 8.6 GFLOP), all of them on
// the fp32 matrix cores here (v_mfma_f32_16x16x4_f32: exact fp32 products and sums, which the forward pass
// relies on), with the gather riding along: the rows a workgroup has loaded for its products are also the rows
// it stores into the epoch-ordered copy, so the set is read once through the permutation instead of once by a
// gather kernel and again by a batched GEMM.
//
// Work split.  A pair (j, j+1) is a [112 x 112 x K] product (B = 100 padded to 7 x 7 MFMA blocks): 7 block rows of 16
// Gram rows, "units" of 1,372 MFMAs, one wave each -- the wave accumulates its block row against all 7 block columns
// (28 accumulator registers), and a 16-k macro-step costs it 1 + 7 LDS reads of 16 bytes for 28 MFMAs.  549 pairs
// are 3,843 units; the chip has 1,024 SIMDs with room for 4 such waves each, so ALL units are resident at once:
// workgroup w of 256 (16 waves, one workgroup per CU) takes units [w U / 256, (w + 1) U / 256) -- 15 or 16 of them
// -- and every SIMD ends at the same time.  (Two workgroups of four waves per pair, the earlier split, ran 1,098
// workgroups of unequal size through 768 slots: a ragged second round, and every row read three times.)
// Consecutive units belong to at most 4 consecutive pairs, i.e. 5 consecutive batches, and a batch that is x_{j+1}
// of one pair is x_j of the next: the workgroup stages each of its batches ONCE per chunk, as an image
// [k-quad][112 rows][4 floats], and every wave reads its 16 A rows from one image and the 7 x 16 B rows from the
// image before it.  Rows are read ~1.6x (the batches at a workgroup's ends are also its neighbours', and neighbours
// share an XCD and its L2).
//
// Staging.  K is walked in chunks of 64 floats (16 k-quads).  A wave-instruction loads 4 rows x 16 k-quads (256
// contiguous bytes of each row) with lane -> (row l & 3, k-quad l >> 2); 25 row groups per batch, at most 125 of
// them per workgroup and chunk = 8 loads per thread.  The image holds row r of k-quad kq at row slot r ^ 4 (kq & 1) of
// the k-quad's 112: a ds_write_b128 is served in groups of 8 contiguous lanes (4 rows of 2 consecutive k-quads, banks
// modulo 32 dwords), and the flipped row bit puts the odd k-quad's rows into the other half of the 128 bytes.  A
// ds_read_b128 is served in four groups of 16 NON-contiguous lanes ({0-3, 12-15, 20-27}, ...: rows 0-3 and 12-15 of
// one lane quarter with rows 4-11 of the next, banks modulo 64 dwords): lane (i = l & 15, q = l >> 4) of a macro-step
// reads k-quad 4 ms + {0, 2, 1, 3}[q] of row i, so that the two quarters of a group read k-quads of the same parity
// and meet 16 different 16-byte slots (rows of 116 with the plain assignment measured 6.0 M conflict cycles of 15.5 M
// LDS cycles per launch).  MFMA s of the macro-step contracts k = 16 ms + 4 {0, 2, 1, 3}[q] + s
// (component s of both operands).  Rows B..111 of an image are never written (but for the zeros that unused load
// slots drop into rows 104-111): Gram element (m, i) depends on A row m and B row i alone, and rows or columns
// beyond B are not stored.  One LDS buffer and one register set: the loads
// of chunk c+1 are issued while chunk c is multiplied, into the registers whose rows of chunk c have just been
// stored.  The row that a thread loads is also the row it stores into the epoch-ordered copy if the row's unit (as
// x_{j+1}) is one of the workgroup's (batch 0: the workgroup that owns unit 0).
#include "dtf_api.h"
#include "dtf_common.h"

namespace {

constexpr int kChunkQ = 16;                 // k-quads per chunk (64 floats)
constexpr int kRowsB = 112;                 // Gram columns with padding (7 block columns): the largest leading dimension
constexpr int kPThreads = 1024;
constexpr int kPWaves = kPThreads / 64;
constexpr int kPRows = 112;                 // rows per k-quad of a batch image (7 MFMA blocks; 448 dwords = 0 modulo 64 banks)
constexpr int kPMaxBat = 5;                 // batches that 16 consecutive units can touch
constexpr int kPBat = kChunkQ * kPRows * 4; // floats per batch image (29,696 bytes)
constexpr int kPLoads = 8;                  // wave-instructions per wave and chunk
constexpr int kBlk = 7;                     // MFMA block rows (and columns) of a pair
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GramArgs {
  const float* src;        // [rows_src][K] master copy
  const int64_t* perm;     // [>= nb * B] epoch order
  float* dst;              // [>= nb * B][K] epoch-ordered copy (written here); null: Gram blocks only
  float* gram;             // [nb - 1][B][ld]
  int B, K, nb, ld;
  int rows_src;
  const float* ysrc;       // optional: [rows_src][C] labels, gathered into ydst[r] = ysrc[perm[r]] for r < nb * B
  float* ydst;
  int C;
  int parts;               // timing experiments: bit 0 global loads per chunk, bit 1 products, bit 2 row stores (7: all)
  long long* stamps;       // timing experiments: [workgroup][24] wall-clock stamps (100 MHz) of wave 0, or null
};

__global__ __launch_bounds__(kPThreads) void gather_gram_kernel(GramArgs a) {
  __shared__ __attribute__((aligned(16))) float img[kPMaxBat * kPBat];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  // blocks b and b + 8 share an XCD under round-robin dispatch: an XCD gets a contiguous range of workgroups, whose
  // end batches are the neighbours' too and meet in its L2
  const int per_x = (int)gridDim.x >> 3;
  const int wg = (int)(blockIdx.x & 7) * per_x + (int)(blockIdx.x >> 3);
  const long U = (long)(a.nb - 1) * kBlk;
  const int u0 = (int)(wg * U / (long)gridDim.x), u1 = (int)((wg + 1) * U / (long)gridDim.x);
  if (u0 >= u1) return;
  const int b0 = u0 / kBlk, b1 = (u1 - 1) / kBlk + 1;         // batches b0 .. b1 (b1: as x_{j+1} only)
  const int nbat = b1 - b0 + 1;
  const int B = a.B, K = a.K, nkq = K >> 2;
  const int ngroups = (B + 3) >> 2;
  const int r4 = lane & 3, kql = lane >> 2;
  const unsigned rowbytes = (unsigned)K * 4u;
#define DTF_GRAM_STAMP(i) \
  if (a.stamps != nullptr && tid == 0) a.stamps[(long)wg * 24 + (i)] = wall_clock64()
  DTF_GRAM_STAMP(0);
#define DTF_GRAM_OWNS(batch, blockrow) \
  (((batch) >= 1 && ((batch) - 1) * kBlk + (blockrow) >= u0 && ((batch) - 1) * kBlk + (blockrow) < u1) || ((batch) == 0 && u0 == 0))

  // ---- the rows this thread loads: item = wave + 16 u -> (batch slot, group of 4 rows); wave-uniform -------------
  // Everything a chunk does with memory is branch-free, so that its loads, LDS writes and stores sit in straight-line
  // code and the compiler counts the outstanding memory operations exactly (with a branch around every load and
  // store it waited for vmcnt(0) before each of them: sixteen memory round trips in a row per chunk).  Loads and
  // stores go through buffer descriptors and an item that is not wanted gets an offset beyond the buffer: the range
  // check drops the store and returns zeros for the load (also for k-quads beyond K in the last chunk).
  const unsigned src_bytes = (unsigned)a.rows_src * rowbytes, dst_bytes = a.dst != nullptr ? (unsigned)(a.nb * B) * rowbytes : 0u;
  const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src), (short)0, (int)src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.dst, (short)0, (int)((a.parts & 4) ? dst_bytes : 0u), 0x00020000);
  constexpr unsigned kBeyond = 0xfffffff0u;                   // (+ 12 for the last of four dwords does not wrap)
  unsigned srcoff[kPLoads];                                   // byte offset of the source row; kBeyond: no load
  int ldsoff[kPLoads];                                        // wave-uniform: the group's first row in img, in units of 16 floats
  unsigned dstrow[kPLoads];                                   // wave-uniform: epoch row of the group's first row
  unsigned own = 0;                                           // bit u: the rows of item u are stored by this workgroup
#pragma unroll
  for (int u = 0; u < kPLoads; ++u) {
    const int it = wave + kPWaves * u;
    const int bs = it / ngroups, rg = it - bs * ngroups;
    const int batch = b0 + bs;
    const bool owned = bs < nbat && DTF_GRAM_OWNS(batch, rg >> 2);
    const bool needed = bs < nbat && (batch < b1 || owned);   // every row of a batch that is some pair's x_j here
    own |= (owned ? 1u : 0u) << u;
    // an item beyond the workgroup's batches writes its zeros into rows that no product keeps (group 27: rows 104 - 111)
    ldsoff[u] = bs < nbat ? bs * (kPBat / 16) + rg : 27;
    dstrow[u] = (unsigned)(batch * B + 4 * rg);
    const int row = 4 * rg + r4;
    const long erow = needed ? (long)batch * B + (row < B ? row : 0) : 0;
    long srow = a.perm[erow];
    srow = srow < 0 ? 0 : (srow >= a.rows_src ? a.rows_src - 1 : srow);            // never outside the master copy
    srcoff[u] = (needed && row < B) ? (unsigned)(srow * (long)rowbytes) : kBeyond;
  }

  const int nchunks = (nkq + kChunkQ - 1) / kChunkQ;
  const int nkq_ld = (a.parts & 1) ? nkq : 0;
  f32x4 stage[kPLoads];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < kPLoads; ++u) {
    const unsigned off = (kql < nkq_ld && srcoff[u] != kBeyond) ? srcoff[u] + (unsigned)kql * 16u : kBeyond;
    stage[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src_rsrc, off, 0, 0));
  }

  f32x4 acc[kBlk];
#pragma unroll
  for (int bi = 0; bi < kBlk; ++bi) acc[bi] = zero4;
  const bool wave_works = u0 + wave < u1;
  const int unit = wave_works ? u0 + wave : u0;
  const int pair = unit / kBlk, brow = unit - pair * kBlk;
  // (the image's swizzle and the k-quad a lane group takes: see "Staging" at the top)
  const int i16x = i16 ^ ((q >> 1) << 2), kqsel = ((q & 1) << 1) | (q >> 1);
  const float* As = img + (pair + 1 - b0) * kPBat + (kqsel * kPRows + 16 * brow + i16x) * 4;
  const float* Bs = img + (pair - b0) * kPBat + (kqsel * kPRows + i16x) * 4;

  // the labels of the rows this workgroup stores travel with them, behind the first chunk's loads: a wave takes the
  // 16 rows of its own unit, four elements per lane at a time so that the loads of a turn are in flight together
  if (a.ydst != nullptr) {
    const int C = a.C;
    if (wave_works) {
      const long erow0 = (long)(pair + 1) * B + 16 * brow;
      const int n = (B - 16 * brow < 16 ? B - 16 * brow : 16) * C;
      for (int e0 = 0; e0 < n; e0 += 256) {
        long srow[4];
        int row[4];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = e0 + lane + 64 * j;
          row[j] = e < n ? e / C : 0;
          srow[j] = a.perm[erow0 + row[j]];
          srow[j] = srow[j] < 0 ? 0 : (srow[j] >= a.rows_src ? a.rows_src - 1 : srow[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = e0 + lane + 64 * j;
          v[j] = a.ysrc[srow[j] * C + (e < n ? e - row[j] * C : 0)];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = e0 + lane + 64 * j;
          if (e < n) a.ydst[(erow0 + row[j]) * C + (e - row[j] * C)] = v[j];
        }
      }
    }
    if (u0 == 0)                                        // batch 0 is nobody's x_{j+1}
      for (int e = tid; e < B * C; e += kPThreads) {
        long srow = a.perm[e / C];
        srow = srow < 0 ? 0 : (srow >= a.rows_src ? a.rows_src - 1 : srow);
        a.ydst[e] = a.ysrc[srow * C + e % C];
      }
  }

  DTF_GRAM_STAMP(1);
  for (int c = 0; c < nchunks; ++c) {
    const int kq = c * kChunkQ + kql;
    __syncthreads();                                  // the previous chunk has been multiplied
    if (c < 20) DTF_GRAM_STAMP(2 + c);
#pragma unroll
    for (int u = 0; u < kPLoads; ++u)
      *reinterpret_cast<f32x4*>(img + 16 * (ldsoff[u] ^ (kql & 1)) + (kql * kPRows + r4) * 4) = stage[u];
    __syncthreads();
    // The epoch-ordered copy is stored from the staging registers AFTER the barrier (no wave waits at the barrier for
    // another wave's global stores; streaming stores: nobody reads the copy before the kernel ends), and the loads
    // of the next chunk follow into the same registers (beyond K: out of range, no traffic).  All 16 waves leave the
    // barrier together: were they to issue their 16 memory instructions there, the matrix pipes would idle until the
    // address unit has taken 256 KB of them.  A full chunk therefore starts with the products of macro-step 0 and
    // issues half of its memory instructions before macro-step 1 and half before macro-step 2, by which time the
    // waves of a SIMD are staggered by the shared matrix pipe and one wave's memory instructions overlap the
    // others' products (136 -> 128 us).
    const unsigned lane_off = (unsigned)r4 * rowbytes + (unsigned)kq * 16u;
    const int kq_next = kq + kChunkQ;
#define DTF_GRAM_MEM(U0, U1)                                                                                       \
  _Pragma("unroll") for (int u = (U0); u < (U1); ++u) {                                                           \
    const bool st_ = (own >> u & 1u) && kq < nkq && srcoff[u] != kBeyond; /* srcoff: the row exists (< B) */      \
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, stage[u]), dst_rsrc,                         \
                                           st_ ? dstrow[u] * rowbytes + lane_off : kBeyond, 0, 2);                \
  }                                                                                                                \
  _Pragma("unroll") for (int u = (U0); u < (U1); ++u) {                                                           \
    const unsigned off_ = (kq_next < nkq_ld && srcoff[u] != kBeyond) ? srcoff[u] + (unsigned)kq_next * 16u : kBeyond; \
    stage[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src_rsrc, off_, 0, 0));            \
  }
    // B operands in two turns (4 + 3 blocks): 12 registers fewer than all seven at once
#define DTF_GRAM_MACRO(ms)                                                                                         \
  do {                                                                                                             \
    const f32x4 av = *reinterpret_cast<const f32x4*>(As + 4 * (ms) * kPRows * 4);                            \
    f32x4 bv[4];                                                                                                   \
    _Pragma("unroll") for (int bi = 0; bi < 4; ++bi)                                                               \
        bv[bi] = *reinterpret_cast<const f32x4*>(Bs + (4 * (ms) * kPRows + 16 * bi) * 4);                    \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) _Pragma("unroll") for (int bi = 0; bi < 4; ++bi)                 \
        acc[bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[bi][s], acc[bi], 0, 0, 0);                        \
    _Pragma("unroll") for (int bi = 0; bi < 3; ++bi)                                                               \
        bv[bi] = *reinterpret_cast<const f32x4*>(Bs + (4 * (ms) * kPRows + 16 * (4 + bi)) * 4);              \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) _Pragma("unroll") for (int bi = 0; bi < 3; ++bi)                 \
        acc[4 + bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[bi][s], acc[4 + bi], 0, 0, 0);                \
  } while (0)
    const int left = nkq - c * kChunkQ;
    const int nms = !(wave_works && (a.parts & 2)) ? 0 : (left >= kChunkQ ? 4 : (left + 3) >> 2);
    if (nms < 4) {
      DTF_GRAM_MEM(0, kPLoads)
    }
#pragma unroll 1
    for (int ms = 0; ms < nms; ++ms) {              // (rolled: unrolled, the macro-steps overlap and spill)
      if (nms == 4 && ms == 1) {
        DTF_GRAM_MEM(0, kPLoads / 2)
      }
      if (nms == 4 && ms == 2) {
        DTF_GRAM_MEM(kPLoads / 2, kPLoads)
      }
      DTF_GRAM_MACRO(ms);
    }
  }

  DTF_GRAM_STAMP(22);
  // ---- epilogue: D[row = 4 q + reg][col = i16] of block (brow, bi) -> gram[pair][m][i] ---------------------------
  if (wave_works) {
    float* g = a.gram + (size_t)pair * B * a.ld;
#pragma unroll
    for (int bi = 0; bi < kBlk; ++bi) {
      const int i = 16 * bi + i16;
      if (i >= a.ld) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * brow + 4 * q + r;
        if (m < B) g[(size_t)m * a.ld + i] = i < B ? 1.0f + acc[bi][r] : 0.f;
      }
    }
  }
  DTF_GRAM_STAMP(23);
}

int g_gram_parts = 7;
long long* g_gram_stamps = nullptr;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int dtf_gather_gram_supported(int B, int K) {
  return (B >= 1 && B <= 100 + 4 && K >= 4 && (K & 3) == 0) ? 1 : 0;
}

// dst[r] = src[perm[r]] for r < nb * B and gram[j][m][i] = 1 + dst[(j+1) B + m] . dst[j B + i] (zero for
// B <= i < ld) for j < nb - 1, in one launch.  ld: multiple of 16 with B <= ld <= 112; every buffer 16-byte
// aligned; rows_src * K * 4 and nb * B * K * 4 must fit 32 bits.  ysrc / ydst (both or neither): [rows][C] labels
// gathered the same way, ydst[r] = ysrc[perm[r]] for r < nb * B.
extern "C" int dtf_gather_gram(const float* src, const int64_t* perm, float* dst, float* gram, int rows_src, int B,
                               int K, int nb, int ld, const float* ysrc, float* ydst, int C, void* stream) {
  if (!src || !perm || !gram) return DTF_ERR_NULL;      // dst may be null: Gram blocks of src[perm] only
  if (!dtf_gather_gram_supported(B, K) || nb < 2 || ld < B || ld > kRowsB || (ld & 15) || rows_src < 1)
    return DTF_ERR_SHAPE;
  if (!aligned16(src) || (dst && !aligned16(dst)) || !aligned16(gram)) return DTF_ERR_SHAPE;
  // byte offsets are 32-bit, and 0xfffffff0 means "no access": a row's offset plus an offset inside a row must stay below it
  if (((long)rows_src + 1) * K * 4 >= 0xfffffff0L || ((long)nb * B + 1) * K * 4 >= 0xfffffff0L) return DTF_ERR_SHAPE;
  if ((ysrc == nullptr) != (ydst == nullptr) || (ydst && C < 1)) return DTF_ERR_NULL;
  GramArgs a{src, perm, dst, gram, B, K, nb, ld, rows_src, ysrc, ydst, C, g_gram_parts, g_gram_stamps};
  // a workgroup's units: at most 16 (one per wave), and few enough that its batches' row groups fit 8 loads a thread
  // (16 units touch at most 5 batches of 25 row groups, 15 units at most 4 batches of 26)
  const long units = (long)(nb - 1) * kBlk;
  const int max_units = B <= 100 ? kPWaves : kPWaves - 1;
  long wgs = units >= 256 * 8 ? 256 : (units + 7) / 8;
  if (wgs * max_units < units) wgs = (units + max_units - 1) / max_units;
  const int blocks = (int)((wgs + 7) / 8 * 8);
  hipLaunchKernelGGL(gather_gram_kernel, dim3(blocks), dim3(kPThreads), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

// Workgroups of the kernel that one CU holds at a time, as the runtime computes it (registers, LDS, wave slots).
extern "C" int dtf_debug_gram_blocks_per_cu(void) {
  int n = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gather_gram_kernel, kPThreads, 0) != hipSuccess) return -1;
  return n;
}

// Timing experiments: run only some parts of the kernel (results are then wrong); 7 restores the full kernel.
extern "C" void dtf_debug_set_gram_parts(int parts) { g_gram_parts = parts; }
// Timing experiments: wave 0 of workgroup w writes wall-clock stamps to stamps[w][24] (0 start, 1 before the chunk loop,
// 2 + c after the first barrier of chunk c, 22 after the loop, 23 at the end); null switches them off.
extern "C" void dtf_debug_set_gram_stamps(void* stamps) { g_gram_stamps = static_cast<long long*>(stamps); }
