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
// Work split.  A pair (j, j+1) is a [112 x 112 x K] product (B = 100 padded to 7 x 7 MFMA blocks).  Two
// workgroups of 4 waves share a pair: half 0 owns block rows 0-3 (m < 64), half 1 block rows 4-6; wave w of a
// workgroup accumulates ONE block row against all 7 block columns (28 accumulator registers), so a 16-k
// macro-step costs a wave 1 + 7 LDS reads of 16 bytes for 28 MFMAs.  1,098 workgroups of unequal size (4 : 3)
// spread over 256 CUs at up to 3 workgroups per CU; the two halves of a pair get block ids 8 apart, i.e. the
// same XCD under round-robin dispatch, so the rows of batch j they both read meet in one L2.
//
// Staging.  K is walked in chunks of 64 floats (16 k-quads).  A wave-instruction loads 8 rows x 8 k-quads:
// lanes 0-7 take the same k-quad of 8 consecutive rows, so every touched 128-byte line is used whole and the
// eight lanes of an LDS store group write 128 contiguous bytes.  The LDS image is [k-quad][row][4 floats] with
// 64 / 112 rows per k-quad: lane (i = l & 15, q = l >> 4) of a macro-step reads k-quad 4 ms + q of row i as one
// ds_read_b128, and the 16 lanes of every ds_read_b128 lane group cover 16 different rows of (at most two)
// k-quads whose row counts are multiples of 16 -- 16 distinct 16-byte slots, no bank conflict.  MFMA s of the
// macro-step contracts k = 16 ms + 4 q + s (component s of both operands).  One LDS buffer and one register set:
// the loads of chunk c+1 are issued right after chunk c was written to LDS and land while chunk c is multiplied.
#include "dtf_api.h"
#include "dtf_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunkQ = 16;                 // k-quads per chunk (64 floats)
constexpr int kRowsA = 64;                  // rows of batch j+1 per workgroup (4 block rows)
constexpr int kRowsB = 112;                 // rows of batch j in the LDS image (7 block columns)
constexpr int kRowGroups = (kRowsA + 104) / 8;   // 8 groups of A rows + 13 groups of B rows (104 >= 100)
constexpr int kLoads = (kRowGroups * 2 + 3) / 4; // wave-instructions per wave and chunk (11)

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
};

__global__ __launch_bounds__(kThreads, 3) void gather_gram_kernel(GramArgs a) {
  __shared__ __attribute__((aligned(16))) float As[kChunkQ * kRowsA * 4];
  __shared__ __attribute__((aligned(16))) float Bs[kChunkQ * kRowsB * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  // Blocks b and b + 8 share an XCD under round-robin dispatch.  An XCD walks a CONTIGUOUS range of pairs in
  // dispatch order, both halves of a pair back to back: the rows of batch j are wanted by pair j - 1 (as x_{j+1})
  // and by both halves of pair j (as x_j), and workgroups that run side by side on one XCD find them in its L2.
  const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int npairs = a.nb - 1;
  const int per_xcd = (npairs + 7) >> 3;
  const int pair = xcd * per_xcd + (slot >> 1), half = slot & 1;
  if ((slot >> 1) >= per_xcd || pair >= npairs) return;
  const int B = a.B, K = a.K, nkq = K >> 2;
  const int m_base = half * kRowsA;                   // first batch row of this workgroup's A rows

  // ---- the rows this thread loads: row group rg = 2 u + (wave >> 1), k-quad group kg = wave & 1 -----------
  // (rg, and with it everything but the row-within-group term, is wave-uniform: only the source offsets, which
  // come through the permutation, are kept per load in vector registers)
  const int r8 = lane & 7, kql = lane >> 3;           // row within the group, k-quad within the group
  const int kg = wave & 1, rg0 = wave >> 1;
  const unsigned rowbytes = (unsigned)K * 4u;
  const unsigned dstA = (unsigned)(((long)(pair + 1) * B + m_base + r8) * (long)rowbytes);   // + 8 rg rows
  const unsigned dstB = (unsigned)(((long)pair * B + r8) * (long)rowbytes);                  // + 8 (rg - 8) rows
  const unsigned ldsA = (unsigned)(((kql + 8 * kg) * kRowsA + r8) * 4);                      // + 32 rg floats
  const unsigned ldsB = (unsigned)(((kql + 8 * kg) * kRowsB + r8) * 4);                      // + 32 (rg - 8) floats
  const bool store_b = pair == 0 && half == 0;        // batch 0 is nobody's "j + 1": pair 0 stores it
  unsigned srcoff[kLoads];      // byte offset of the source row (through the permutation)
#pragma unroll
  for (int u = 0; u < kLoads; ++u) {
    const int rg = 2 * u + rg0;
    const bool isA = rg < kRowsA / 8;
    const int brow = isA ? m_base + 8 * rg + r8 : 8 * (rg - kRowsA / 8) + r8;     // row inside its batch
    const bool ok = rg < kRowGroups && brow < B;
    const long erow = (long)(pair + (isA ? 1 : 0)) * B + (ok ? brow : 0);         // row of the epoch order
    long srow = a.perm[erow];
    srow = srow < 0 ? 0 : (srow >= a.rows_src ? a.rows_src - 1 : srow);            // never outside the master copy
    srcoff[u] = (unsigned)(srow * (long)rowbytes);
  }
  // rows 104-111 of the B image are never loaded: zero them once (rows 100-103 are written as zeros per chunk)
  for (int e = tid; e < kChunkQ * 8; e += kThreads) {
    const int kq = e >> 3, r = 104 + (e & 7);
    *reinterpret_cast<f32x4*>(Bs + (kq * kRowsB + r) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // the labels of the rows this workgroup stores travel with them (one element per thread and turn)
  if (a.ydst != nullptr) {
    const int C = a.C;
    const int rowsA = B - m_base < kRowsA ? B - m_base : kRowsA;
    for (int e = tid; e < rowsA * C; e += kThreads) {
      const long erow = (long)(pair + 1) * B + m_base + e / C;
      long srow = a.perm[erow];
      srow = srow < 0 ? 0 : (srow >= a.rows_src ? a.rows_src - 1 : srow);
      a.ydst[erow * C + e % C] = a.ysrc[srow * C + e % C];
    }
    if (store_b)
      for (int e = tid; e < B * C; e += kThreads) {
        long srow = a.perm[e / C];
        srow = srow < 0 ? 0 : (srow >= a.rows_src ? a.rows_src - 1 : srow);
        a.ydst[e] = a.ysrc[srow * C + e % C];
      }
  }

  const char* src = reinterpret_cast<const char*>(a.src);
  char* dst = reinterpret_cast<char*>(a.dst);
  const int nchunks = (nkq + kChunkQ - 1) / kChunkQ;
  f32x4 stage[kLoads];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define DTF_GRAM_ISSUE(c)                                                                           \
  do {                                                                                              \
    const int kq_ = (c) * kChunkQ + 8 * kg + kql;                                                   \
    const unsigned koff_ = (unsigned)((kq_ < nkq ? kq_ : 0) * 16);                                  \
    _Pragma("unroll") for (int u = 0; u < kLoads; ++u)                                              \
        stage[u] = *reinterpret_cast<const f32x4*>(src + srcoff[u] + koff_);                        \
  } while (0)
  DTF_GRAM_ISSUE(0);

  f32x4 acc[7];
#pragma unroll
  for (int bi = 0; bi < 7; ++bi) acc[bi] = zero4;
  const bool wave_works = m_base + 16 * wave < B;     // half 1: block row 7 does not exist

  for (int c = 0; c < nchunks; ++c) {
    const int kq = c * kChunkQ + 8 * kg + kql;
    const bool kq_ok = kq < nkq;
    __syncthreads();                                  // the previous chunk has been multiplied
#pragma unroll
    for (int u = 0; u < kLoads; ++u) {
      const int rg = 2 * u + rg0;                     // wave-uniform
      if (rg >= kRowGroups) continue;
      const bool isA = rg < kRowsA / 8;
      const int rgl = isA ? rg : rg - kRowsA / 8;
      const bool ok = (isA ? m_base + 8 * rgl + r8 : 8 * rgl + r8) < B;
      const f32x4 v = (ok && kq_ok) ? stage[u] : zero4;
      *reinterpret_cast<f32x4*>((isA ? As + ldsA : Bs + ldsB) + 32 * rgl) = v;
    }
    __syncthreads();
    // the epoch-ordered copy: stored from the staging registers AFTER the barrier, so that no wave waits at the
    // barrier for another wave's global stores (the registers are reloaded right below); streaming stores: 172 MB
    // that nobody reads before the kernel ends (loads + stores alone 134 -> 97 us)
    if (a.dst != nullptr && (a.parts & 4) && kq_ok) {
#pragma unroll
      for (int u = 0; u < kLoads; ++u) {
        const int rg = 2 * u + rg0;
        if (rg >= kRowGroups) continue;
        const bool isA = rg < kRowsA / 8;
        const int rgl = isA ? rg : rg - kRowsA / 8;
        const bool ok = (isA ? m_base + 8 * rgl + r8 : 8 * rgl + r8) < B;
        if (ok && (isA || store_b))
          __builtin_nontemporal_store(stage[u], reinterpret_cast<f32x4*>(dst + (isA ? dstA : dstB) + (unsigned)(8 * rgl) * rowbytes + (unsigned)kq * 16u));
      }
    }
    if (c + 1 < nchunks && (a.parts & 1)) DTF_GRAM_ISSUE(c + 1);       // lands while this chunk is multiplied
    if (wave_works && (a.parts & 2)) {
      const int left = nkq - c * kChunkQ;
      const int nms = left >= kChunkQ ? 4 : (left + 3) >> 2;
#pragma unroll 1
      for (int ms = 0; ms < nms; ++ms) {
        {
          const f32x4 av = *reinterpret_cast<const f32x4*>(As + ((4 * ms + q) * kRowsA + 16 * wave + i16) * 4);
          f32x4 bv[7];
#pragma unroll
          for (int bi = 0; bi < 7; ++bi)
            bv[bi] = *reinterpret_cast<const f32x4*>(Bs + ((4 * ms + q) * kRowsB + 16 * bi + i16) * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int bi = 0; bi < 7; ++bi)
              acc[bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[bi][s], acc[bi], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: D[row = 4 q + reg][col = i16] of block (wave, bi) -> gram[pair][m][i] ---------------------
  if (wave_works) {
    float* g = a.gram + (size_t)pair * B * a.ld;
#pragma unroll
    for (int bi = 0; bi < 7; ++bi) {
      const int i = 16 * bi + i16;
      if (i >= a.ld) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m_base + 16 * wave + 4 * q + r;
        if (m < B) g[(size_t)m * a.ld + i] = i < B ? 1.0f + acc[bi][r] : 0.f;
      }
    }
  }
}

int g_gram_parts = 7;

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
  if ((long)rows_src * K * 4 >= (1L << 32) || (long)nb * B * K * 4 >= (1L << 32)) return DTF_ERR_SHAPE;
  if ((ysrc == nullptr) != (ydst == nullptr) || (ydst && C < 1)) return DTF_ERR_NULL;
  GramArgs a{src, perm, dst, gram, B, K, nb, ld, rows_src, ysrc, ydst, C, g_gram_parts};
  const int npairs = nb - 1;
  const int blocks = 2 * ((npairs + 7) / 8) * 8;
  hipLaunchKernelGGL(gather_gram_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

// Workgroups of the kernel that one CU holds at a time, as the runtime computes it (registers, LDS, wave slots).
extern "C" int dtf_debug_gram_blocks_per_cu(void) {
  int n = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gather_gram_kernel, kThreads, 0) != hipSuccess) return -1;
  return n;
}

// Timing experiments: run only some parts of the kernel (results are then wrong); 7 restores the full kernel.
extern "C" void dtf_debug_set_gram_parts(int parts) { g_gram_parts = parts; }
