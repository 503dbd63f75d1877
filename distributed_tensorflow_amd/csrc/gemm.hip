// Dense-layer kernels for models other than the reference's 784-100-10 network: an fp32 GEMM family on the gfx950
// matrix cores with the layer's element-wise work in its epilogue, a column sum (bias gradients) and a softmax
// cross-entropy head.  Together they are every launch of a fully connected stack of any depth and width
// (models/dense.py) and the MatMul of the graph executor's op-by-op path (ops/gemm.py).
//
//   C[M][N] = epilogue( sum_k opA(m, k) * opB(k, n) )
//
//   layout NN   A[m * lda + k]   B[k * ldb + n]      forward            a = act(x W + b)
//   layout NT   A[m * lda + k]   B[n * ldb + k]      input gradient     dx = (dz W^T) * act'(a)
//   layout TN   A[k * lda + m]   B[k * ldb + n]      weight gradient    W -= lr * x^T dz      (or dW = x^T dz)
//
// (the MatMul, BiasAdd, Sigmoid / Relu, their gradients and ApplyGradientDescent of a graph built like
// tfdist_between.py:55-66, for any number of layers).
//
// Tiling.  A workgroup owns a BM x BN tile of C (128 x 128 with four waves, or 64 x 64 with eight when the larger tile
// would leave CUs idle) and walks K in chunks of 32.  Both operands of a chunk sit in LDS as [k-quad][row or column][4 floats]
// images, so that lane (i = l & 15, q = l >> 4) reads the four k of k-quad 4 ms + q for row / column i with ONE
// ds_read_b128 (16 lanes = 256 contiguous bytes: no bank conflict) and feeds them to four v_mfma_f32_16x16x4_f32: MFMA s
// of macro-step ms contracts k = 16 ms + 4 q + s.  A wave computes a (BM/2) x (BN/2) quadrant of the 128-tile: 8 reads for 64 MFMAs
// per macro-step (32 x 16 of the 64-tile: 3 reads for 8 MFMAs).  Two chunk images: the loads of chunk c + 1 are requested before chunk c is multiplied
// and written to the other image after it; one barrier per chunk.
//
// Staging.  An operand whose k runs along memory (A of NN / NT, B of NT) arrives as 16-byte loads, 8 rows x 8 k-quads
// per wave-instruction (whole 128-byte lines), and is stored as it comes; an operand whose k is strided (B of NN, both of
// TN) arrives as four coalesced dwords per (k-quad, column) slot and is transposed in registers on its way to the 16-byte
// LDS store.  Rows, columns and k past the edge are clamped on the way in and zeroed on the way to LDS, so any M, N, K
// works; operands that are not 16-byte aligned take dword loads.
//
// Placement.  Workgroup ids are relabelled so that the workgroups of one XCD (ids equal modulo 8) own a contiguous run
// of tiles, ordered in groups of 8 tile rows: the tiles in flight on an XCD share a few row panels of A and column panels
// of B in that XCD's L2.
#include "dtf_api.h"
#include "dtf_common.h"
#include "dtf_internal.h"

#include <atomic>
#include <type_traits>

#ifndef DTF_GEMM_DEPTH64
#define DTF_GEMM_DEPTH64 1
#endif

namespace {

constexpr int kKQ = 8;                // k-quads per chunk
constexpr int kChunk = 4 * kKQ;       // k per chunk
constexpr int kGroupM = 8;            // tile rows per group of the tile order

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long lda, ldb, ldc;
  int epi, act;
  const float* bias;       // DTF_EPI_BIAS_ACT: [N] (may be null: no bias)
  const float* aux;        // DTF_EPI_ACT_BWD: activation outputs [M][ldaux]
  long ldaux;
  const float* lr;         // DTF_EPI_SGD: device learning rate
  float scale;             // DTF_EPI_SGD: C -= lr * scale * acc;  DTF_EPI_NONE: C = scale * acc
  int tiles_m, tiles_n;
  int parts;               // timing experiments (results then wrong, pinned chunk off): bit 0 loads, 1 LDS writes,
                           //   2 barrier, 3 LDS reads; 15 = the kernel as it is
  int splits;              // workgroups per tile, each with a range of K (1: none)
  long ws_floats;          // size of ws
  float* ws;               // splits > 1: [1024 tile counters, zero between launches][splits][tiles][BM * BN] partial sums
};

// What one thread moves per chunk for one operand: NS = 8 R / threads slots of four k each (NW waves per workgroup).
//
// Slot j of a thread, by layout of the operand in memory:
//   k along memory (KC)   row 8 (wave + NW j) + (lane & 7), k-quad lane >> 3         (8 rows x 128 bytes per instruction)
//   k strided             row or column tid % R, k-quad tid / R + (threads / R) j    (256-byte row segments)
// Every load is a buffer load: the descriptor points at the tile (KC) or at the tile's rows of the chunk (strided), the
// per-lane part of the address is a 32-bit byte offset computed ONCE (rows and columns past the edge of the operand
// clamped to its last one), and what changes per chunk, slot and k is a scalar offset - no vector arithmetic per load.
// (Forming 64-bit addresses per load instead cost ~280 vector instructions per chunk and wave against ~80, which the
// matrix pipe does not hide: 103 -> 118 TFLOP/s at 4096^3 NN.)  Only in the last chunk of a K that is not a multiple of
// 32 are lanes whose k lies past the end redirected to the chunk's first k; rows, columns and k past the edge are zeroed
// on their way to LDS.
template <int R, bool KC, bool VEC, int D, int NW>
struct Stager {
  static constexpr int NT = NW * 64;                        // threads of the workgroup
  static constexpr int NS = R * kKQ / NT;
  static constexpr int NV = KC ? NS : 1;                    // distinct per-lane offsets
  static constexpr int kQStep = KC ? 0 : NT / R;            // k-quads between the slots of a thread (strided form)
  static constexpr int kLdsStep = KC ? 8 * NW * 4 : kQStep * R * 4;
  const float* origin;     // first element of the tile: (row r0, k 0) or (k 0, column r0)
  long ld;
  int K;
  int wave, lane, tid;
  unsigned voff[NV];       // bytes from the descriptor's base: the per-lane part of every address
  unsigned vcol;           // strided form: the column part of voff[0] alone (see kq_lane)
  int kq_lane;             // strided form with R < 64: k-quads between this lane and lane 0 of its wave (0 or 1, ...);
                           //   that part of the address is per lane and sits in voff[0], the wave's part is scalar
  bool ok[NV];             // the slot's row / column lies inside the operand
  int lds0;                // floats: LDS offset of slot 0 (slot j: + kLdsStep * j)
  bool interior;
  f32x4 v[D][NS];          // D chunks in flight between memory and LDS

  __device__ __forceinline__ static __amdgpu_buffer_rsrc_t descriptor(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, (int)0xffffffffu, 0x00020000);
  }
  __device__ __forceinline__ void init(const float* base, long ld_, int rows, int r0, int K_, int tid_) {
    ld = ld_;
    K = K_;
    tid = tid_;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    interior = r0 + R <= rows;
    origin = KC ? base + (long)r0 * ld : base + r0;
    if (KC) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int r = 8 * (wave + NW * j) + (lane & 7);
        ok[j] = r0 + r < rows;
        voff[j] = (unsigned)(((long)(ok[j] ? r : rows - 1 - r0) * ld + 4 * (lane >> 3)) * 4);
      }
      lds0 = ((lane >> 3) * R + 8 * wave + (lane & 7)) * 4;
    } else {
      const int c = tid % R;
      ok[0] = r0 + c < rows;
      vcol = (unsigned)((ok[0] ? c : rows - 1 - r0) * 4);
      // R >= 64: a wave's lanes share one k-quad.  R = 32: lanes 32-63 hold the next one, 4 rows of the operand on.
      kq_lane = tid / R - __builtin_amdgcn_readfirstlane(tid / R);
      voff[0] = vcol + (unsigned)((long)(4 * kq_lane) * ld * 4);
      lds0 = ((tid / R) * R + c) * 4;
    }
  }
  template <int S>
  __device__ __forceinline__ void issue(int k0) {
    const bool full = k0 + kChunk <= K;
    if (KC) {
      const __amdgpu_buffer_rsrc_t rsrc = descriptor(origin);
      const int kl = 4 * (lane >> 3);                 // this lane's first k inside the chunk
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (VEC) {        // K is a multiple of 4: a quad is inside or outside as a whole
          const unsigned off = (full || k0 + kl < K) ? voff[j] : voff[j] - 4 * kl;
          v[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, k0 * 4, 0));
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const unsigned off = (full || k0 + kl + s < K) ? voff[j] + 4 * s : voff[j] - 4 * kl;
            v[S][j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, k0 * 4, 0));
          }
        }
      }
    } else {
      const __amdgpu_buffer_rsrc_t rsrc = descriptor(origin + (long)k0 * ld);      // the chunk's first row
      const int kq_w = __builtin_amdgcn_readfirstlane(tid / R);                    // lane 0's k-quad: uniform in a wave
      if (full) {
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            v[S][j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                       rsrc, voff[0], (int)((4 * (kq_w + kQStep * j) + s) * ld * 4), 0));
      } else {
        // the last chunk of a K that is no multiple of 32: the whole address per lane, a k past the end reads the
        // chunk's first row instead (and is zeroed on its way to LDS)
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int kk = 4 * (kq_w + kq_lane + kQStep * j) + s;
            const unsigned off = k0 + kk < K ? vcol + (unsigned)((long)kk * ld * 4) : vcol;
            v[S][j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
          }
      }
    }
  }
  // The same two steps for a tile that lies inside the operand and a chunk that lies inside K: straight-line code (no
  // test, no clamp), so that the caller can pin its instructions between the MFMAs of a chunk.
  static constexpr int kLoads = (KC && VEC) ? NS : 4 * NS;        // load instructions per chunk
  template <int S>
  __device__ __forceinline__ void issue_fast(int k0) {
    if (KC) {
      const __amdgpu_buffer_rsrc_t rsrc = descriptor(origin);
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (VEC) {
          v[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j], k0 * 4, 0));
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
            v[S][j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j] + 4 * s, k0 * 4, 0));
        }
      }
    } else {
      const __amdgpu_buffer_rsrc_t rsrc = descriptor(origin + (long)k0 * ld);
      const int kq_w = __builtin_amdgcn_readfirstlane(tid / R);
#pragma unroll
      for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          v[S][j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     rsrc, voff[0], (int)((4 * (kq_w + kQStep * j) + s) * ld * 4), 0));
    }
  }
  template <int S>
  __device__ __forceinline__ void write_fast(float* img) const {
#pragma unroll
    for (int j = 0; j < NS; ++j) *reinterpret_cast<f32x4*>(img + lds0 + kLdsStep * j) = v[S][j];
  }
  template <int S>
  __device__ __forceinline__ void write(int k0, float* img) const {
    if (interior && k0 + kChunk <= K) {
#pragma unroll
      for (int j = 0; j < NS; ++j) *reinterpret_cast<f32x4*>(img + lds0 + kLdsStep * j) = v[S][j];
    } else {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int k = k0 + 4 * (KC ? lane >> 3 : tid / R + kQStep * j);
        const bool row_ok = ok[KC ? j : 0];
        f32x4 o = v[S][j];
#pragma unroll
        for (int s = 0; s < 4; ++s) o[s] = (row_ok && k + s < K) ? o[s] : 0.f;
        *reinterpret_cast<f32x4*>(img + lds0 + kLdsStep * j) = o;
      }
    }
  }
};

template <int BM, int BN, int LAYOUT, bool VEC, int NW>
__global__ __launch_bounds__(NW * 64, 2) void gemm_f32_kernel(GemmArgs a) {
  constexpr bool AK = LAYOUT != DTF_GEMM_TN, BK = LAYOUT == DTF_GEMM_NT;
  constexpr int kThreads = NW * 64;
  // The waves of a workgroup tile its output 2 x 2 (four waves: the 128-tile) or 2 x 4 (eight waves: the 64-tile, and the
  // 128-tile where a CU holds one workgroup anyway).
  // Eight waves on the 128-tile (four per SIMD with two workgroups per CU, 128 registers each) measured 108 - 120 TFLOP/s
  // against 117 - 126 with four, and did not fit its registers.  The 64-tile is what shapes with few tiles get, often ONE
  // workgroup per CU: with four waves that is one wave per SIMD, whose matrix instructions do not follow each other
  // without gaps (the K loop with everything but its MFMAs switched off: 115 TFLOP/s); eight waves put two on every
  // SIMD -- 1024^3 NN 28.7 -> 24.9 us, 8192 x 128 x 2048 NN 49.5 -> 44.7 us, 100 x 1024 x 784 TN 13.4 -> 11.2 us.
  constexpr int WGN = NW / 2;
  constexpr int TM = BM / 32, TN = BN / (16 * WGN);    // 16 x 16 blocks per wave in each direction
  constexpr int kAs = kKQ * BM * 4, kBs = kKQ * BN * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][kAs]
  float* Bs = smem + 2 * kAs;        // [2][kBs]

  // ---- which tile (see "Placement") ----
  const int T = a.tiles_m * a.tiles_n;
  int tile, split;
  {
    const int total = T * a.splits;
    const int orig = blockIdx.x, xcd = orig & 7, qq = total >> 3, rr = total & 7;
    const int id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    tile = id / a.splits;          // the workgroups of a tile are neighbours on one XCD
    split = id - tile * a.splits;
  }
  const int group_sz = kGroupM * a.tiles_n, g = tile / group_sz, first_m = g * kGroupM;
  const int gm = a.tiles_m - first_m < kGroupM ? a.tiles_m - first_m : kGroupM;
  const int in_group = tile - g * group_sz;
  const int m0 = (first_m + in_group % gm) * BM, n0 = (in_group / gm) * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int wm0 = (wave / WGN) * (BM / 2), wn0 = (wave % WGN) * (BN / WGN);

  // Chunks in flight between memory and LDS.  Two, three or four for the small tile (-DDTF_GEMM_DEPTH64=2|3|4, pinned
  // chunks D at a time) measured the same as one or a little worse (10000 x 100 x 784 NT: 32.8 / 33.3 / 35.0 / 35.5 us,
  // 8192 x 128 x 784 NT: 20.3 / 19.7 / 20.9 / 21.5 us): a chunk of the small tile is not waiting for memory.
  constexpr int D = BM <= 64 ? DTF_GEMM_DEPTH64 : 1;
  Stager<BM, AK, VEC, D, NW> sa;
  Stager<BN, BK, VEC, D, NW> sb;
  sa.init(a.A, a.lda, a.M, m0, a.K, tid);
  sb.init(a.B, a.ldb, a.N, n0, a.K, tid);

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[TM][TN];
#pragma unroll
  for (int bi = 0; bi < TM; ++bi)
#pragma unroll
    for (int bj = 0; bj < TN; ++bj) acc[bi][bj] = zero4;

  // this workgroup's chunks of K: all of them, or range `split` of `splits` equal ranges (the launcher leaves none empty)
  const int all_chunks = (a.K + kChunk - 1) / kChunk;
  const int per_split = (all_chunks + a.splits - 1) / a.splits;
  const int cbeg = split * per_split;
  const int nchunks = all_chunks - cbeg < per_split ? all_chunks - cbeg : per_split;
  // operand registers of one macro-step: [0] = macro-step 0 of a chunk, [1] = macro-step 1
  f32x4 av[2][TM], bv[2][TN];
  auto read_ops = [&](const float* Ac, const float* Bc, int ms, f32x4 (&ar)[TM], f32x4 (&br)[TN]) {
    const int kq = 4 * ms + q;
#pragma unroll
    for (int bi = 0; bi < TM; ++bi) ar[bi] = *reinterpret_cast<const f32x4*>(Ac + (kq * BM + wm0 + 16 * bi + i16) * 4);
#pragma unroll
    for (int bj = 0; bj < TN; ++bj) br[bj] = *reinterpret_cast<const f32x4*>(Bc + (kq * BN + wn0 + 16 * bj + i16) * 4);
  };
  auto multiply = [&](const f32x4 (&ar)[TM], const f32x4 (&br)[TN], int s0, int s1) {
#pragma unroll
    for (int s = s0; s < s1; ++s)
#pragma unroll
      for (int bi = 0; bi < TM; ++bi)
#pragma unroll
        for (int bj = 0; bj < TN; ++bj)
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[bi][s], br[bj][s], acc[bi][bj], 0, 0, 0);
  };

  // Order of one chunk (so that a wave has matrix work queued whenever it waits for memory, LDS or the barrier):
  //   operands of macro-step 1 requested from LDS | macro-step 0 multiplied | chunk c+1 (loaded a chunk ago) written to
  //   the other image, chunk c+2 requested from memory | three quarters of macro-step 1 | barrier | operands of chunk
  //   c+1's macro-step 0 requested from LDS | the last quarter of macro-step 1.
  // Image (c+1) & 1 is written while everybody reads image c & 1, and read only after the barrier that follows the
  // writes; its previous contents (chunk c-1) were last read before the barrier of chunk c-1.
  // Loads of chunk t land in register set t % D; chunk c writes chunk c+1 to LDS and requests chunk c+1+D into the set
  // that this frees.
  auto chunk = [&](int c, auto parity) {
    constexpr int S = (decltype(parity)::value + 1) % D;          // register set of chunk c+1
    const float* Ac = As + (c & 1) * kAs;
    const float* Bc = Bs + (c & 1) * kBs;
    float* An = As + ((c + 1) & 1) * kAs;
    float* Bn = Bs + ((c + 1) & 1) * kBs;
    const bool more = c + 1 < nchunks;
    if (a.parts & 8) read_ops(Ac, Bc, 1, av[1], bv[1]);
    multiply(av[0], bv[0], 0, 4);
    if (more) {
      if (a.parts & 2) {
        sa.template write<S>((cbeg + c + 1) * kChunk, An);
        sb.template write<S>((cbeg + c + 1) * kChunk, Bn);
      }
      if (c + 1 + D < nchunks && (a.parts & 1)) {
        sa.template issue<S>((cbeg + c + 1 + D) * kChunk);
        sb.template issue<S>((cbeg + c + 1 + D) * kChunk);
      }
    }
    multiply(av[1], bv[1], 0, 3);
    __syncthreads();
    if (more) read_ops(An, Bn, 0, av[0], bv[0]);
    multiply(av[1], bv[1], 3, 4);
  };
  // The steady-state chunk (chunks c+1 and c+1+D inside K): the same steps as straight-line
  // code with their order PINNED, every LDS and memory instruction issued in the shadow of a matrix instruction - a wave
  // issues in order, and an MFMA occupies the matrix pipe for 32 cycles of which its own issue takes 4: one or two other
  // instructions fit behind each, a block of sixteen of them does not (measured with parts of the chunk switched off:
  // the unpinned chunk loses 15 % to each of its load, LDS-write, LDS-read and barrier blocks on its own).
  constexpr int kMfma = 0x008, kVmemRead = 0x020, kDsRead = 0x100, kDsWrite = 0x200;
  constexpr int NRD = TM + TN;                                   // LDS reads of one macro-step
  constexpr int MF = 4 * TM * TN;                                // MFMAs of one macro-step
  constexpr int NWR = Stager<BM, AK, VEC, D, NW>::NS + Stager<BN, BK, VEC, D, NW>::NS;
  constexpr int NVM = Stager<BM, AK, VEC, D, NW>::kLoads + Stager<BN, BK, VEC, D, NW>::kLoads;
  auto chunk_fast = [&](int c, auto parity) {
    constexpr int S = (decltype(parity)::value + 1) % D;
    const float* Ac = As + (c & 1) * kAs;
    const float* Bc = Bs + (c & 1) * kBs;
    float* An = As + ((c + 1) & 1) * kAs;
    float* Bn = Bs + ((c + 1) & 1) * kBs;
    read_ops(Ac, Bc, 1, av[1], bv[1]);
    multiply(av[0], bv[0], 0, 4);
    sa.template write_fast<S>(An);
    sb.template write_fast<S>(Bn);
    sa.template issue_fast<S>((cbeg + c + 1 + D) * kChunk);
    sb.template issue_fast<S>((cbeg + c + 1 + D) * kChunk);
    multiply(av[1], bv[1], 0, 3);
    {   // macro-step 0: the operand reads of macro-step 1 spread over its first half
      constexpr int per = MF / (2 * NRD) > 0 ? MF / (2 * NRD) : 1;
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        __builtin_amdgcn_sched_group_barrier(kDsRead, 1, 0);
        __builtin_amdgcn_sched_group_barrier(kMfma, per, 0);
      }
      __builtin_amdgcn_sched_group_barrier(kMfma, MF - per * NRD, 0);
      // three quarters of macro-step 1: an LDS write behind each of the first MFMAs, then the loads of chunk c+1+D
      constexpr int M1 = 3 * TM * TN;
      constexpr int left = M1 - NWR > 0 ? M1 - NWR : 1;          // MFMAs left for the loads
      constexpr int lper = (NVM + left - 1) / left;              // loads behind one MFMA
      constexpr int lgroups = (NVM + lper - 1) / lper;
#pragma unroll
      for (int i = 0; i < NWR; ++i) {
        __builtin_amdgcn_sched_group_barrier(kDsWrite, 1, 0);
        __builtin_amdgcn_sched_group_barrier(kMfma, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < lgroups; ++i) {
        __builtin_amdgcn_sched_group_barrier(kVmemRead, lper, 0);
        __builtin_amdgcn_sched_group_barrier(kMfma, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(kMfma, M1 - NWR - lgroups > 0 ? M1 - NWR - lgroups : 0, 0);
    }
    __syncthreads();
    read_ops(An, Bn, 0, av[0], bv[0]);
    multiply(av[1], bv[1], 3, 4);
    {   // the last quarter: the operand reads of the next chunk's macro-step 0 behind its MFMAs
      constexpr int M3 = TM * TN;
      constexpr int per = M3 / NRD > 0 ? M3 / NRD : 1;
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        __builtin_amdgcn_sched_group_barrier(kDsRead, 1, 0);
        __builtin_amdgcn_sched_group_barrier(kMfma, per, 0);
      }
      __builtin_amdgcn_sched_group_barrier(kMfma, M3 - per * NRD > 0 ? M3 - per * NRD : 0, 0);
    }
  };
  // Chunk c may take the pinned form when chunks c+1 and c+1+D exist and are full.  Edge tiles take it too: the per-lane
  // offsets of rows and columns past the edge point at the operand's last row or column (valid memory), and what those
  // lanes carry only reaches accumulators whose row or column the epilogue does not store.
  auto fast_ok = [&](int c) { return (a.parts == 15) && c + 1 + D < nchunks && (cbeg + c + 2 + D) * kChunk <= a.K; };

  sa.template issue<0>(cbeg * kChunk);
  sb.template issue<0>(cbeg * kChunk);
  sa.template write<0>(cbeg * kChunk, As);
  sb.template write<0>(cbeg * kChunk, Bs);
  if (nchunks > 1) {
    sa.template issue<1 % D>((cbeg + 1) * kChunk);
    sb.template issue<1 % D>((cbeg + 1) * kChunk);
  }
  if (D > 1 && nchunks > 2) {
    sa.template issue<2 % D>((cbeg + 2) * kChunk);
    sb.template issue<2 % D>((cbeg + 2) * kChunk);
  }
  if (D > 2 && nchunks > 3) {
    sa.template issue<3 % D>((cbeg + 3) * kChunk);
    sb.template issue<3 % D>((cbeg + 3) * kChunk);
  }
  if (D > 3 && nchunks > 4) {
    sa.template issue<4 % D>((cbeg + 4) * kChunk);
    sb.template issue<4 % D>((cbeg + 4) * kChunk);
  }
  __syncthreads();
  read_ops(As, Bs, 0, av[0], bv[0]);
  {
    int c = 0;
    // the steady state: pinned chunks, D at a time (chunk c + i works with register set (c + i + 1) % D)
    for (; fast_ok(c + D - 1); c += D) {
      chunk_fast(c, std::integral_constant<int, 0>{});
      if constexpr (D > 1) chunk_fast(c + 1, std::integral_constant<int, 1 % D>{});
      if constexpr (D > 2) chunk_fast(c + 2, std::integral_constant<int, 2 % D>{});
      if constexpr (D > 3) chunk_fast(c + 3, std::integral_constant<int, 3 % D>{});
    }
    for (; c < nchunks; c += D) {       // the last chunks: the same steps under conditions
      chunk(c, std::integral_constant<int, 0>{});
      if (D > 1 && c + 1 < nchunks) chunk(c + 1, std::integral_constant<int, 1 % D>{});
      if (D > 2 && c + 2 < nchunks) chunk(c + 2, std::integral_constant<int, 2 % D>{});
      if (D > 3 && c + 3 < nchunks) chunk(c + 3, std::integral_constant<int, 3 % D>{});
    }
  }

  // ---- K split over several workgroups: partial sums meet in the workspace, the LAST workgroup of a tile to arrive
  // adds them in range order (the same bits whatever the arrival order) and runs the epilogue.  Hand-off: plain stores,
  // every wave waits for its stores, barrier, ONE agent-scope release ahead of the ticket, and ONE agent-scope acquire in
  // the workgroup that drew the last ticket before anybody reads; the counter goes back to zero for the next launch.
  if (a.splits > 1) {
    constexpr int kTileFloats = BM * BN;
    unsigned* counters = reinterpret_cast<unsigned*>(a.ws);
    float* slabs = a.ws + 1024;
    float* mine = slabs + ((long)split * T + tile) * kTileFloats;
#pragma unroll
    for (int bi = 0; bi < TM; ++bi)
#pragma unroll
      for (int bj = 0; bj < TN; ++bj)
        *reinterpret_cast<f32x4*>(mine + ((bi * TN + bj) * kThreads + tid) * 4) = acc[bi][bj];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* last = reinterpret_cast<int*>(smem);          // (LDS is free: the K loop ended behind a barrier)
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned ticket = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *last = ticket == (unsigned)a.splits - 1u ? 1 : 0;
      if (*last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        counters[tile] = 0u;
      }
    }
    __syncthreads();
    if (*last == 0) return;
#pragma unroll
    for (int bi = 0; bi < TM; ++bi)
#pragma unroll
      for (int bj = 0; bj < TN; ++bj) acc[bi][bj] = zero4;
    for (int sidx = 0; sidx < a.splits; ++sidx) {
      const float* slab = slabs + ((long)sidx * T + tile) * kTileFloats;
#pragma unroll
      for (int bi = 0; bi < TM; ++bi)
#pragma unroll
        for (int bj = 0; bj < TN; ++bj)
          acc[bi][bj] += *reinterpret_cast<const f32x4*>(slab + ((bi * TN + bj) * kThreads + tid) * 4);
    }
  }

  // ---- epilogue: register r of block (bi, bj) is C[m0 + wm0 + 16 bi + 4 q + r][n0 + wn0 + 16 bj + i16] ----
  const float step = a.epi == DTF_EPI_SGD ? a.lr[0] * a.scale : a.scale;
#pragma unroll
  for (int bj = 0; bj < TN; ++bj) {
    const int n = n0 + wn0 + 16 * bj + i16;
    if (n >= a.N) continue;
    const float bias = (a.epi == DTF_EPI_BIAS_ACT && a.bias != nullptr) ? a.bias[n] : 0.f;
#pragma unroll
    for (int bi = 0; bi < TM; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm0 + 16 * bi + 4 * q + r;
        if (m >= a.M) continue;
        float* c = a.C + (long)m * a.ldc + n;
        const float s = acc[bi][bj][r];
        if (a.epi == DTF_EPI_BIAS_ACT)
          *c = a.act < 0 ? s + bias : dtf_act_fwd(s + bias, a.act);
        else if (a.epi == DTF_EPI_ACT_BWD)
          *c = s * dtf_act_bwd(a.aux[(long)m * a.ldaux + n], a.act);
        else if (a.epi == DTF_EPI_SGD)
          *c -= step * s;
        else
          *c = step * s;
      }
  }
}

int g_parts = 15;                                        // dtf_debug_set_gemm_parts
int g_split_max_tiles = 128, g_split_target = 256;       // tuning knobs of the K split (dtf_debug_set_gemm_split)

template <int BM, int BN, int LAYOUT, bool VEC, int NW = 4>
int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  constexpr int kThreads = NW * 64;
  constexpr int bytes = 2 * kKQ * (BM + BN) * 4 * (int)sizeof(float);
  static std::atomic<unsigned long long> configured{0};    // one bit per device: the attribute belongs to the device's copy
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DTF_ERR_LAUNCH;
  if (dev >= 64 || !((configured.load(std::memory_order_acquire) >> dev) & 1ull)) {
    if (hipFuncSetAttribute((const void*)gemm_f32_kernel<BM, BN, LAYOUT, VEC, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            bytes) != hipSuccess)
      return DTF_ERR_LAUNCH;
    if (dev < 64) configured.fetch_or(1ull << dev, std::memory_order_release);
  }
  GemmArgs b = a;
  b.tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.N + BN - 1) / BN;
  // K split: when the tiles leave most CUs idle and K is long, several workgroups share a tile (see the kernel's end).
  // Needs the caller's workspace: 1024 counters + splits x tiles x BM x BN floats.
  const long T = (long)b.tiles_m * b.tiles_n, chunks = (a.K + kChunk - 1) / kChunk;
  int splits = 1;
  if (a.ws != nullptr && T <= g_split_max_tiles && chunks >= 4) {
    long want = a.splits > 0 ? a.splits : g_split_target / T;  // fill the chip once
    if (want > chunks / 2) want = chunks / 2;                  // at least two chunks per workgroup
    if (want > 16) want = 16;
    const long room = (a.ws_floats - 1024) / (T * BM * BN);
    if (want > room) want = room;
    if (want >= 2) {
      const long per = (chunks + want - 1) / want;
      splits = (int)((chunks + per - 1) / per);                // no empty range
    }
  }
  b.splits = splits;
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, LAYOUT, VEC, NW>), dim3((unsigned)(T * splits)), dim3(kThreads), bytes, stream,
                     b);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

template <int LAYOUT, bool VEC>
int launch_layout(const GemmArgs& a, int tile, hipStream_t stream) {
  // Which tile: the one whose rounds of co-resident workgroups end first.  512 large tiles (two per CU) or 1024 small
  // ones (four per CU) run together; a round of small tiles does half the work of a round of large ones at about 0.9 of
  // the rate (measured: 2048^3 as 1024 small tiles 113 TFLOP/s, 4096^3 as 1024 large tiles 124), so a large-tile grid
  // that would leave its last round mostly empty (3072^3: 576 tiles) is better served by small tiles.
  const long big = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const long small = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
  const double t_big = (double)((big + 511) / 512), t_small = 0.55 * (double)((small + 1023) / 1024);
  if (tile == 128 || (tile == 0 && big >= 512 && t_big <= t_small)) return launch_gemm<128, 128, LAYOUT, VEC, 4>(a, stream);
  // A grid of large tiles that fills the chip once without filling it twice (2048^3, 1024 x 4096 x 1024: 256 tiles)
  // runs as ONE workgroup per CU; with eight waves (a 64 x 32 quadrant each, two waves per SIMD) that beats both four
  // waves (one per SIMD) and the same work as 1024 small tiles: 2048^3 0.85 - 0.89 -> 0.92 - 0.97 of the library.
  if (tile == 129 || (tile == 0 && big > 224 && big <= 256)) return launch_gemm<128, 128, LAYOUT, VEC, 8>(a, stream);
  // Between one and one and a half small tiles per CU (10000 x 100 x 784: 314 of them), the CUs with two set the time;
  // tiles of half the height spread the same work as three per CU at most: 32.7 -> 27.8 us there (NT).  (TN: A arrives
  // strided, and the two halves of a wave then stage neighbouring k-quads of the tile's 32 rows.)  Not where K is split
  // (few tiles).
  {
    const long half = (long)((a.M + 31) / 32) * ((a.N + 63) / 64);
    const long load_small = (small + 255) / 256 * 2, load_half = (half + 255) / 256;
    if (tile == 32 || (tile == 0 && small > 256 && 5 * load_half <= 4 * load_small))
      return launch_gemm<32, 64, LAYOUT, VEC, 4>(a, stream);
  }
  return launch_gemm<64, 64, LAYOUT, VEC, 8>(a, stream);
}

// ---- column sums: out[n] (op)= scale * sum_m z[m][n] -------------------------------------------------------------
// A workgroup owns 64 columns and a segment of up to kColsumRows rows; its 16 waves walk the rows 16 apart with four
// loads in flight each (every load a 256-byte row segment; the 13 dependent round trips of a four-wave, two-load version
// took 4.6 us for 100 rows) and meet in LDS.  Several segments (tall matrices) add their parts with float atomics: the
// SGD form is a sum of decrements anyway, the plain form then needs `out` zeroed first (done by the launcher).
constexpr int kColsumRows = 2048;
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ z, long ldz, int M, int N, float* out,
                                                      int mode, float scale, const float* lr) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int m_begin = blockIdx.y * kColsumRows, m_end = m_begin + kColsumRows < M ? m_begin + kColsumRows : M;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    int m = m_begin + wave;
    for (; m + 48 < m_end; m += 64) {
      s0 += z[(long)m * ldz + n];
      s1 += z[(long)(m + 16) * ldz + n];
      s2 += z[(long)(m + 32) * ldz + n];
      s3 += z[(long)(m + 48) * ldz + n];
    }
    for (; m < m_end; m += 16) s0 += z[(long)m * ldz + n];
  }
  part[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && n < N) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += part[w][lane];
    const float v = mode == DTF_EPI_SGD ? -lr[0] * scale * s : scale * s;
    if (gridDim.y > 1)
      atomicAdd(out + n, v);
    else if (mode == DTF_EPI_SGD)
      out[n] += v;
    else
      out[n] = v;
  }
}

// The same for several matrices in ONE launch (the bias gradients of all layers at the end of a backward pass): block
// b belongs to the matrix whose range of column blocks contains it.
struct ColsumTable {
  const float* z[16];
  float* out[16];
  long ldz[16];
  int M[16], N[16];
  int first_block[17];      // column blocks of matrix t: [first_block[t], first_block[t + 1])
  int count;
};
__global__ __launch_bounds__(1024) void colsum_multi_kernel(ColsumTable t, int mode, float scale, const float* lr) {
  __shared__ float part[16][64];
  int which = 0;
#pragma unroll 1
  while (which + 1 < t.count && (int)blockIdx.x >= t.first_block[which + 1]) ++which;
  const float* __restrict__ z = t.z[which];
  const long ldz = t.ldz[which];
  const int M = t.M[which], N = t.N[which];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = ((int)blockIdx.x - t.first_block[which]) * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    int m = wave;
    for (; m + 48 < M; m += 64) {
      s0 += z[(long)m * ldz + n];
      s1 += z[(long)(m + 16) * ldz + n];
      s2 += z[(long)(m + 32) * ldz + n];
      s3 += z[(long)(m + 48) * ldz + n];
    }
    for (; m < M; m += 16) s0 += z[(long)m * ldz + n];
  }
  part[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && n < N) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += part[w][lane];
    float* out = t.out[which];
    if (mode == DTF_EPI_SGD)
      out[n] -= lr[0] * scale * s;
    else
      out[n] = scale * s;
  }
}

// ---- start of a training step: the step's loss / accuracy accumulators cleared, the step counter bumped (one thread) ---
__global__ void dense_step_begin_kernel(float* loss_sum, int* correct, float* global_step) {
  if (loss_sum != nullptr) *loss_sum = 0.f;
  if (correct != nullptr) *correct = 0;
  if (global_step != nullptr) *global_step += 1.0f;
}

// ---- softmax cross-entropy head over logits [M][C] -----------------------------------------------------------------
//   loss_sum += sum_m -sum_c y log softmax(z)     correct += #{m : argmax z == argmax y}      (accumulated)
//   dz[m][c] = grad_scale * (softmax(z)[m][c] * sum_c' y[m][c'] - y[m][c])                      (dz may be null)
// (the Softmax, Log, Mul, Sum, Mean, ArgMax, Equal nodes of tfdist_between.py:59-70 and their gradients, with the
// logarithm taken of the shifted exponentials' sum instead of the probability.)  A wave per row, 4 rows per workgroup.
// One row by one wave: returns the row's loss (every lane) and whether it is classified like its labels.
__device__ __forceinline__ float xent_row(const float* __restrict__ zr, const float* __restrict__ yr, int C, int lane,
                                          float* dzr, float grad_scale, int& hit) {
  float mx = -INFINITY, ymx = -INFINITY;
  int zi = 0x7fffffff, yi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = zr[c], t = yr[c];
    if (v > mx) { mx = v; zi = c; }
    if (t > ymx) { ymx = t; yi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {      // ties go to the smallest index (tf.argmax)
    const float ov = __shfl_xor(mx, o, 64), oy = __shfl_xor(ymx, o, 64);
    const int oi = __shfl_xor(zi, o, 64), oyi = __shfl_xor(yi, o, 64);
    if (ov > mx || (ov == mx && oi < zi)) { mx = ov; zi = oi; }
    if (oy > ymx || (oy == ymx && oyi < yi)) { ymx = oy; yi = oyi; }
  }
  float se = 0.f, ys = 0.f, yz = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float t = yr[c];
    se += expf(zr[c] - mx);
    ys += t;
    yz += t * (zr[c] - mx);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    se += __shfl_xor(se, o, 64);
    ys += __shfl_xor(ys, o, 64);
    yz += __shfl_xor(yz, o, 64);
  }
  hit = zi == yi ? 1 : 0;
  if (dzr != nullptr) {
    const float inv = 1.f / se;
    for (int c = lane; c < C; c += 64) dzr[c] = grad_scale * (expf(zr[c] - mx) * inv * ys - yr[c]);
  }
  return ys * logf(se) - yz;                // -sum_c y (z - mx - lse)
}

__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                           int M, int C, float* dz, float grad_scale, float* loss_sum,
                                                           int* correct, float* row_loss) {
  __shared__ float wl[4];
  __shared__ int wc[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  float loss = 0.f;
  int hit = 0;
  if (m < M) {
    loss = xent_row(z + (long)m * C, y + (long)m * C, C, lane, dz != nullptr ? dz + (long)m * C : nullptr, grad_scale, hit);
    if (row_loss != nullptr && lane == 0) row_loss[m] = loss;
  }
  if (lane == 0) {
    wl[wave] = loss;
    wc[wave] = hit;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (loss_sum != nullptr) atomicAdd(loss_sum, (wl[0] + wl[1]) + (wl[2] + wl[3]));
    if (correct != nullptr) atomicAdd(correct, wc[0] + wc[1] + wc[2] + wc[3]);
  }
}

// The head of a training step over MANY workgroups without an accumulator to clear first: every workgroup leaves the
// loss and the hit count of its 4 rows in `scratch` ([blocks] losses, [blocks] counts as floats, one ticket word, all
// owned by the caller and zero-ticketed between launches), and the workgroup that draws the last ticket adds them in
// block order (the same bits on every run), STORES the step's loss sum and correct count, bumps the step counter and
// puts the ticket back to zero.  Hand-off as in the K split of the GEMM: plain stores, one agent-scope release ahead of
// the ticket, one agent-scope acquire in the last workgroup before anybody reads.
__global__ __launch_bounds__(256) void softmax_xent_head_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                                int M, int C, float* dz, float grad_scale, float* scratch,
                                                                float* loss_out, int* correct_out, float* global_step) {
  __shared__ float wl[4];
  __shared__ int wc[4];
  __shared__ int last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  const int blocks = gridDim.x;
  float loss = 0.f;
  int hit = 0;
  if (m < M) loss = xent_row(z + (long)m * C, y + (long)m * C, C, lane, dz != nullptr ? dz + (long)m * C : nullptr, grad_scale, hit);
  if (lane == 0) {
    wl[wave] = loss;
    wc[wave] = hit;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    scratch[blockIdx.x] = (wl[0] + wl[1]) + (wl[2] + wl[3]);
    scratch[blocks + blockIdx.x] = (float)(wc[0] + wc[1] + wc[2] + wc[3]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 2 * blocks);
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = t == (unsigned)blocks - 1u ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      *ticket = 0u;
    }
  }
  __syncthreads();
  if (!last) return;
  // fixed order: thread t adds blocks t, t + 256, ...; then the 256 partial sums in index order
  __shared__ float pl[256], pc[256];
  float sl = 0.f, sc = 0.f;
  for (int b = threadIdx.x; b < blocks; b += 256) {
    sl += scratch[b];
    sc += scratch[blocks + b];
  }
  pl[threadIdx.x] = sl;
  pc[threadIdx.x] = sc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tl = 0.f, tc = 0.f;
    for (int i = 0; i < 256; ++i) {
      tl += pl[i];
      tc += pc[i];
    }
    if (loss_out != nullptr) *loss_out = tl;
    if (correct_out != nullptr) *correct_out = (int)(tc + 0.5f);
    if (global_step != nullptr) *global_step += 1.0f;
  }
}

// The head of a training step in ONE workgroup (batches of up to a few thousand rows): the 16 waves walk the rows 16
// apart, the row losses are added in a fixed order, and the step's loss sum and correct count are STORED (no accumulator
// to clear first, the same bits on every run) together with the bump of the step counter.
__global__ __launch_bounds__(1024) void softmax_xent_step_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                                 int M, int C, float* dz, float grad_scale,
                                                                 float* loss_out, int* correct_out, float* global_step) {
  __shared__ float wl[16];
  __shared__ int wc[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float loss = 0.f;
  int hits = 0;
  for (int m = wave; m < M; m += 16) {
    int hit;
    loss += xent_row(z + (long)m * C, y + (long)m * C, C, lane, dz != nullptr ? dz + (long)m * C : nullptr, grad_scale, hit);
    hits += hit;
  }
  if (lane == 0) {
    wl[wave] = loss;
    wc[wave] = hits;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    int h = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      s += wl[w];
      h += wc[w];
    }
    if (loss_out != nullptr) *loss_out = s;
    if (correct_out != nullptr) *correct_out = h;
    if (global_step != nullptr) *global_step += 1.0f;
  }
}

}  // namespace

// Timing experiments on the UNPINNED chunk (see GemmArgs::parts); 15 restores the kernel.
extern "C" void dtf_debug_set_gemm_parts(int parts) { g_parts = parts; }

extern "C" void dtf_debug_set_gemm_split(int max_tiles, int target_workgroups) {
  g_split_max_tiles = max_tiles;
  g_split_target = target_workgroups;
}

extern "C" int dtf_gemm_f32(int layout, int epi, const float* A, const float* B, float* C, int M, int N, int K,
                            int64_t lda, int64_t ldb, int64_t ldc, int act, const float* bias, const float* aux,
                            int64_t ldaux, const float* lr, float scale, int tile, void* stream) {
  return dtf_gemm_f32_ws(layout, epi, A, B, C, M, N, K, lda, ldb, ldc, act, bias, aux, ldaux, lr, scale, tile, nullptr, 0,
                         0, stream);
}

extern "C" int dtf_gemm_f32_ws(int layout, int epi, const float* A, const float* B, float* C, int M, int N, int K,
                               int64_t lda, int64_t ldb, int64_t ldc, int act, const float* bias, const float* aux,
                               int64_t ldaux, const float* lr, float scale, int tile, float* workspace,
                               int64_t workspace_floats, int splits, void* stream) {
  if (A == nullptr || B == nullptr || C == nullptr) return DTF_ERR_NULL;
  if (M < 1 || N < 1 || K < 1 || ldc < N) return DTF_ERR_SHAPE;
  if (layout < DTF_GEMM_NN || layout > DTF_GEMM_TN || epi < DTF_EPI_NONE || epi > DTF_EPI_SGD) return DTF_ERR_SHAPE;
  if ((layout == DTF_GEMM_TN ? lda < M : lda < K) || (layout == DTF_GEMM_NT ? ldb < K : ldb < N)) return DTF_ERR_SHAPE;
  if (epi == DTF_EPI_ACT_BWD && (aux == nullptr || ldaux < N)) return DTF_ERR_NULL;
  if (epi == DTF_EPI_SGD && lr == nullptr) return DTF_ERR_NULL;
  if (tile != 0 && tile != 32 && tile != 64 && tile != 128 && tile != 129) return DTF_ERR_SHAPE;
  // byte offsets inside a tile (k along memory: 128 rows; k strided: the 32 rows of a chunk) are 32-bit
  if ((128 * lda + K) * 4 >= (1LL << 31) || (128 * ldb + K) * 4 >= (1LL << 31)) return DTF_ERR_UNSUPPORTED;
  GemmArgs a{};
  a.A = A; a.B = B; a.C = C;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.epi = epi; a.act = act;
  a.bias = bias; a.aux = aux; a.ldaux = ldaux;
  a.lr = lr; a.scale = scale;
  a.ws = workspace_floats > 1024 ? workspace : nullptr;
  a.ws_floats = workspace_floats;
  a.splits = splits;
  a.parts = g_parts;
  // 16-byte loads along k need K, the leading dimension and the base address to be multiples of four floats (a
  // compile-time variant: every operand of the layout whose k runs along memory must qualify)
  const bool vecA = (K & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15u) == 0;
  const bool vecB = (K & 3) == 0 && (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15u) == 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (layout == DTF_GEMM_NN) return vecA ? launch_layout<DTF_GEMM_NN, true>(a, tile, s) : launch_layout<DTF_GEMM_NN, false>(a, tile, s);
  if (layout == DTF_GEMM_NT)
    return vecA && vecB ? launch_layout<DTF_GEMM_NT, true>(a, tile, s) : launch_layout<DTF_GEMM_NT, false>(a, tile, s);
  return launch_layout<DTF_GEMM_TN, false>(a, tile, s);
}

extern "C" int dtf_colsum(const float* z, int64_t ldz, int M, int N, float* out, int mode, float scale, const float* lr,
                          void* stream) {
  if (z == nullptr || out == nullptr) return DTF_ERR_NULL;
  if (M < 1 || N < 1 || ldz < N) return DTF_ERR_SHAPE;
  if (mode != DTF_EPI_NONE && mode != DTF_EPI_SGD) return DTF_ERR_SHAPE;
  if (mode == DTF_EPI_SGD && lr == nullptr) return DTF_ERR_NULL;
  const int segments = (M + kColsumRows - 1) / kColsumRows;
  if (segments > 1 && mode == DTF_EPI_NONE && dtf_launch_zero(out, N, static_cast<hipStream_t>(stream)) != DTF_OK)
    return DTF_ERR_LAUNCH;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, segments), dim3(1024), 0, static_cast<hipStream_t>(stream), z,
                     (long)ldz, M, N, out, mode, scale, lr);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_colsum_multi(int count, const uint64_t* z, const int64_t* ldz, const int32_t* M, const int32_t* N,
                                const uint64_t* out, int mode, float scale, const float* lr, void* stream) {
  if (count < 1 || count > 16 || z == nullptr || ldz == nullptr || M == nullptr || N == nullptr || out == nullptr)
    return DTF_ERR_SHAPE;
  if (mode != DTF_EPI_NONE && mode != DTF_EPI_SGD) return DTF_ERR_SHAPE;
  if (mode == DTF_EPI_SGD && lr == nullptr) return DTF_ERR_NULL;
  ColsumTable t{};
  t.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (z[i] == 0 || out[i] == 0) return DTF_ERR_NULL;
    if (M[i] < 1 || N[i] < 1 || ldz[i] < N[i]) return DTF_ERR_SHAPE;
    t.z[i] = reinterpret_cast<const float*>(z[i]);
    t.out[i] = reinterpret_cast<float*>(out[i]);
    t.ldz[i] = ldz[i];
    t.M[i] = M[i];
    t.N[i] = N[i];
    t.first_block[i] = blocks;
    blocks += (N[i] + 63) / 64;
  }
  t.first_block[count] = blocks;
  hipLaunchKernelGGL(colsum_multi_kernel, dim3(blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), t, mode, scale, lr);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_softmax_xent_step(const float* logits, const float* labels, int M, int C, float* dz, float grad_scale,
                                     float* loss_out, int32_t* correct_out, float* global_step, void* stream) {
  if (logits == nullptr || labels == nullptr) return DTF_ERR_NULL;
  if (M < 1 || C < 1 || M > 8192) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(softmax_xent_step_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), logits, labels, M,
                     C, dz, grad_scale, loss_out, correct_out, global_step);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_dense_step_begin(float* loss_sum, int32_t* correct, float* global_step, void* stream) {
  hipLaunchKernelGGL(dense_step_begin_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), loss_sum, correct,
                     global_step);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_softmax_xent_head(const float* logits, const float* labels, int M, int C, float* dz, float grad_scale,
                                     float* scratch, int64_t scratch_floats, float* loss_out, int32_t* correct_out,
                                     float* global_step, void* stream) {
  if (logits == nullptr || labels == nullptr || scratch == nullptr) return DTF_ERR_NULL;
  if (M < 1 || C < 1 || M > (1 << 24)) return DTF_ERR_SHAPE;          // (counts travel as floats: exact up to 2^24)
  const int blocks = (M + 3) / 4;
  if (scratch_floats < 2 * (int64_t)blocks + 1) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(softmax_xent_head_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), logits, labels,
                     M, C, dz, grad_scale, scratch, loss_out, correct_out, global_step);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_softmax_xent(const float* logits, const float* labels, int M, int C, float* dz, float grad_scale,
                                float* loss_sum, int32_t* correct, float* row_loss, void* stream) {
  if (logits == nullptr || labels == nullptr) return DTF_ERR_NULL;
  if (M < 1 || C < 1) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(softmax_xent_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                     labels, M, C, dz, grad_scale, loss_sum, correct, row_loss);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}
