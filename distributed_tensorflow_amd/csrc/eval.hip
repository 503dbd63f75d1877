// Evaluation of the two-layer MLP on a large set in ONE kernel:
//
//   loss_sum += sum_r CE(softmax(act(x_r W1 + b1) W2 + b2), y_r),   correct += #{r : argmax logits_r == argmax y_r}
//
// (the accuracy / cost fetch on mnist.test at the end of every epoch, tfdist_between.py:59-71,110 and
// tfsingle.py:37-49,73).  x is read once, the hidden activations never leave the CU.
//
// Work split.  A workgroup of 8 waves owns 48 rows of x and ALL hidden units: a [48 x 112 x K] product on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32; H <= 112 = 7 MFMA columns), 209 workgroups for the 10,000 test rows, one per
// CU.  K is walked in chunks of 64 floats; inside a chunk wave w contracts k-quads 4 (w & 3) .. + 3 (one 16-k
// macro-step) for all 3 block rows and for block columns 0-3 (w < 4) or 4-6 (w >= 4): the two waves of a SIMD hold the
// two column groups of the same k range, so the four SIMDs carry equal work, and a macro-step costs a wave 3 + 4 (or
// 3 + 3) LDS reads of 16 bytes for 48 (36) MFMAs.  The four k-partial sums of every output meet in LDS at the end.
//
// Staging.  LDS holds three chunk images of both operands in [k-quad][row or column][4 floats] form, so that lane
// (i = l & 15, q = l >> 4) reads k-quad q of its macro-step for row / column i as one ds_read_b128 (16 distinct
// 16-byte slots per 16-lane group: no bank conflict) and MFMA s contracts k = 16 ms + 4 q + s.  x rows arrive as
// 16-byte loads (8 rows x 8 k-quads per wave-instruction: every 128-byte line is used whole); W1 is [K][H], k-major,
// so a lane loads the four k of a quad for one column as four coalesced dwords and writes them as one 16-byte LDS
// store.  While chunk c is multiplied (from registers), the operand registers of chunk c+1 are read from LDS, chunk
// c+2 is written to the third image and the loads of chunk c+3 are in flight; one barrier per chunk.
//
// Epilogue.  bias + activation into a [48][116] LDS image, the [48 x 112] x [112 x 16] second layer on the matrix
// cores (waves 0-2, one block row each), softmax / cross-entropy / argmax inside 16-lane groups, one pair of atomics
// per workgroup.
#include "dtf_api.h"
#include "dtf_common.h"
#include "dtf_internal.h"
#include <atomic>

namespace {

constexpr int kThreads = 512;
constexpr int kRows = 48;                    // rows of x per workgroup (3 block rows)
constexpr int kCols = 112;                   // hidden units in the LDS image (7 block columns)
constexpr int kChunkQ = 16;                  // k-quads per chunk
constexpr int kXs = kChunkQ * kRows * 4;     // floats of one x chunk image
constexpr int kWs = kChunkQ * kCols * 4;     // floats of one W1 chunk image
constexpr int kTiles = 3 * 7;
constexpr int kRed = 4 * kTiles * 256;       // four k-partial images of the [48][112] outputs in (tile, reg, lane) order
constexpr int kLda = 116;                    // row stride of the activation image: lanes (i, q) of an MFMA operand read
                                             // hit 64 different banks (116 i + q mod 64 is a bijection)
constexpr int kA2 = kRows * kLda;
constexpr int kW2 = kCols * 16;
constexpr int kStage = 3 * (kXs + kWs);      // three chunk images of both operands
constexpr int kSmemFloats = kStage + kA2 + kW2 + kCols + 16 + 16;
static_assert(kRed <= kStage, "the k-partial images reuse the staging buffers");
static_assert(kSmemFloats * 4 <= 160 * 1024, "LDS of one CU");

struct EvalArgs {
  const float *x, *y, *W1, *b1, *W2, *b2;
  float* loss_sum;
  int* correct;
  int M, K, H, C, act;
  long long* stamps;       // timing experiments: wall-clock stamps of workgroup 100 (null: none)
  int parts;               // timing experiments: bit 0 global loads per chunk, bit 1 products (3: all)
};

template <int NCT>
__device__ __forceinline__ void read_operands(const float* Xb, const float* Wb, int kq, int ct0, int i16, f32x4 (&av)[3],
                                              f32x4 (&bv)[4]) {
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) av[rt] = *reinterpret_cast<const f32x4*>(Xb + (kq * kRows + 16 * rt + i16) * 4);
#pragma unroll
  for (int j = 0; j < NCT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(Wb + (kq * kCols + 16 * (ct0 + j) + i16) * 4);
}

template <int NCT, int S0, int S1>
__device__ __forceinline__ void multiply(const f32x4 (&av)[3], const f32x4 (&bv)[4], f32x4 (&acc)[3][4]) {
#pragma unroll
  for (int s = S0; s < S1; ++s)
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
      for (int j = 0; j < NCT; ++j)
        acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][s], bv[j][s], acc[rt][j], 0, 0, 0);
}

// What one thread moves per chunk: two 16-byte pieces of x rows and four (k-quad, column) slots of W1 (four dwords
// each), the same count in every wave so that the chunk loop is straight-line code.  The 12 + 28 wave-instructions a
// chunk needs are 0-3 and 8-11 (x) and W1 groups 4-27 for waves 0-3 plus their share, 4-7 (x) and W1 groups 0-3 for
// waves 4-7; the spare sixth slot of a wave repeats one of its own (same data to the same LDS address).
struct Stager {
  const float* xrow[2];
  const float* wcolp[4];   // W1 + column
  int xlds[2], xkq[2];
  int wkq[4], wlds[4];
  bool wcol_ok[4];
  f32x4 sx[2], sw[4];
  int nkq, H;

  __device__ __forceinline__ void issue(int c) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int kq = c * kChunkQ + xkq[u];
      kq = kq < nkq ? kq : 0;
      sx[u] = *reinterpret_cast<const f32x4*>(xrow[u] + 4 * kq);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int kq = c * kChunkQ + wkq[j];
      kq = kq < nkq ? kq : 0;
      const float* p = wcolp[j] + (long)(4 * kq) * H;
      sw[j][0] = p[0];
      sw[j][1] = p[H];
      sw[j][2] = p[2 * H];
      sw[j][3] = p[3 * H];
    }
  }
  __device__ __forceinline__ void write(int c, float* Xb, float* Wb) const {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
      *reinterpret_cast<f32x4*>(Xb + xlds[u]) = (c * kChunkQ + xkq[u] < nkq) ? sx[u] : zero4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(Wb + wlds[j]) = (wcol_ok[j] && c * kChunkQ + wkq[j] < nkq) ? sw[j] : zero4;
  }
};

// The instruction order of one steady-state chunk, pinned for the scheduler: the LDS stores of chunk c+2 (whose loads
// were requested a whole chunk ago) go first, then the loads of chunk c+3, then the LDS reads of chunk c+1's operands,
// each group spread between the MFMAs of chunk c, so that neither the LDS pipe nor the vector memory unit is used in
// a burst while the matrix cores idle.
template <int NCT>
__device__ __forceinline__ void pin_chunk_order() {
  constexpr int kMfma = 0x008, kVmemRead = 0x020, kDsRead = 0x100, kDsWrite = 0x200;
  constexpr int total = 12 * NCT;
  constexpr int per_write = 2, per_vmem_pair = 1, per_read = NCT == 4 ? 3 : 2;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    __builtin_amdgcn_sched_group_barrier(kDsWrite, 1, 0);
    __builtin_amdgcn_sched_group_barrier(kMfma, per_write, 0);
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    __builtin_amdgcn_sched_group_barrier(kVmemRead, 2, 0);
    __builtin_amdgcn_sched_group_barrier(kMfma, per_vmem_pair, 0);
  }
#pragma unroll
  for (int i = 0; i < 3 + NCT; ++i) {
    __builtin_amdgcn_sched_group_barrier(kDsRead, 1, 0);
    __builtin_amdgcn_sched_group_barrier(kMfma, per_read, 0);
  }
  __builtin_amdgcn_sched_group_barrier(kMfma, total - 6 * per_write - 9 * per_vmem_pair - (3 + NCT) * per_read, 0);
}

// The K loop of one wave: NCT block columns starting at ct0.  Chunk c lives in image c % 3.
template <int NCT>
__device__ __forceinline__ void k_loop(Stager& st, float* Xs, float* Ws, int nchunks, int kq_mine, int ct0, int i16,
                                       int parts, f32x4 (&acc)[3][4]) {
  f32x4 avA[3], bvA[4], avB[3], bvB[4];
  read_operands<NCT>(Xs, Ws, kq_mine, ct0, i16, avA, bvA);
  int b1 = 1, b2 = 2, b0 = 0;          // images of chunks c+1, c+2, c (rotated every chunk)
  int c = 0;
#define DTF_EVAL_FAST(av, bv, avn, bvn)                                                  \
  do {                                                                                   \
    st.write(c + 2, Xs + b2 * kXs, Ws + b2 * kWs);                                       \
    st.issue(c + 3);                                                                     \
    read_operands<NCT>(Xs + b1 * kXs, Ws + b1 * kWs, kq_mine, ct0, i16, avn, bvn);       \
    multiply<NCT, 0, 4>(av, bv, acc);                                                    \
    pin_chunk_order<NCT>();                                                              \
    __syncthreads();                                                                     \
    const int t_ = b0; b0 = b1; b1 = b2; b2 = t_;                                        \
    ++c;                                                                                 \
  } while (0)
  if (parts == 3)
    while (c + 4 < nchunks) {          // both chunks of a turn have a chunk c+3 to request
      DTF_EVAL_FAST(avA, bvA, avB, bvB);
      DTF_EVAL_FAST(avB, bvB, avA, bvA);
    }
#undef DTF_EVAL_FAST
  // the last chunks (and the timing experiments): the same steps under conditions, in program order
#define DTF_EVAL_SLOW(av, bv, avn, bvn)                                                  \
  do {                                                                                   \
    if (c + 1 < nchunks) read_operands<NCT>(Xs + b1 * kXs, Ws + b1 * kWs, kq_mine, ct0, i16, avn, bvn); \
    if (parts & 2) multiply<NCT, 0, 4>(av, bv, acc);                                     \
    if (c + 2 < nchunks) {                                                               \
      st.write(c + 2, Xs + b2 * kXs, Ws + b2 * kWs);                                     \
      if (c + 3 < nchunks && (parts & 1)) st.issue(c + 3);                               \
    }                                                                                    \
    __syncthreads();                                                                     \
    const int t_ = b0; b0 = b1; b1 = b2; b2 = t_;                                        \
    ++c;                                                                                 \
  } while (0)
  while (c < nchunks) {
    DTF_EVAL_SLOW(avA, bvA, avB, bvB);
    if (c < nchunks) DTF_EVAL_SLOW(avB, bvB, avA, bvA);
  }
#undef DTF_EVAL_SLOW
}

__global__ __launch_bounds__(kThreads) void eval_fused_kernel(EvalArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                       // [3][kXs]
  float* Ws = smem + 3 * kXs;             // [3][kWs]
  float* red = smem;                      // after the K loop
  float* a2s = smem + kStage;             // [48][kLda]
  float* W2s = a2s + kA2;                 // [112][16]
  float* b1s = W2s + kW2;                 // [112]
  float* b2s = b1s + kCols;               // [16]
  float* scal = b2s + 16;                 // [0] loss sum, [1] correct count

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
#define DTF_EVAL_STAMP(i)                                                                          \
  do {                                                                                             \
    if (a.stamps != nullptr && blockIdx.x == 100 && tid == 0) a.stamps[i] = (long long)wall_clock64(); \
  } while (0)
  DTF_EVAL_STAMP(0);
  const int M = a.M, K = a.K, H = a.H, C = a.C;
  const int nkq = K >> 2, nchunks = (nkq + kChunkQ - 1) / kChunkQ;
  const int r0 = blockIdx.x * kRows;

  // the small operands: requested here, stored to LDS after the first chunk's loads have been issued
  float w2r[(kW2 + kThreads - 1) / kThreads], b1r = 0.f, b2r = 0.f;
#pragma unroll
  for (int u = 0; u < (kW2 + kThreads - 1) / kThreads; ++u) {
    const int e = tid + u * kThreads, k = e >> 4, c = e & 15;
    const bool ok = e < kW2 && k < H && c < C;
    const float v = a.W2[ok ? k * C + c : 0];         // always a load: four requests in flight, not four round trips
    w2r[u] = ok ? v : 0.f;
  }
  if (tid < H) b1r = a.b1[tid];
  if (tid < C) b2r = a.b2[tid];
  // the labels of the rows whose losses this lane computes in the epilogue (waves 0-2: block row = wave)
  float yreg[4] = {0.f, 0.f, 0.f, 0.f};
  if (wave < 3 && i16 < C) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 16 * wave + 4 * q + r;
      if (row < M) yreg[r] = a.y[(long)row * C + i16];
    }
  }

  // ---- staging plan (see Stager) -----------------------------------------------------------------------------
  const int r8 = lane & 7, kql = lane >> 3;
  Stager st;
  st.nkq = nkq;
  st.H = H;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int wi = (u == 1 && wave >= 4) ? wave : wave + 8 * u;      // waves 4-7: the second piece repeats the first
    const int rg = wi >> 1, kg = wi & 1;
    int row = r0 + 8 * rg + r8;
    row = row < M ? row : M - 1;                      // rows past the end are masked in the epilogue
    st.xrow[u] = a.x + (long)row * K;
    st.xkq[u] = 8 * kg + kql;
    st.xlds[u] = (st.xkq[u] * kRows + 8 * rg + r8) * 4;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int wj = j == 0 ? wave - 4 : 8 * j - 4 + wave;    // 4 + wave, 12 + wave, 20 + wave
    if (wj < 0) wj = 4 + wave;                        // waves 0-3: the first slot repeats the second
    const int slot = wj * 64 + lane;
    st.wkq[j] = slot / kCols;
    const int col = slot - st.wkq[j] * kCols;
    st.wcol_ok[j] = col < H;
    st.wcolp[j] = a.W1 + (col < H ? col : H - 1);
    st.wlds[j] = slot * 4;
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  st.issue(0);
#pragma unroll
  for (int u = 0; u < (kW2 + kThreads - 1) / kThreads; ++u)
    if (tid + u * kThreads < kW2) W2s[tid + u * kThreads] = w2r[u];
  if (tid < kCols) b1s[tid] = b1r;
  if (tid < 16) b2s[tid] = b2r;
  if (tid < 2) scal[tid] = 0.f;
  DTF_EVAL_STAMP(1);
  st.write(0, Xs, Ws);
  if (nchunks > 1) {
    st.issue(1);
    st.write(1, Xs + kXs, Ws + kWs);
    if (nchunks > 2 && (a.parts & 1)) st.issue(2);
  }
  __syncthreads();
  DTF_EVAL_STAMP(2);

  const int kpart = wave & 3, cg = wave >> 2;
  const int kq_mine = 4 * kpart + q;
  f32x4 acc[3][4];
#pragma unroll
  for (int rt = 0; rt < 3; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[rt][j] = zero4;
  // a k-quad past the end of K holds zeros in LDS: the short last chunk needs no special case
  if (cg == 0)
    k_loop<4>(st, Xs, Ws, nchunks, kq_mine, 0, i16, a.parts, acc);
  else
    k_loop<3>(st, Xs, Ws, nchunks, kq_mine, 4, i16, a.parts, acc);

  DTF_EVAL_STAMP(3);
  // ---- the four k-partial sums of every output meet in LDS --------------------------------------------------
  {
    const int nct = cg == 0 ? 4 : 3, ct0 = cg == 0 ? 0 : 4;
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nct) {
          const int tile = rt * 7 + ct0 + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((kpart * kTiles + tile) * 4 + r) * 64 + lane] = acc[rt][j][r];
        }
  }
  __syncthreads();
  // four consecutive lanes of one (tile, register) image are four consecutive columns of one output row
#pragma unroll
  for (int u = 0; u < (kTiles * 64 + kThreads - 1) / kThreads; ++u) {
    const int e4 = tid + u * kThreads;
    if (e4 < kTiles * 64) {
      const f32x4* r4 = reinterpret_cast<const f32x4*>(red) + e4;
      const f32x4 s4 = (r4[0] + r4[kTiles * 64]) + (r4[2 * kTiles * 64] + r4[3 * kTiles * 64]);
      const int tile = e4 >> 6, r = (e4 >> 4) & 3, l = (e4 & 15) * 4;
      const int rt = tile / 7, ct = tile - rt * 7;
      const int row = 16 * rt + 4 * (l >> 4) + r, col = 16 * ct + (l & 15);
      f32x4 o;
#pragma unroll
      for (int v = 0; v < 4; ++v) o[v] = col + v < H ? dtf_act_fwd(s4[v] + b1s[col + v], a.act) : 0.f;
      *reinterpret_cast<f32x4*>(a2s + row * kLda + col) = o;
    }
  }
  __syncthreads();

  DTF_EVAL_STAMP(4);
  // ---- second layer, softmax cross-entropy, accuracy: one block row per wave (waves 0-2) ---------------------
  if (wave < 3) {
    f32x4 acc0 = zero4, acc1 = zero4;                 // two chains: MFMA latency exceeds its issue time
    const float* arow = a2s + (16 * wave + i16) * kLda + q;
    const float* brow = W2s + q * 16 + i16;
#pragma unroll
    for (int k = 0; k < kCols; k += 8) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k], brow[k * 16], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k + 4], brow[(k + 4) * 16], acc1, 0, 0, 0);
    }
    const f32x4 z4 = acc0 + acc1;
    const bool col_ok = i16 < C;
    const float bias2 = b2s[i16];
    float wloss = 0.f, wcorrect = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 16 * wave + 4 * q + r;
      const bool row_ok = row < M;
      const float z = z4[r] + bias2;
      const float zm = col_ok ? z : -INFINITY;
      const float mx = row16_max(zm);
      const float se = row16_sum(col_ok ? expf(z - mx) : 0.f);
      const float logp = z - mx - logf(se);
      const float y = yreg[r];                        // zero outside the valid rows and classes
      const float rloss = row16_sum(col_ok ? -y * logp : 0.f);
      const int az = row16_argmax(zm, i16);
      const int ay = row16_argmax(col_ok ? y : -INFINITY, i16);
      if (row_ok && i16 == 0) {
        wloss += rloss;
        wcorrect += az == ay ? 1.f : 0.f;
      }
    }
    // lanes 0, 16, 32, 48 carry the sums of their four rows
    wloss += __shfl_xor(wloss, 16, 64);
    wcorrect += __shfl_xor(wcorrect, 16, 64);
    wloss += __shfl_xor(wloss, 32, 64);
    wcorrect += __shfl_xor(wcorrect, 32, 64);
    if (lane == 0) {
      atomicAdd(&scal[0], wloss);
      atomicAdd(&scal[1], wcorrect);
    }
  }
  __syncthreads();
  DTF_EVAL_STAMP(5);
  if (tid == 0) {
    atomicAdd(a.loss_sum, scal[0]);
    atomicAdd(a.correct, (int)(scal[1] + 0.5f));
  }
  DTF_EVAL_STAMP(6);
#undef DTF_EVAL_STAMP
}

int g_eval_parts = 3;
long long* g_eval_stamps = nullptr;

}  // namespace

// Timing experiments: run only some parts of the kernel (results are then wrong); 3 restores the full kernel.
extern "C" void dtf_debug_set_eval_parts(int parts) { g_eval_parts = parts; }
extern "C" void dtf_debug_set_eval_stamps(long long* device_buffer) { g_eval_stamps = device_buffer; }

int dtf_eval_fused_supported(int M, int K, int H, int C) {
  return (M >= 1 && K >= 4 && (K & 3) == 0 && H >= 1 && H <= kCols && C >= 1 && C <= 16) ? 1 : 0;
}

// loss_sum / correct are accumulated into (the caller zeroes them).  x [M][K], y [M][C], 16-byte aligned x.
int dtf_launch_eval_fused(const float* x, const float* y, int M, int K, int H, int C, int act, const float* W1,
                          const float* b1, const float* W2, const float* b2, float* loss_sum, int* correct,
                          hipStream_t stream) {
  if (!dtf_eval_fused_supported(M, K, H, C) || (reinterpret_cast<uintptr_t>(x) & 15u)) return DTF_ERR_SHAPE;
  static std::atomic<unsigned long long> configured{0};    // one bit per device: the attribute belongs to the device's copy
  constexpr int bytes = kSmemFloats * (int)sizeof(float);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DTF_ERR_LAUNCH;
  if (dev >= 64 || !((configured.load(std::memory_order_acquire) >> dev) & 1ull)) {
    if (hipFuncSetAttribute((const void*)eval_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) !=
        hipSuccess)
      return DTF_ERR_LAUNCH;
    if (dev < 64) configured.fetch_or(1ull << dev, std::memory_order_release);
  }
  EvalArgs a{x, y, W1, b1, W2, b2, loss_sum, correct, M, K, H, C, act, g_eval_stamps, g_eval_parts};
  const int blocks = (M + kRows - 1) / kRows;
  hipLaunchKernelGGL(eval_fused_kernel, dim3(blocks), dim3(kThreads), bytes, stream, a);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}
