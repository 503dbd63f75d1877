// Two-launch training step for small batches (B <= 128, H <= 128, C <= 16):
//
//   launch 1  fwd_tile16   a2 = act(x W1 + b1)                      (F1-F3)
//   launch 2  bwd_all      logits, softmax-xent, accuracy, dz3, dW2, db2,
//                          dz2, db1, dW1 and (optionally) the SGD update
//                                                                   (F4-F12, G1-G7, U1-U2)
// (op numbers: the reference graph, tfdist_between.py:55-70).
//
// Why this shape.  A kernel boundary on MI355X costs ~1.5 us and one CU retires
// only ~128 fp32 lanes per clock, so at batch 100 both the NUMBER of launches and
// any work funnelled through a single workgroup dominate the step:
//   * the forward GEMM is tiled 16x16 (49 workgroups for 100x100) with the K
//     dimension split over the 16 waves of the workgroup and reduced in LDS —
//     no inter-workgroup atomics, deterministic, and the bias + activation are
//     spread over 49 CUs instead of one;
//   * a CU ingests only ~40-60 GB/s on a latency-bound stream, so no workgroup
//     may load more than a few tens of KB: the forward workgroups add their
//     share of the LOGITS (a [16 x 10] partial product) into a 6 KB accumulator
//     with f32 atomics, and every backward workgroup redoes the softmax /
//     cross-entropy gradient for the whole batch from those 6 KB instead of
//     waiting for a single-workgroup head kernel and another boundary;
//   * dz2 is never materialised — it is formed in registers as the B operand of
//     the dW1 MFMA:  dz2[m][h] = act'(a2[m][h]) * sum_c dz3[m][c] W2[h][c]
//   * dW2 tiles, db2, db1 and the step statistics are "duties" dealt to the
//     first workgroups of the same launch.
// Each kernel makes ONE global-memory round trip: all loads are issued before
// the first one is consumed.
//
// In-place SGD and the W2 race: the backward launch reads W2/b2 in every
// workgroup and (fused mode) also writes them.  Workgroups of one launch are
// not ordered, so the readers use a SHADOW copy of W2/b2 that workgroup 0 of
// the forward launch refreshes from the live parameters (the forward pass does
// not read W2).  Live parameters are therefore always current.
#include "dtf_api.h"
#include "dtf_internal.h"

namespace {

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_act(float z, int act) {
  if (act == DTF_ACT_SIGMOID) return fast_rcp(1.0f + fast_exp(-z));
  return fmaxf(z, 0.0f);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
constexpr int kXor1 = 0xB1, kXor2 = 0x4E, kHalfMirror = 0x141, kRowMirror = 0x140;

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<kXor1>(v));
  v = fmaxf(v, dpp_f<kXor2>(v));
  v = fmaxf(v, dpp_f<kHalfMirror>(v));
  return fmaxf(v, dpp_f<kRowMirror>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<kXor1>(v);
  v += dpp_f<kXor2>(v);
  v += dpp_f<kHalfMirror>(v);
  return v + dpp_f<kRowMirror>(v);
}
template <int CTRL>
__device__ __forceinline__ void argmax_step(float& v, int& idx) {
  const float ov = dpp_f<CTRL>(v);
  const int oi = dpp_i<CTRL>(idx);
  if (ov > v || (ov == v && oi < idx)) {
    v = ov;
    idx = oi;
  }
}
__device__ __forceinline__ int row16_argmax(float v, int idx) {  // ties -> smallest index
  argmax_step<kXor1>(v, idx);
  argmax_step<kXor2>(v, idx);
  argmax_step<kHalfMirror>(v, idx);
  argmax_step<kRowMirror>(v, idx);
  return idx;
}

// ---------------------------------------------------------------------------
// Forward: 16x16 output tile per workgroup, 16 waves split K.
// MFMA 16x16x4 maps (lane l, i = l & 15, q = l >> 4): A[i][k=q], B[k=q][j=i],
// D[row = 4q + reg][col = i].  A macro-step covers 16 k: lane (i, q) holds
// x[m0+i][16*ms + 4q + s] for s = 0..3 (one 16-B load) and MFMA s contracts
// k = 16*ms + 4q + s.
//
// Epilogue: besides a2 the workgroup adds its share of the logits,
//   z3[m][c] += sum_{h in tile} a2[m][h] W2[h][c],
// into the ping-pong accumulator zacc (7 workgroups per row tile, f32 atomics),
// and block 0 zeroes the OTHER ping-pong half for the next step.
constexpr int kFwdWaves = 16;
constexpr int kFwdMaxMs = 4;  // macro-steps per wave held in registers (K <= 1024)

__global__ __launch_bounds__(kFwdWaves * 64) void fwd_tile16_kernel(
    const float* __restrict__ xbase, long batch_index, const float* __restrict__ W,
    const float* __restrict__ b, const float* __restrict__ W2, float* __restrict__ out,
    float* __restrict__ zacc, float* __restrict__ zacc_other, int zacc_n, int M, int K, int H, int C,
    int act, float* __restrict__ shadow_dst, const float* __restrict__ b2) {
  __shared__ float red[kFwdWaves][4][64];
  __shared__ float a2t[16][17];
  __shared__ float w2t[16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, q = lane >> 4;
  const int tiles_n = (H + 15) >> 4;
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int m0 = tm * 16, n0 = tn * 16;
  const float* __restrict__ X = xbase + batch_index * (long)M * (long)K;
  const int nms = (K + 15) >> 4;
  const int row = m0 + i16, col = n0 + i16;
  const bool row_ok = row < M, col_ok = col < H;
  const bool vec_ok = (K & 3) == 0;

  // epilogue operands, requested up front (threads < 256 own one tile element each)
  const int e_reg = threadIdx.x >> 6, e_lane = threadIdx.x & 63;
  const int e_col = n0 + (e_lane & 15);
  const float bias = (threadIdx.x < 256 && e_col < H) ? b[e_col] : 0.f;
  const int w_h = threadIdx.x >> 4, w_c = threadIdx.x & 15;   // threads < 256: W2 tile element
  const float w2v = (threadIdx.x < 256 && n0 + w_h < H && w_c < C) ? W2[(n0 + w_h) * C + w_c] : 0.f;

  // block 0: refresh the shadow copy of W2/b2 and clear the other logit accumulator
  if (blockIdx.x == 0) {
    if (shadow_dst) {
      for (int e = threadIdx.x; e < H * C + C; e += kFwdWaves * 64)
        shadow_dst[e] = e < H * C ? W2[e] : b2[e - H * C];
    }
    for (int e = threadIdx.x; e < zacc_n; e += kFwdWaves * 64) zacc_other[e] = 0.f;
  }

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int base = wave; base < nms; base += kFwdWaves * kFwdMaxMs) {
    f32x4 a[kFwdMaxMs], bv[kFwdMaxMs];
#pragma unroll
    for (int j = 0; j < kFwdMaxMs; ++j) {
      const int ms = base + j * kFwdWaves;
      const int kb = ms * 16 + q * 4;
      a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ms < nms) {
        if (row_ok) {
          const float* px = X + (long)row * K + kb;
          if (vec_ok && kb + 3 < K) {
            a[j] = *reinterpret_cast<const f32x4*>(px);
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              if (kb + s < K) a[j][s] = px[s];
          }
        }
        if (col_ok) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (kb + s < K) bv[j][s] = W[(long)(kb + s) * H + col];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kFwdMaxMs; ++j) {
      if (base + j * kFwdWaves < nms) {  // wave-uniform
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][0], bv[j][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][1], bv[j][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][2], bv[j][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][3], bv[j][3], acc1, 0, 0, 0);
      }
    }
  }
  const f32x4 acc = acc0 + acc1;
#pragma unroll
  for (int g = 0; g < 4; ++g) red[wave][g][lane] = acc[g];
  if (threadIdx.x < 256) w2t[w_h][w_c] = w2v;
  __syncthreads();
  if (threadIdx.x < 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kFwdWaves; ++w) s += red[w][e_reg][e_lane];
    const int lrow = 4 * (e_lane >> 4) + e_reg;
    const int orow = m0 + lrow;
    float a = 0.f;
    if (orow < M && e_col < H) {
      a = fast_act(s + bias, act);
      out[(long)orow * H + e_col] = a;
    }
    a2t[lrow][e_lane & 15] = a;
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int lrow = threadIdx.x >> 4, c = threadIdx.x & 15;
    if (m0 + lrow < M && c < C) {
      float z = 0.f;
#pragma unroll
      for (int h = 0; h < 16; ++h) z += a2t[lrow][h] * w2t[h][c];
      atomicAdd(&zacc[(m0 + lrow) * 16 + c], z);
    }
  }
}

// ---------------------------------------------------------------------------
// Backward: one 32x32 tile of dW1 per workgroup, head recomputed from the logits.
constexpr int kBwdWaves = 8;
constexpr int kBwdThreads = kBwdWaves * 64;
constexpr int kMaxSteps = 8;          // ceil(ceil(128/2)/8) batch-row pairs per wave
constexpr int kLdw = 17;              // leading dimension of LDS tiles read down a column

struct BwdParams {
  const float* a2;        // [M,H] activations from the forward launch
  const float* zacc;      // [R,16] logits without bias
  const float* xbase;
  const float* ybase;
  long batch_index;
  const float* W2r;       // W2 to READ (shadow in fused mode, live otherwise) [H,C]
  const float* b2r;       // b2 to READ [C]
  float* oW1;             // fused: live W1; grad: gW1
  float* ob1;             // fused: live b1; grad: gb1
  float* oW2;             // fused: live W2; grad: gW2
  float* ob2;             // fused: live b2; grad: gb2
  const float* lr;
  float grad_scale;
  float* loss_hist;
  float* acc_hist;
  int hist_slot;
  float* global_step;
  int M, K, H, C, R, act, fused;
  long long* stamps;      // optional phase stamps (profiling)
};

#define DTF_STAMP2(i)                                                          \
  do {                                                                         \
    if (p.stamps && threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) {         \
      p.stamps[2 * (i)] = (long long)wall_clock64();                           \
      p.stamps[2 * (i) + 1] = (long long)clock64();                            \
    }                                                                          \
  } while (0)

__global__ __launch_bounds__(kBwdThreads) void bwd_all_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float dz3s[128 * 16];   // [R][16]
  __shared__ float red[kBwdWaves * 16 * 64];                      // partial tiles / scratch
  __shared__ float w2t[32 * kLdw];                                // W2 rows of this h tile
  __shared__ float a2d[128 * kLdw];                               // dW2 duty: a2 column tile
  __shared__ float b2s[16];
  __shared__ float scal[4];

  const int R = p.R, H = p.H, C = p.C, M = p.M, K = p.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, q = lane >> 4;
  const int r32 = lane & 31, half = lane >> 5;
  const int tiles_h = (H + 31) >> 5;
  const int th = blockIdx.x % tiles_h, tk = blockIdx.x / tiles_h;
  const int k0 = tk * 32, h0 = th * 32;
  const int h_tiles = (H + 15) >> 4;
  const int n_duties = h_tiles + 1;
  const bool stats_block = blockIdx.x == (unsigned)(h_tiles % gridDim.x);

  DTF_STAMP2(0);
  // ---- phase 0: issue every global load -------------------------------------
  const float* __restrict__ X = p.xbase + p.batch_index * (long)M * (long)K;
  const float* __restrict__ Y = p.ybase + p.batch_index * (long)M * (long)C;
  const float lr = p.fused ? *p.lr : 0.f;

  // logits + labels, flat [row][16] image: thread element e = tid + u*512
  float zreg[4], yreg[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + u * kBwdThreads;
    const int row = e >> 4, c = e & 15;
    const bool ok = row < M && c < C;
    zreg[u] = ok ? p.zacc[e] : 0.f;
    yreg[u] = ok ? Y[(long)row * C + c] : 0.f;
  }
  const float b2reg = tid < C ? p.b2r[tid] : 0.f;
  // W2 rows of this h tile
  const int wt_h = tid >> 4, wt_c = tid & 15;     // 512 threads = 32 rows x 16
  const float w2reg = (h0 + wt_h < H && wt_c < C) ? p.W2r[(h0 + wt_h) * C + wt_c] : 0.f;
  // dW1 operands in MFMA layout: wave w owns batch-row pairs w, w+8, ...
  const int nsteps = (M + 1) >> 1;
  const int kk = k0 + r32, hh = h0 + r32;
  float xreg[kMaxSteps], areg[kMaxSteps];
#pragma unroll
  for (int j = 0; j < kMaxSteps; ++j) {
    const int m = 2 * (wave + j * kBwdWaves) + half;
    xreg[j] = (m < M && kk < K) ? X[(long)m * K + kk] : 0.f;
    areg[j] = (m < M && hh < H) ? p.a2[(long)m * H + hh] : 0.f;
  }
  // old W1 values of this thread's two output elements (fused update)
  float wold[2] = {0.f, 0.f};
  if (p.fused) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * kBwdThreads;
      const int g = e >> 6, l = e & 63;
      const int orow = k0 + (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
      const int ocol = h0 + (l & 31);
      wold[j] = (orow < K && ocol < H) ? p.oW1[(long)orow * H + ocol] : 0.f;
    }
  }
  const float b1old = (p.fused && tk == 0 && tid < 32 && h0 + tid < H) ? p.ob1[h0 + tid] : 0.f;
  // dW2 duty: a2 column tile [R][16] (first duty of this block, if any)
  const int duty0 = blockIdx.x < (unsigned)h_tiles ? (int)blockIdx.x : -1;
  float dreg[4] = {0.f, 0.f, 0.f, 0.f};
  float w2old = 0.f;   // old value of this thread's dW2 duty element (threads < 256)
  if (duty0 >= 0 && tid < 256) {
    const int h = duty0 * 16 + 4 * ((tid & 63) >> 4) + (tid >> 6), c = tid & 15;
    w2old = (h < H && c < C) ? p.W2r[h * C + c] : 0.f;
  }
  if (duty0 >= 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * kBwdThreads;
      const int row = e >> 4, hcol = duty0 * 16 + (e & 15);
      dreg[u] = (row < M && hcol < H) ? p.a2[(long)row * H + hcol] : 0.f;
    }
  }

  DTF_STAMP2(1);
  // ---- phase 1: softmax-xent on the flat image (one 16-lane DPP row per batch row) ---
  if (tid < 16) b2s[tid] = b2reg;
  if (tid < 4) scal[tid] = 0.f;
  w2t[wt_h * kLdw + wt_c] = w2reg;
  if (duty0 >= 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * kBwdThreads;
      if (e < R * 16) a2d[(e >> 4) * kLdw + (e & 15)] = dreg[u];
    }
  }
  __syncthreads();
  const float inv_batch = 1.0f / (float)M;
  {
    const bool col_ok = i16 < C;
    const float bias2 = b2s[i16];
    float wloss = 0.f, wcorrect = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * kBwdThreads;
      const int row = e >> 4;
      const bool row_ok = row < M;
      const float z = zreg[u] + bias2;
      const float zm = col_ok ? z : -INFINITY;
      const float mx = row16_max(zm);
      const float ex = col_ok ? fast_exp(z - mx) : 0.f;
      const float se = row16_sum(ex);
      const float prob = ex * fast_rcp(se);
      const float y = yreg[u];
      const float ysum = row16_sum(y);
      if (e < R * 16) dz3s[e] = (row_ok && col_ok) ? (prob * ysum - y) * inv_batch : 0.f;
      if (stats_block) {   // block-uniform
        const float logp = z - mx - fast_log(se);
        const float rloss = row16_sum(col_ok ? -y * logp : 0.f);
        const int az = row16_argmax(zm, i16);
        const int ay = row16_argmax(col_ok ? y : -INFINITY, i16);
        if (row_ok && i16 == 0) {
          wloss += rloss;
          wcorrect += (az == ay) ? 1.f : 0.f;
        }
      }
    }
    if (stats_block) {
      wloss += __shfl_xor(wloss, 16);
      wloss += __shfl_xor(wloss, 32);
      wcorrect += __shfl_xor(wcorrect, 16);
      wcorrect += __shfl_xor(wcorrect, 32);
      if (lane == 0) {
        atomicAdd(&scal[0], wloss);
        atomicAdd(&scal[1], wcorrect);
      }
    }
  }
  __syncthreads();
  DTF_STAMP2(2);

  // ---- phase 2: dW1 tile; dz2 is formed in registers as the MFMA B operand --------------
  float w2row[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) w2row[c] = w2t[r32 * kLdw + c];
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float db1part = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxSteps; ++j) {
    const int step = wave + j * kBwdWaves;
    if (step < nsteps) {   // wave-uniform
      const int m = 2 * step + half;          // < R (R = M rounded up to 16)
      const f32x4* d4 = reinterpret_cast<const f32x4*>(dz3s + m * 16);   // broadcast reads
      const f32x4 d0 = d4[0], d1 = d4[1], d2 = d4[2], d3 = d4[3];
      float da = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        da += d0[c] * w2row[c];
        da += d1[c] * w2row[4 + c];
        da += d2[c] * w2row[8 + c];
        da += d3[c] * w2row[12 + c];
      }
      const float dz2 = da * dtf_act_bwd(areg[j], p.act);   // areg is 0 outside the valid range
      db1part += dz2;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xreg[j], dz2, acc, 0, 0, 0);
    }
  }
  DTF_STAMP2(3);
#pragma unroll
  for (int g = 0; g < 16; ++g) red[(wave * 16 + g) * 64 + lane] = acc[g];
  __syncthreads();
  DTF_STAMP2(4);
  const float gs = p.grad_scale;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + j * kBwdThreads;
    const int g = e >> 6, l = e & 63;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kBwdWaves; ++w) s += red[(w * 16 + g) * 64 + l];
    const int orow = k0 + (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
    const int ocol = h0 + (l & 31);
    if (orow < K && ocol < H) p.oW1[(long)orow * H + ocol] = p.fused ? wold[j] - lr * s : gs * s;
  }
  DTF_STAMP2(5);

  // ---- duty: db1 (workgroups of k-tile 0 see every batch row of their columns) -----
  if (tk == 0) {
    __syncthreads();   // red is reused
    db1part += __shfl_xor(db1part, 32);
    if (lane < 32) red[wave * 32 + lane] = db1part;
    __syncthreads();
    if (tid < 32 && h0 + tid < H) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kBwdWaves; ++w) s += red[w * 32 + tid];
      p.ob1[h0 + tid] = p.fused ? b1old - lr * s : gs * s;
    }
  }
  DTF_STAMP2(6);

  // ---- duties dealt round-robin over the grid: dW2 tiles, then db2 + statistics -------
  for (int duty = blockIdx.x; duty < n_duties; duty += gridDim.x) {
    __syncthreads();   // red / a2d are reused
    if (duty < h_tiles) {
      if (duty != duty0) {   // grids smaller than the duty list: stage this tile now
        for (int e = tid; e < R * 16; e += kBwdThreads) {
          const int row = e >> 4, hcol = duty * 16 + (e & 15);
          a2d[row * kLdw + (e & 15)] = (row < M && hcol < H) ? p.a2[(long)row * H + hcol] : 0.f;
        }
        __syncthreads();
      }
      // dW2[h][c] = sum_b a2[b][h] dz3[b][c] for h in tile `duty`; waves split the batch
      const float* arow = a2d + q * kLdw + i16;
      const float* brow = dz3s + q * 16 + i16;
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      for (int kb = wave * 4; kb < R; kb += kBwdWaves * 4)
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[kb * kLdw], brow[kb * 16], d, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) red[(wave * 4 + g) * 64 + lane] = d[g];
      __syncthreads();
      if (tid < 256) {
        const int g = tid >> 6, l = tid & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kBwdWaves; ++w) s += red[(w * 4 + g) * 64 + l];
        const int h = duty * 16 + 4 * (l >> 4) + g, c = l & 15;
        if (h < H && c < C) {
          // fused: old value from the read-side copy (never the live tensor other blocks may be writing)
          const float old = (duty == duty0) ? w2old : p.W2r[h * C + c];
          p.oW2[h * C + c] = p.fused ? old - lr * s : gs * s;
        }
      }
    } else {
      if (tid < C) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += dz3s[r * 16 + tid];
        p.ob2[tid] = p.fused ? b2s[tid] - lr * s : gs * s;
      }
      if (tid == 0) {
        if (p.loss_hist) p.loss_hist[p.hist_slot] = scal[0] * inv_batch;
        if (p.acc_hist) p.acc_hist[p.hist_slot] = scal[1] * inv_batch;
        if (p.fused && p.global_step) *p.global_step = *p.global_step + 1.0f;
      }
    }
  }
  DTF_STAMP2(7);
}

long long* g_stamps2 = nullptr;

}  // namespace

bool dtf_step2_supported(int B, int K, int H, int C) {
  return B >= 1 && B <= 128 && H >= 1 && H <= 128 && C >= 1 && C <= 16 && K >= 1;
}

// mode: DTF_HEAD_FUSED (in-place SGD) or DTF_HEAD_GRAD (gradients * grad_scale).
// parity selects the half of the logit ping-pong workspace this step accumulates
// into; consecutive steps MUST alternate it (the other half is cleared for the
// next step by the forward launch).
int dtf_launch_step2(const DtfMlpDesc* d, long batch_index, int parity, int mode, float grad_scale,
                     hipStream_t stream) {
  const int B = d->batch, K = d->in_dim, H = d->hidden, C = d->classes;
  if (!dtf_step2_supported(B, K, H, C)) return DTF_ERR_UNSUPPORTED;
  const bool fused = mode == DTF_HEAD_FUSED;
  if (!d->zws || (fused && (!d->shadow || !d->lr))) return DTF_ERR_NULL;
  if (d->batches_per_epoch > 0 && (batch_index < 0 || batch_index >= d->batches_per_epoch))
    return DTF_ERR_SHAPE;
  const int R = ((B + 15) / 16) * 16;
  float* zacc = d->zws + (parity & 1) * 2048;
  float* zother = d->zws + ((parity & 1) ^ 1) * 2048;
  const int tiles_m = (B + 15) / 16, tiles_n = (H + 15) / 16;
  hipLaunchKernelGGL(fwd_tile16_kernel, dim3(tiles_m * tiles_n), dim3(kFwdWaves * 64), 0, stream, d->x,
                     batch_index, d->W1, d->b1, d->W2, d->z2, zacc, zother, R * 16, B, K, H, C, d->act,
                     fused ? d->shadow : nullptr, d->b2);
  BwdParams p;
  p.a2 = d->z2;
  p.zacc = zacc;
  p.xbase = d->x;
  p.ybase = d->y;
  p.batch_index = batch_index;
  p.W2r = fused ? d->shadow : d->W2;
  p.b2r = fused ? d->shadow + H * C : d->b2;
  p.oW1 = fused ? d->W1 : d->gW1;
  p.ob1 = fused ? d->b1 : d->gb1;
  p.oW2 = fused ? d->W2 : d->gW2;
  p.ob2 = fused ? d->b2 : d->gb2;
  p.lr = d->lr;
  p.grad_scale = grad_scale;
  p.loss_hist = d->loss_hist;
  p.acc_hist = d->acc_hist;
  p.hist_slot = d->hist_cap > 0 ? (int)(batch_index % d->hist_cap) : 0;
  if (d->hist_cap <= 0) {
    p.loss_hist = nullptr;
    p.acc_hist = nullptr;
  }
  p.global_step = d->global_step;
  p.M = B; p.K = K; p.H = H; p.C = C; p.R = R; p.act = d->act; p.fused = fused ? 1 : 0;
  p.stamps = g_stamps2;
  const int tiles_k = (K + 31) / 32, tiles_h = (H + 31) / 32;
  hipLaunchKernelGGL(bwd_all_kernel, dim3(tiles_k * tiles_h), dim3(kBwdThreads), 0, stream, p);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

// Profiling hook: device buffer of >= 32 int64 receiving {wall_clock64, clock64}
// at the phase boundaries of the LAST workgroup of bwd_all; null disables.
extern "C" void dtf_debug_set_bwd_stamps(long long* device_buffer) { g_stamps2 = device_buffer; }
