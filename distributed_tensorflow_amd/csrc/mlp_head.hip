// Output head of the MLP in ONE kernel: hidden activation, second linear
// layer, softmax cross-entropy, batch accuracy, the closed-form logit gradient
// and the whole backward pass down to dZ2 — ops F4-F12 and G1-G5, G7 of the
// reference graph (tfdist_between.py:57-62, 68-70 and the autodiff of :65-66),
// ~25 TensorFlow kernel launches collapsed into one workgroup's LDS.
//
// A workgroup owns a block of R batch rows (R multiple of 16).  For the
// training batch of 100 a single workgroup does everything, which also lets
// it apply SGD to W2/b2/b1 in place (mode FUSED) without another launch.
//
// All three small GEMMs run on v_mfma_f32_16x16x4_f32 out of LDS:
//   z3  [R,16]   = a2 [R,H]  * W2 [H,16]          (K = H)
//   dW2 [H,16]   = a2^T [H,R] * dz3 [R,16]        (K = R)
//   da2 [R,H]    = dz3 [R,16] * W2^T [16,H]       (K = 16 >= C)
// MFMA maps (lane l, i = l & 15, q = l >> 4): A[i][k = q], B[k = q][j = i],
// D[row = 4*q + reg][col = i].
//
// Numerics: log-softmax is computed stably (max-subtracted); the reference
// takes log(softmax(z)) unguarded and can emit NaN when a probability
// underflows.  For one-hot labels the gradient is (p - y)/B; soft labels use
// the general form (p * sum(y) - y)/B.
#include "dtf_api.h"
#include "dtf_internal.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kCP = 16;   // class dimension padded to one MFMA tile
constexpr int kLdc = 17;  // LDS leading dimension of [*, 16] operands read along k

struct HeadParams {
  float* zin;            // [M,H] pre-activation sums (pre_act=0) or activations (pre_act=1)
  int pre_act;
  int zero_zin;          // write zeros back after reading (split-K accumulator recycle)
  const float* b1;
  const float* W2;
  const float* b2;
  const float* ybase;    // labels of batch 0
  const int* cursor;
  int bias;
  int batches_per_epoch;
  float* dz2;            // [M,H] out (GRAD/FUSED)
  float* oW2;            // gradient slab (GRAD) or the live W2 (FUSED)
  float* ob2;
  float* ob1;
  float grad_scale;      // GRAD: multiply gradients (1 or 1/replicas)
  const float* lr;
  float* loss_sum;       // EVAL: global accumulators
  int* correct;
  float* loss_hist;
  float* acc_hist;
  int hist_cap;
  float* global_step;
  int M, H, C, R, act, mode;
};

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
// argmax with ties resolved to the smallest index (tf.argmax behaviour).
__device__ __forceinline__ int group16_argmax(float v, int idx) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    const float ov = __shfl_xor(v, m);
    const int oi = __shfl_xor(idx, m);
    if (ov > v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
  return idx;
}
// Sum over the four 16-lane groups of a wave (rows of one MFMA tile).
__device__ __forceinline__ float across_groups_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

__global__ __launch_bounds__(kThreads) void mlp_head_kernel(HeadParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H16 = (p.H + 15) & ~15;
  const int lda = H16 + 1;  // odd: column-k reads of 16 rows hit distinct banks
  float* a2s = lds;                       // [R][lda]
  float* W2s = a2s + p.R * lda;           // [H16][kLdc]
  float* dz3s = W2s + H16 * kLdc;         // [R][kLdc]
  float* db1s = dz3s + p.R * kLdc;        // [H16]
  float* db2s = db1s + H16;               // [16]
  float* scal = db2s + 16;                // [0] loss sum, [1] correct count

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i16 = lane & 15;
  const int q = lane >> 4;
  const int r0 = blockIdx.x * p.R;
  const int nrows = min(p.R, p.M - r0);
  const int H = p.H, C = p.C, R = p.R;
  const bool single = gridDim.x == 1;

  // ---- phase 0: stage parameters and activations -------------------------
  for (int e = tid; e < H16 * kCP; e += kThreads) {
    const int h = e >> 4, c = e & 15;
    W2s[h * kLdc + c] = (h < H && c < C) ? p.W2[h * C + c] : 0.f;
  }
  for (int e = tid; e < H16; e += kThreads) db1s[e] = 0.f;
  if (tid < 16) db2s[tid] = 0.f;
  if (tid < 2) scal[tid] = 0.f;
  {
    // Four independent loads per thread are issued before any of them is
    // consumed, so the staging costs one global round trip, not twelve.
    constexpr int U = 4;
    const int total = R * H16;
    for (int base = tid; base < total; base += kThreads * U) {
      float z[U], bv[U];
      int rowv[U], hv[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = base + u * kThreads;
        rowv[u] = e / H16;
        hv[u] = e - rowv[u] * H16;
        ok[u] = e < total && rowv[u] < nrows && hv[u] < H;
        z[u] = ok[u] ? p.zin[(long)(r0 + rowv[u]) * H + hv[u]] : 0.f;
        bv[u] = (ok[u] && !p.pre_act) ? p.b1[hv[u]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = base + u * kThreads;
        if (e < total) {
          float a = 0.f;
          if (ok[u]) {
            a = p.pre_act ? z[u] : dtf_act_fwd(z[u] + bv[u], p.act);
            if (p.zero_zin) p.zin[(long)(r0 + rowv[u]) * H + hv[u]] = 0.f;
          }
          a2s[rowv[u] * lda + hv[u]] = a;
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 1: logits, softmax, loss, accuracy, dz3 ----------------------
  const float* __restrict__ Y =
      p.ybase + dtf_batch_index(p.cursor, p.bias, p.batches_per_epoch) * (long)p.M * C;
  const float inv_batch = 1.0f / (float)p.M;
  const int row_tiles = R >> 4;
  const int h_tiles = H16 >> 4;
  const int H4 = (H + 3) & ~3;
  for (int t = wave; t < row_tiles; t += kWaves) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* arow = a2s + (t * 16 + i16) * lda + q;
    const float* brow = W2s + q * kLdc + i16;
    for (int k = 0; k < H4; k += 4)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k], brow[k * kLdc], acc, 0, 0, 0);
    const bool col_ok = i16 < C;
    const float bias2 = col_ok ? p.b2[i16] : 0.f;
    float colsum = 0.f, wloss = 0.f, wcorrect = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = t * 16 + q * 4 + g;
      const bool row_ok = row < nrows;
      const float z = acc[g] + bias2;
      const float zm = col_ok ? z : -INFINITY;
      const float mx = group16_max(zm);
      const float e = col_ok ? expf(z - mx) : 0.f;
      const float se = group16_sum(e);
      const float logp = z - mx - logf(se);
      const float prob = e / se;
      const float y = (row_ok && col_ok) ? Y[(long)(r0 + row) * C + i16] : 0.f;
      const float ysum = group16_sum(y);
      const float rloss = group16_sum(col_ok ? -y * logp : 0.f);
      const int az = group16_argmax(zm, i16);
      const int ay = group16_argmax(col_ok ? y : -INFINITY, i16);
      if (row_ok && i16 == 0) {
        wloss += rloss;
        wcorrect += (az == ay) ? 1.f : 0.f;
      }
      const float dz = (row_ok && col_ok) ? (prob * ysum - y) * inv_batch : 0.f;
      dz3s[row * kLdc + i16] = dz;
      colsum += dz;
    }
    // lanes with i16 == 0 of the four groups carry the row sums
    wloss = across_groups_sum(wloss);
    wcorrect = across_groups_sum(wcorrect);
    if (lane == 0) {
      atomicAdd(&scal[0], wloss);
      atomicAdd(&scal[1], wcorrect);
    }
    if (p.mode != DTF_HEAD_EVAL) {
      colsum = across_groups_sum(colsum);
      if (q == 0 && col_ok) atomicAdd(&db2s[i16], colsum);
    }
  }
  __syncthreads();

  if (p.mode == DTF_HEAD_EVAL) {
    if (tid == 0) {
      atomicAdd(p.loss_sum, scal[0]);
      atomicAdd(p.correct, (int)(scal[1] + 0.5f));
    }
    return;
  }

  const float lr = (p.mode == DTF_HEAD_FUSED) ? *p.lr : 0.f;
  const float gs = p.grad_scale;

  // ---- phase 2: dW2 tiles, then da2/dz2 tiles ------------------------------
  const int n_dw = h_tiles;
  const int n_da = row_tiles * h_tiles;
  const int C4 = (C + 3) & ~3;
  for (int item = wave; item < n_dw + n_da; item += kWaves) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (item < n_dw) {
      // dW2[h][c] = sum_b a2[b][h] dz3[b][c]
      const int th = item;
      const float* arow = a2s + q * lda + th * 16 + i16;
      const float* brow = dz3s + q * kLdc + i16;
      for (int kb = 0; kb < R; kb += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[kb * lda], brow[kb * kLdc], acc, 0, 0, 0);
      if (i16 < C) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int h = th * 16 + q * 4 + g;
          if (h < H) {
            float* dst = p.oW2 + h * C + i16;
            if (p.mode == DTF_HEAD_FUSED) {
              *dst = *dst - lr * acc[g];
            } else if (single) {
              *dst = gs * acc[g];
            } else {
              atomicAdd(dst, gs * acc[g]);
            }
          }
        }
      }
    } else {
      // da2[b][h] = sum_c dz3[b][c] W2[h][c];  dz2 = da2 * act'(a2)
      const int it = item - n_dw;
      const int tb = it / h_tiles, th = it - tb * h_tiles;
      const float* arow = dz3s + (tb * 16 + i16) * kLdc + q;
      const float* brow = W2s + (th * 16 + i16) * kLdc + q;
      for (int kc = 0; kc < C4; kc += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[kc], brow[kc], acc, 0, 0, 0);
      const int h = th * 16 + i16;
      float csum = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = tb * 16 + q * 4 + g;
        const float a = a2s[row * lda + h];
        const float dz = acc[g] * dtf_act_bwd(a, p.act);
        if (row < nrows && h < H) {
          p.dz2[(long)(r0 + row) * H + h] = dz;
          csum += dz;
        }
      }
      csum = across_groups_sum(csum);
      if (q == 0 && h < H) atomicAdd(&db1s[h], csum);
    }
  }
  __syncthreads();

  // ---- phase 3: bias gradients / updates, bookkeeping ----------------------
  for (int h = tid; h < H; h += kThreads) {
    const float g = db1s[h];
    if (p.mode == DTF_HEAD_FUSED) {
      p.ob1[h] = p.ob1[h] - lr * g;
    } else if (single) {
      p.ob1[h] = gs * g;
    } else {
      atomicAdd(&p.ob1[h], gs * g);
    }
  }
  if (tid < C) {
    const float g = db2s[tid];
    if (p.mode == DTF_HEAD_FUSED) {
      p.ob2[tid] = p.ob2[tid] - lr * g;
    } else if (single) {
      p.ob2[tid] = gs * g;
    } else {
      atomicAdd(&p.ob2[tid], gs * g);
    }
  }
  if (tid == 0) {
    if (p.loss_hist && p.hist_cap > 0) {
      const long step = (p.cursor ? (long)(*p.cursor) : 0L) + p.bias;
      const int slot = (int)(step % p.hist_cap);
      const float l = scal[0] * inv_batch;   // block's share of the batch-mean loss
      const float a = scal[1] * inv_batch;
      if (single) {
        p.loss_hist[slot] = l;
        if (p.acc_hist) p.acc_hist[slot] = a;
      } else {
        atomicAdd(&p.loss_hist[slot], l);  // slot zeroed by the launcher
        if (p.acc_hist) atomicAdd(&p.acc_hist[slot], a);
      }
    }
    if (p.mode == DTF_HEAD_FUSED && p.global_step) *p.global_step = *p.global_step + 1.0f;
  }
}

size_t head_lds_bytes(int R, int H) {
  const int H16 = (H + 15) & ~15;
  const size_t floats = (size_t)R * (H16 + 1) + (size_t)H16 * kLdc + (size_t)R * kLdc + H16 + 16 + 2;
  return floats * sizeof(float);
}

// Rows per workgroup: the whole batch when it fits (<= 128 rows), otherwise
// the largest multiple of 16 whose LDS image stays under 144 KiB.
int pick_rows(int M, int H) {
  int R = ((M + 15) / 16) * 16;
  if (R > 128) R = 128;
  while (R > 16 && head_lds_bytes(R, H) > 144 * 1024) R -= 16;
  return R;
}

bool g_attr_set = false;

}  // namespace

int dtf_launch_head(const DtfHeadLaunch& a, hipStream_t stream) {
  if (!a.zin || !a.W2 || !a.b2 || !a.ybase) return DTF_ERR_NULL;
  if (!a.pre_act && !a.b1) return DTF_ERR_NULL;
  if (a.M <= 0 || a.H <= 0 || a.C <= 0 || a.C > kCP) return DTF_ERR_SHAPE;
  if (a.mode == DTF_HEAD_EVAL && (!a.loss_sum || !a.correct)) return DTF_ERR_NULL;
  if (a.mode != DTF_HEAD_EVAL && (!a.dz2 || !a.oW2 || !a.ob2 || !a.ob1)) return DTF_ERR_NULL;
  if (a.mode == DTF_HEAD_FUSED && !a.lr) return DTF_ERR_NULL;
  const int R = pick_rows(a.M, a.H);
  const size_t lds = head_lds_bytes(R, a.H);
  if (lds > 160 * 1024) return DTF_ERR_UNSUPPORTED;
  const int grid = (a.M + R - 1) / R;
  if (a.mode == DTF_HEAD_FUSED && grid != 1) return DTF_ERR_UNSUPPORTED;
  if (!g_attr_set) {
    if (hipFuncSetAttribute((const void*)mlp_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return DTF_ERR_LAUNCH;
    g_attr_set = true;
  }
  HeadParams p;
  p.zin = a.zin; p.pre_act = a.pre_act; p.zero_zin = a.zero_zin;
  p.b1 = a.b1; p.W2 = a.W2; p.b2 = a.b2; p.ybase = a.ybase;
  p.cursor = a.cursor; p.bias = a.bias; p.batches_per_epoch = a.bpe;
  p.dz2 = a.dz2; p.oW2 = a.oW2; p.ob2 = a.ob2; p.ob1 = a.ob1;
  p.grad_scale = a.grad_scale; p.lr = a.lr;
  p.loss_sum = a.loss_sum; p.correct = a.correct;
  p.loss_hist = a.loss_hist; p.acc_hist = a.acc_hist; p.hist_cap = a.hist_cap;
  p.global_step = a.global_step;
  p.M = a.M; p.H = a.H; p.C = a.C; p.R = R; p.act = a.act; p.mode = a.mode;
  hipLaunchKernelGGL(mlp_head_kernel, dim3(grid), dim3(kThreads), lds, stream, p);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

int dtf_head_grid(int M, int H) {
  const int R = pick_rows(M, H);
  return (M + R - 1) / R;
}
