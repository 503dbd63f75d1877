// Parameter-server side kernels over flat fp32 slabs:
//   sgd_apply   : p -= lr*scale*g        (ApplyGradientDescent of every variable of a
//                 shard in one launch, plus AssignAdd(global_step, 1);
//                 reference tfdist_between.py:65-66)
//   accumulate  : acc += g               (ConditionalAccumulator.apply_grad of
//                 SyncReplicasOptimizer, tfdist_between_sync.py:66-68)
//   philox_normal: tf.random_normal initialiser (tfdist_between.py:47-50)
//   gather_rows : per-epoch reshuffle of the GPU-resident training set
//                 (mnist.train.next_batch, tfdist_between.py:92)
// All are bandwidth-trivial (a shard is <= 318 KB); they are written as 16-B
// vector grid-stride loops so that the same code serves multi-GB slabs.
#include <chrono>

#include "dtf_api.h"
#include "dtf_common.h"

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n_vec) {
  int64_t g = (n_vec + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

// The step counter usually lives INSIDE the slab that an apply updates (it is a variable like the others, with a zero
// gradient).  The kernels below bump it with one atomic, so nothing else may write it: a plain read-modify-write of the
// vector that holds it -- by another thread of this launch, or by the launch of another worker that applies to the shard
// at the same time without a lock -- could put an old count back.  Index of the counter inside [p, p + n), or -4
// (no vector has index -1, no element index -4).
__device__ __forceinline__ int64_t step_index_in(const float* p, int64_t n, const float* global_step) {
  if (global_step == nullptr || global_step < p || global_step >= p + n) return -4;
  return global_step - p;
}

__global__ __launch_bounds__(kBlock) void sgd_apply_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           int64_t n, const float* __restrict__ lr_ptr,
                                                           float scale, int zero_grad,
                                                           float* global_step) {
  const float a = (*lr_ptr) * scale;
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int64_t gs_at = step_index_in(p, n, global_step);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    const f32x4 gv = g4[i];
    if (i == (gs_at >> 2)) {
      for (int e = 0; e < 4; ++e)
        if (4 * i + e != gs_at) p[4 * i + e] -= a * gv[e];
    } else {
      p4[i] = p4[i] - a * gv;
    }
    if (zero_grad) g4[i] = zero;
  }
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (t < n) {
      if (t != gs_at) p[t] -= a * g[t];
      if (zero_grad) g[t] = 0.f;
    }
    // atomic: several workers (processes) may apply to the same shard concurrently — the lock-free
    // asynchronous mode — and every applied gradient has to count
    if (threadIdx.x == 0 && global_step) atomicAdd(global_step, 1.0f);
  }
}

// The same update for a shard that several service streams update at once (use_locking=True): every element
// is one float atomic, so concurrent applies interleave per element and none is lost.  (The unlocked kernel
// above reads, subtracts and writes whole vectors: concurrent applies may overwrite each other, which is the
// meaning of use_locking=False in the reference, tfdist_between.py:65.)
__global__ __launch_bounds__(kBlock) void sgd_apply_atomic_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                  int64_t n, const float* __restrict__ lr_ptr,
                                                                  float scale, float* global_step) {
  const float a = -(*lr_ptr) * scale;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    (void)__hip_atomic_fetch_add(p + i, a * g[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x == 0 && threadIdx.x == 0 && global_step) atomicAdd(global_step, 1.0f);
}

// SyncReplicas aggregation in ONE launch: p -= lr * scale * (g_0 + g_1 + ... + g_{N-1}), the N gradient
// mailboxes summed in index order (take_grad of every accumulator + ApplyGradientDescent + the step bump,
// tfdist_between_sync.py:65-68), instead of N SGD launches.
constexpr int kMaxSources = 16;
struct SumApplyArgs {
  const float* g[kMaxSources];
  int count;
};

__global__ __launch_bounds__(kBlock) void sum_apply_kernel(float* __restrict__ p, SumApplyArgs s, int64_t n,
                                                           const float* __restrict__ lr_ptr, float scale,
                                                           float* global_step) {
  const float a = (*lr_ptr) * scale;
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  const int64_t gs_at = step_index_in(p, n, global_step);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    f32x4 sum = reinterpret_cast<const f32x4*>(s.g[0])[i];
#pragma unroll
    for (int k = 1; k < kMaxSources; ++k)
      if (k < s.count) sum = sum + reinterpret_cast<const f32x4*>(s.g[k])[i];
    if (i == (gs_at >> 2)) {
      for (int e = 0; e < 4; ++e)
        if (4 * i + e != gs_at) p[4 * i + e] -= a * sum[e];
    } else {
      p4[i] = p4[i] - a * sum;
    }
  }
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (t < n && t != gs_at) {
      float sum = s.g[0][t];
#pragma unroll
      for (int k = 1; k < kMaxSources; ++k)
        if (k < s.count) sum += s.g[k][t];
      p[t] -= a * sum;
    }
    if (threadIdx.x == 0 && global_step) atomicAdd(global_step, 1.0f);
  }
}

__global__ __launch_bounds__(kBlock) void accumulate_kernel(float* __restrict__ acc,
                                                            const float* __restrict__ g, int64_t n) {
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  f32x4* a4 = reinterpret_cast<f32x4*>(acc);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride)
    a4[i] = a4[i] + g4[i];
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 2) + threadIdx.x;
    if (t < n) acc[t] += g[t];
  }
}

// Philox4x32-10 (Salmon et al., SC'11): counter = element-group index, key = seed.
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                             uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
  const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t group) {
  uint32_t c0 = (uint32_t)group, c1 = (uint32_t)(group >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  // (0,1] uniforms -> Box-Muller pairs
  const float s = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)c0 + 1.0f) * s > 1.0f ? 1.0f : ((float)c0 + 1.0f) * s;
  const float u1 = (float)c1 * s;
  const float u2 = ((float)c2 + 1.0f) * s > 1.0f ? 1.0f : ((float)c2 + 1.0f) * s;
  const float u3 = (float)c3 * s;
  const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
  const float two_pi = 6.283185307179586f;
  f32x4 out;
  out[0] = r0 * cosf(two_pi * u1);
  out[1] = r0 * sinf(two_pi * u1);
  out[2] = r1 * cosf(two_pi * u3);
  out[3] = r1 * sinf(two_pi * u3);
  return out;
}

// Element i takes component (offset+i) & 3 of group (offset+i) >> 2, so a fill
// of a sub-range reproduces the values of the full-range fill.
__global__ __launch_bounds__(kBlock) void philox_normal_kernel(float* __restrict__ out, int64_t n,
                                                               uint64_t seed, uint64_t offset,
                                                               float mean, float stddev) {
  const uint64_t first_group = offset >> 2;
  const uint64_t last_group = (offset + (uint64_t)n + 3) >> 2;  // exclusive
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (uint64_t grp = first_group + (uint64_t)blockIdx.x * kBlock + threadIdx.x; grp < last_group;
       grp += stride) {
    const f32x4 v = philox_normal4(seed, grp);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint64_t gi = (grp << 2) + c;
      if (gi >= offset && gi < offset + (uint64_t)n) out[gi - offset] = mean + stddev * v[c];
    }
  }
}

// err (may be null): [0] number of indices found outside [0, src_rows), [1] the first such row of idx.  A bad index is
// clamped, so a corrupted permutation is REPORTED (hb.gather_errors) instead of faulting the device.
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ src,
                                                             const int64_t* __restrict__ idx,
                                                             float* __restrict__ dst, int64_t rows,
                                                             int cols, int64_t src_rows, int* __restrict__ err) {
  // one wave per row, 16-B vectors when the row length allows
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  const bool vec = (cols & 3) == 0;
  for (int64_t r = wave; r < rows; r += nwaves) {
    int64_t from = idx[r];
    if (from < 0 || from >= src_rows) {
      if (err && lane == 0 && atomicAdd(&err[0], 1) == 0) err[1] = (int)r;
      from = from < 0 ? 0 : src_rows - 1;
    }
    const float* s = src + from * (int64_t)cols;
    float* d = dst + r * (int64_t)cols;
    if (vec) {
      const f32x4* s4 = reinterpret_cast<const f32x4*>(s);
      f32x4* d4 = reinterpret_cast<f32x4*>(d);
      for (int c = lane; c < (cols >> 2); c += 64) d4[c] = s4[c];
    } else {
      for (int c = lane; c < cols; c += 64) d[c] = s[c];
    }
  }
}

__global__ void advance_cursor_kernel(int32_t* cursor, int n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *cursor = *cursor + n;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int dtf_sgd_apply(float* p, float* g, int64_t n, const float* lr, float scale, int zero_grad,
                             float* global_step, void* stream) {
  if (!p || !g || !lr) return DTF_ERR_NULL;
  if (n < 0) return DTF_ERR_SHAPE;
  if (!aligned16(p) || !aligned16(g)) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(sgd_apply_kernel, dim3(grid_for(n >> 2)), dim3(kBlock), 0, (hipStream_t)stream, p,
                     g, n, lr, scale, zero_grad, global_step);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_sgd_apply_atomic(float* p, const float* g, int64_t n, const float* lr, float scale,
                                    float* global_step, void* stream) {
  if (!p || !g || !lr) return DTF_ERR_NULL;
  if (n < 0) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(sgd_apply_atomic_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, p, g, n, lr,
                     scale, global_step);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

// grads: HOST array of `count` device addresses (1 <= count <= 16), every one n floats, 16-byte aligned.
extern "C" int dtf_sum_apply(float* p, const uint64_t* grads, int count, int64_t n, const float* lr, float scale,
                             float* global_step, void* stream) {
  if (!p || !grads || !lr) return DTF_ERR_NULL;
  if (n < 0 || count < 1 || count > kMaxSources || !aligned16(p)) return DTF_ERR_SHAPE;
  SumApplyArgs s;
  for (int k = 0; k < kMaxSources; ++k) {
    s.g[k] = reinterpret_cast<const float*>(grads[k < count ? k : 0]);
    if (!s.g[k] || !aligned16(s.g[k])) return DTF_ERR_NULL;
  }
  s.count = count;
  hipLaunchKernelGGL(sum_apply_kernel, dim3(grid_for(n >> 2)), dim3(kBlock), 0, (hipStream_t)stream, p, s, n, lr,
                     scale, global_step);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

// Rounds of the chief of synchronous replicas whose barrier is the shared-memory board (runtime/shm_board.cpp), on their
// common path, without a line of interpreter in between (the chief's thread shares an interpreter, and its lock, with the
// training loop of the chief's own replica): take the arrivals of all `workers`, and if they are what a round in step
// looks like -- one per worker, all of step `round_step`, all in the same mailbox row `slot` -- apply the mean of that
// row's gradients to every shard (dtf_sum_apply, one launch per shard, worker order), wait for the launches and deal
// `copies` tokens of step round_step + 1 to every worker (tfdist_between_sync.py:65-68, 86-89); then the next round,
// up to max_rounds of them.  take / release are dtf_board_take / dtf_board_release, handed over as addresses (they live
// in the runtime library).  shards: n_shards entries; tabs[slot * n_shards + k]: HOST array of `workers` device
// addresses (the mailbox rows).  *rounds: rounds applied.  Returns DTF_OK (max_rounds applied), 1 (arrivals that are
// not such a round: they are in out[0 .. 3 * *got), not applied, the caller goes on with them) or 2 (time-out of a
// round: the *got arrivals taken for it are in out).
extern "C" int dtf_sync_rounds(void* take_fn, void* release_fn, void* board, int workers, int64_t round_step, int timeout_ms,
                               int n_shards, float* const* params, const uint64_t* const* tabs, const int64_t* numel,
                               float* const* global_step, const float* lr, int copies, void* stream, int max_rounds,
                               int64_t* out, int* got, int* rounds) {
  typedef int (*TakeFn)(void*, int, int, int64_t*, int*);
  typedef int (*ReleaseFn)(void*, int64_t, int64_t, int);
  if (!take_fn || !release_fn || !board || !params || !tabs || !numel || !global_step || !lr || !out || !got || !rounds)
    return DTF_ERR_NULL;
  if (workers < 1 || workers > kMaxSources || n_shards < 1) return DTF_ERR_SHAPE;
  *rounds = 0;
  for (int r = 0; r < max_rounds; ++r, ++round_step) {
    *got = 0;
    int n = 0;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    while (n < workers) {       // (a take returns with the arrivals there are once there is one)
      const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
      int g = 0;
      const int rc = reinterpret_cast<TakeFn>(take_fn)(board, workers - n, left > 0 ? (int)left : 0, out + 3 * n, &g);
      n += g > 0 ? g : 0;
      *got = n;
      if (rc != 0 && n < workers) return 2;
    }
    unsigned seen = 0;
    const int64_t slot = out[2];
    for (int i = 0; i < n; ++i) {
      const int64_t w = out[3 * i], step = out[3 * i + 1];
      if (w < 0 || w >= workers || step != round_step || out[3 * i + 2] != slot || (seen >> w & 1u)) return 1;
      seen |= 1u << w;
    }
    if (slot < 0 || slot > 1) return 1;
    for (int k = 0; k < n_shards; ++k) {
      const int e = dtf_sum_apply(params[k], tabs[slot * n_shards + k], workers, numel[k], lr, 1.0f / (float)workers,
                                  global_step[k], stream);
      if (e != DTF_OK) return e;
    }
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return DTF_ERR_LAUNCH;   // tokens must not overtake the update
    *got = 0;
    if (reinterpret_cast<ReleaseFn>(release_fn)(board, -1, round_step + 1, copies) != 0) return DTF_ERR_LAUNCH;
    *rounds = r + 1;
  }
  return DTF_OK;
}

extern "C" int dtf_accumulate(float* acc, const float* g, int64_t n, void* stream) {
  if (!acc || !g) return DTF_ERR_NULL;
  if (n < 0 || !aligned16(acc) || !aligned16(g)) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(accumulate_kernel, dim3(grid_for(n >> 2)), dim3(kBlock), 0, (hipStream_t)stream,
                     acc, g, n);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, float mean,
                                 float stddev, void* stream) {
  if (!out) return DTF_ERR_NULL;
  if (n < 0) return DTF_ERR_SHAPE;
  if (n == 0) return DTF_OK;
  hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for((n + 3) >> 2)), dim3(kBlock), 0,
                     (hipStream_t)stream, out, n, seed, offset, mean, stddev);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int cols,
                               int64_t src_rows, int* err, void* stream) {
  if (!src || !idx || !dst) return DTF_ERR_NULL;
  if (rows < 0 || cols <= 0 || src_rows < 1) return DTF_ERR_SHAPE;
  if (rows == 0) return DTF_OK;
  if ((cols & 3) == 0 && (!aligned16(src) || !aligned16(dst))) return DTF_ERR_SHAPE;
  int64_t blocks = (rows * 64 + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((int)blocks), dim3(kBlock), 0, (hipStream_t)stream, src,
                     idx, dst, rows, cols, src_rows, err);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

extern "C" int dtf_advance_cursor(int32_t* cursor, int n, void* stream) {
  if (!cursor) return DTF_ERR_NULL;
  hipLaunchKernelGGL(advance_cursor_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cursor, n);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}
