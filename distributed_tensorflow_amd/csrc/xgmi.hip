// One-shot gradient all-reduce fused with the SGD apply, over peer-mapped HBM.
//
// N replicas (one process per GPU, or several processes on one GPU) each hold
//   * two gradient slabs (even / odd steps) that every peer maps through HIP-IPC,
//   * a small array of arrival flags in uncached memory that the peers write into.
// A step's exchange is ONE kernel per replica:
//
//   announce   block 0 publishes "my gradient of exchange s is complete" by storing s
//              into its slot of every peer's flag array (release, system scope: the
//              gradient written by the preceding backward kernel is flushed first);
//   wait       every block polls the LOCAL flag array until all peers have announced s
//              (bounded by a wall-clock limit: on expiry the kernel records an error,
//              skips the update and still terminates);
//   reduce     acquire at system scope, then each thread reads the same 16 bytes of all
//              N slabs -- its own from local HBM, the others over xGMI, all loads in
//              flight together -- and adds them in rank order, so every replica computes
//              bit-identical sums;
//   apply      p -= lr * sum on the local replica of the variables, global_step += 1.
//
// The mean of the reference's SyncReplicasOptimizer (tfdist_between_sync.py:65-68) comes
// from gradients that the backward kernel already scaled by 1/N.  With 318 KB per
// replica the exchange is latency-bound: one flag round trip plus one remote read, instead
// of the several ring steps of a library all-reduce.  All peers reach every link of the
// fully connected xGMI mesh at once (7 x 318 KB inbound per GPU per step).
//
// Why two slabs are enough: a replica overwrites slab (s+1)&1 only after it passed the wait
// of exchange s, i.e. after every peer has STARTED exchange s, and a peer starts exchange s
// only after its own exchange s-1 -- the last reader of slab (s-1)&1 -- has finished.
#include <cstring>

#include "dtf_api.h"
#include "dtf_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxRanks = DTF_XGMI_MAX_RANKS;

// state words
constexpr int kSeq = 0, kDone = 1, kError = 2;

// All pointers travel in the kernel arguments: with a compile-time replica count the compiler keeps
// them in scalar registers and the kernel starts without a dependent table load.
struct XgmiArgs {
  const float* grad[kMaxRanks];      // [rank] gradient slab of this exchange's slot (own entry: local)
  uint32_t* peer_flags[kMaxRanks];   // [rank] base of that rank's arrival-flag array
  const uint32_t* my_flags;
  float* params;
  int64_t n;
  const float* lr;
  float* global_step;
  uint32_t* state;
  int rank, world;
  int64_t timeout_ticks;
  int required;                      // subset form only: how many of the `world` gradients count (the first to arrive)
};

template <int W>
__device__ __forceinline__ const float* pick(const float* const (&tab)[kMaxRanks], int r, int world) {
  const float* p = tab[0];
  if (W > 0) {
#pragma unroll
    for (int k = 1; k < W; ++k) p = (r == k) ? tab[k] : p;
  } else {
    for (int k = 1; k < world; ++k) p = (r == k) ? tab[k] : p;
  }
  return p;
}

template <int W>
__global__ __launch_bounds__(kBlock) void xgmi_allreduce_apply_kernel(XgmiArgs a) {
  __shared__ int failed;
  const int tid = threadIdx.x;
  const int world = W > 0 ? W : a.world;
  const uint32_t seq = __hip_atomic_load(&a.state[kSeq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t poisoned = __hip_atomic_load(&a.state[kError], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) failed = poisoned ? 1 : 0;

  // Local operands of this thread's first vector are requested before the wait: the parameters and the
  // replica's own gradient (complete since the preceding kernel) do not depend on any peer.
  const int64_t nvec = a.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t first = (int64_t)blockIdx.x * kBlock + tid;
  f32x4* p4 = reinterpret_cast<f32x4*>(a.params);
  const f32x4* own4 = reinterpret_cast<const f32x4*>(pick<W>(a.grad, a.rank, world));
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 p_first = zero, own_first = zero;
  if (first < nvec) {
    p_first = p4[first];
    own_first = own4[first];
  }
  const float lr = *a.lr;
  __syncthreads();

  if (!poisoned) {
    if (world > 1) {
      if (blockIdx.x == 0 && tid < world && tid != a.rank) {
        uint32_t* theirs = a.peer_flags[0];
        if (W > 0) {
#pragma unroll
          for (int k = 1; k < W; ++k) theirs = (tid == k) ? a.peer_flags[k] : theirs;
        } else {
          for (int k = 1; k < world; ++k) theirs = (tid == k) ? a.peer_flags[k] : theirs;
        }
        __hip_atomic_store(&theirs[a.rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (tid < world && tid != a.rank) {
        const long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(&a.my_flags[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
          if (wall_clock64() - t0 > a.timeout_ticks) {
            failed = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();
      if (!failed) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: peers' slabs are read fresh
    }
    if (!failed) {
      for (int64_t i = first; i < nvec; i += stride) {
        const bool head = i == first;
        const f32x4 p = head ? p_first : p4[i];
        f32x4 sum;
        if (W > 0) {
          f32x4 v[W > 0 ? W : 1];
#pragma unroll
          for (int r = 0; r < W; ++r) {
            if (r == a.rank) v[r] = head ? own_first : own4[i];
            else v[r] = reinterpret_cast<const f32x4*>(a.grad[r])[i];
          }
          sum = v[0];
#pragma unroll
          for (int r = 1; r < W; ++r) sum = sum + v[r];     // rank order: identical bits on every replica
        } else {
          sum = reinterpret_cast<const f32x4*>(pick<0>(a.grad, 0, world))[i];
          for (int r = 1; r < world; ++r) sum = sum + reinterpret_cast<const f32x4*>(pick<0>(a.grad, r, world))[i];
        }
        p4[i] = p - lr * sum;
      }
      if (blockIdx.x == 0) {
        const int64_t t = (nvec << 2) + tid;
        if (t < a.n) {
          float sum = pick<W>(a.grad, 0, world)[t];
          for (int r = 1; r < world; ++r) sum += pick<W>(a.grad, r, world)[t];
          a.params[t] -= lr * sum;
        }
        // thread 0 of block 0 also owns vector 0, so a global_step that lives inside the slab is
        // updated after its (zero-gradient) SGD write, in program order
        if (tid == 0 && a.global_step) *a.global_step += 1.0f;
      }
    }
  }

  __syncthreads();
  if (tid == 0) {
    if (failed && !poisoned) atomicExch(&a.state[kError], 1u);
    // Plain counting: the end of the kernel publishes everything to the next launch on this stream.
    const uint32_t done = __hip_atomic_fetch_add(&a.state[kDone], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {      // last block out: the next launch sees seq + 1
      __hip_atomic_store(&a.state[kDone], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.state[kSeq], seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- the same exchange with BACKUP replicas: the first `required` of the `world` gradients count --------------------
// SyncReplicasOptimizer(replicas_to_aggregate = R < total_num_replicas = N) (tfdist_between_sync.py:66-67 passes equal
// values; the optimizer it calls takes fewer): an update is the mean of the first R gradients that arrive for a step, the
// other N - R are dropped, every replica goes on with the same new values.
//
//   ticket     thread 0 of block 0 draws the replica's arrival number for exchange s from ONE 8-byte word
//              {s, arrivals so far} in replica 0's flag memory (compare-and-swap at system scope; the word of the
//              exchange two before, same parity, is recycled by whoever arrives first);
//   announce   {s, ticket} goes into slot (s & 1, rank) of EVERY replica's flag array, the replica's own included
//              (release: the gradient slab is complete);
//   wait       every block polls its local array until all N slots carry s -- all of them: the replicas keep one copy
//              of the variables each and stay in step, what the backup semantics change is whose gradients are summed --
//              and notes which replicas drew a ticket below R;
//   reduce     the members' slabs are added in rank order (the others' are not even read) and applied.
// The backward kernels scale their gradients by 1 / R.  Slots alternate with the parity of s: a replica that is one
// exchange ahead announces into the other slot, and it cannot be two ahead (it needs this replica's announcement of
// s + 1 first).  state[3] keeps the member mask of the last exchange.
constexpr int kAnnounce = 32;        // first word of the 2 x 16 announcement slots in a flag array
constexpr int kTicketWord = 64;      // index, in 8-byte words, of the two ticket words in replica 0's flag memory

__global__ __launch_bounds__(kBlock) void xgmi_allreduce_apply_subset_kernel(XgmiArgs a) {
  __shared__ int failed;
  __shared__ uint32_t members, announce;
  const int tid = threadIdx.x;
  const int world = a.world;
  const uint32_t seq = __hip_atomic_load(&a.state[kSeq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t poisoned = __hip_atomic_load(&a.state[kError], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    failed = poisoned ? 1 : 0;
    members = 0u;
  }
  const int64_t nvec = a.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t first = (int64_t)blockIdx.x * kBlock + tid;
  f32x4* p4 = reinterpret_cast<f32x4*>(a.params);
  const float lr = *a.lr;
  const int slot0 = kAnnounce + 16 * (int)(seq & 1u);
  __syncthreads();

  if (!poisoned) {
    if (blockIdx.x == 0) {
      if (tid == 0) {
        uint64_t* word = reinterpret_cast<uint64_t*>(a.peer_flags[0]) + kTicketWord + (seq & 1u);
        uint64_t seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        uint32_t ticket;
        for (;;) {
          const bool current = (uint32_t)(seen >> 32) == seq;
          ticket = current ? (uint32_t)seen : 0u;
          const uint64_t want = ((uint64_t)seq << 32) | (uint64_t)(ticket + 1u);
          if (__hip_atomic_compare_exchange_strong(word, &seen, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_SYSTEM))
            break;
        }
        announce = (seq << 4) | (ticket < 15u ? ticket : 15u);
      }
      __syncthreads();                                  // (block 0 only: the branch is uniform within a block)
      if (tid < world) {
        uint32_t* theirs = const_cast<uint32_t*>(a.my_flags);
        for (int k = 0; k < world; ++k) theirs = (tid == k && k != a.rank) ? a.peer_flags[k] : theirs;
        __hip_atomic_store(&theirs[slot0 + a.rank], announce, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (tid < world) {
      const long long t0 = wall_clock64();
      for (;;) {
        const uint32_t w = __hip_atomic_load(&a.my_flags[slot0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((w >> 4) == seq) {
          if ((int)(w & 15u) < a.required) atomicOr(&members, 1u << tid);
          break;
        }
        if (wall_clock64() - t0 > a.timeout_ticks) {
          failed = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    if (!failed) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");     // system scope: the members' slabs are read fresh
      const uint32_t mask = members;
      for (int64_t i = first; i < nvec; i += stride) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r)                 // rank order: identical bits on every replica
          if ((mask >> r) & 1u) sum = sum + reinterpret_cast<const f32x4*>(pick<0>(a.grad, r, world))[i];
        p4[i] = p4[i] - lr * sum;
      }
      if (blockIdx.x == 0) {
        const int64_t t = (nvec << 2) + tid;
        if (t < a.n) {
          float sum = 0.f;
          for (int r = 0; r < world; ++r)
            if ((mask >> r) & 1u) sum += pick<0>(a.grad, r, world)[t];
          a.params[t] -= lr * sum;
        }
        if (tid == 0) {
          if (a.global_step) *a.global_step += 1.0f;   // (after thread 0's own write of vector 0, in program order)
          a.state[3] = mask;
        }
      }
    }
  }

  __syncthreads();
  if (tid == 0) {
    if (failed && !poisoned) atomicExch(&a.state[kError], 1u);
    const uint32_t done = __hip_atomic_fetch_add(&a.state[kDone], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {      // last block out: the next launch sees seq + 1
      __hip_atomic_store(&a.state[kDone], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.state[kSeq], seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- asynchronous parameter-server traffic over the same mappings ---------------------------------
// The variables are one flat slab cut into `world` equal shards; shard k's authoritative copy lives in
// rank k's HBM (inside a buffer as large as the whole slab, so an element has the same index everywhere).
//   pull : local[i] = shard_owner(i)[i]            one launch reads every shard, remote ones over xGMI
//   push : shard_owner(i)[i] -= lr * grad[i]       one launch updates every shard IN PLACE, one-sided and
//          lock-free: concurrent workers interleave (and may lose) updates, the meaning of
//          use_locking=False in the reference (tfdist_between.py:65); global_step += 1 atomically.
// No worker waits for any other: there is no flag, no barrier and no matching receive.
struct ShardArgs {
  float* shard[kMaxRanks];
  int64_t chunk;        // floats per shard, a multiple of 4
  int64_t n;            // floats in the slab, world * chunk
  int world;
};

__device__ __forceinline__ float* shard_of(const ShardArgs& a, int owner) {
  float* p = a.shard[0];
  for (int k = 1; k < a.world; ++k) p = (owner == k) ? a.shard[k] : p;
  return p;
}

// (32-bit vector indices: the host checks that the slab has fewer than 2^31 vectors)
__global__ __launch_bounds__(kBlock) void shard_push_sgd_kernel(ShardArgs a, const float* __restrict__ grad,
                                                                const float* __restrict__ lr_ptr, int64_t step_index) {
  const float lr = *lr_ptr;
  const unsigned nvec = (unsigned)(a.n >> 2), cvec = (unsigned)(a.chunk >> 2);
  const unsigned stride = gridDim.x * kBlock;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(grad);
  const unsigned step_vec = step_index >= 0 ? (unsigned)(step_index >> 2) : 0xffffffffu;
  for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    float* dst = shard_of(a, (int)(i / cvec));
    const f32x4 g = g4[i];
    if (i != step_vec) {
      f32x4* d4 = reinterpret_cast<f32x4*>(dst);
      d4[i] = d4[i] - lr * g;
    } else {
      // the vector that holds global_step: the counter is only ever touched by the atomic below
      for (int c = 0; c < 4; ++c) {
        const int64_t e = ((int64_t)i << 2) + c;
        if (e != step_index) dst[e] -= lr * g[c];
      }
      __hip_atomic_fetch_add(dst + step_index, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Shards of different sizes (whole variables dealt round-robin over parameter-server tasks): shard k holds the
// slab vectors [startv[k], startv[k+1]) and base[k] is the address slab vector 0 WOULD have in its buffer (the
// host subtracts the shard's offset), so the owner of a vector is found with comparisons and no division.
struct RangeArgs {
  const f32x4* base[kMaxRanks];
  unsigned startv[kMaxRanks];
  unsigned nvec;
  int world;
};

__global__ __launch_bounds__(kBlock) void range_pull_kernel(RangeArgs a, f32x4* __restrict__ local) {
  const unsigned stride = gridDim.x * kBlock;
  for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < a.nvec; i += stride) {
    const f32x4* src = a.base[0];
    for (int k = 1; k < a.world; ++k) src = (i >= a.startv[k]) ? a.base[k] : src;
    local[i] = src[i];
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int dtf_xgmi_max_ranks(void) { return kMaxRanks; }

// ---- can a kernel on stream b run while a kernel on stream a waits for it? ---------------------------------------
// The streams of one process are multiplexed onto a few hardware queues (4 by default), a queue runs its kernels in
// order, and which queue a stream gets depends on what the process did before.  Two streams whose kernels wait for
// EACH OTHER (towers sharing a GPU) must therefore be checked, not assumed, to sit on different queues.
namespace {
__global__ void wait_flag_kernel(int* flag, int* out, long long timeout_ticks) {
  const long long t0 = (long long)wall_clock64();
  int seen = 0;
  while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) &&
         (long long)wall_clock64() - t0 < timeout_ticks)
    __builtin_amdgcn_s_sleep(8);
  *out = seen;
}
__global__ void set_flag_kernel(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}  // namespace

// 1: yes, 0: the waiting kernel timed out (the streams share a queue), < 0: a HIP call failed.
extern "C" int dtf_streams_independent(void* stream_a, void* stream_b, int timeout_us) {
  hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
  int* buf = nullptr;
  if (hipMalloc(&buf, 2 * sizeof(int)) != hipSuccess) return -1;
  int result = -1, seen = 0;
  if (hipMemsetAsync(buf, 0, 2 * sizeof(int), a) == hipSuccess && hipStreamSynchronize(a) == hipSuccess) {
    hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(1), 0, a, buf, buf + 1, (long long)timeout_us * 100);   // 100 MHz
    hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, b, buf);
    if (hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess &&
        hipMemcpy(&seen, buf + 1, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
      result = seen ? 1 : 0;
  }
  (void)hipFree(buf);
  return result;
}

// Let kernels launched on device `dev` dereference memory that lives on device `peer` (a shard mapped from a
// ps task on another GPU).  Already-enabled access is success; the current device is restored.
extern "C" int dtf_enable_peer_access(int dev, int peer) {
  if (dev == peer) return DTF_OK;
  int count = 0, current = 0, can = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || dev < 0 || peer < 0 || dev >= count || peer >= count)
    return DTF_ERR_SHAPE;
  if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) {
    (void)hipGetLastError();
    return DTF_ERR_UNSUPPORTED;
  }
  (void)hipGetDevice(&current);
  if (hipSetDevice(dev) != hipSuccess) return DTF_ERR_LAUNCH;
  const hipError_t rc = hipDeviceEnablePeerAccess(peer, 0);
  (void)hipGetLastError();
  (void)hipSetDevice(current);
  return (rc == hipSuccess || rc == hipErrorPeerAccessAlreadyEnabled) ? DTF_OK : DTF_ERR_LAUNCH;
}

// Device memory that peers may map.  uncached != 0: fine-grained memory that no L2 keeps a copy
// of, for words one GPU polls while another writes them.  The memory is zero-filled.
extern "C" int dtf_xgmi_alloc(void** out, int64_t bytes, int uncached) {
  if (!out || bytes <= 0) return DTF_ERR_NULL;
  void* p = nullptr;
  hipError_t rc = uncached ? hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached)
                           : hipMalloc(&p, (size_t)bytes);
  if (rc != hipSuccess || !p) {
    (void)hipGetLastError();
    return DTF_ERR_LAUNCH;
  }
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return DTF_ERR_LAUNCH;
  }
  *out = p;
  return DTF_OK;
}

extern "C" int dtf_xgmi_fill(void* p, int byte, int64_t bytes) {
  if (!p || bytes <= 0) return DTF_ERR_NULL;
  if (hipMemset(p, byte, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    return DTF_ERR_LAUNCH;
  }
  return DTF_OK;
}

extern "C" int dtf_xgmi_free(void* p) {
  if (!p) return DTF_OK;
  return hipFree(p) == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

// handle: 64 bytes (hipIpcMemHandle_t), valid in other processes of this node.
extern "C" int dtf_xgmi_export(void* p, void* handle) {
  if (!p || !handle) return DTF_ERR_NULL;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipGetLastError();
    return DTF_ERR_LAUNCH;
  }
  std::memcpy(handle, &h, sizeof(h));
  return DTF_OK;
}

extern "C" int dtf_xgmi_open(const void* handle, void** out) {
  if (!handle || !out) return DTF_ERR_NULL;
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) {
    (void)hipGetLastError();
    return DTF_ERR_LAUNCH;
  }
  *out = p;
  return DTF_OK;
}

extern "C" int dtf_xgmi_close(void* p) {
  if (!p) return DTF_OK;
  return hipIpcCloseMemHandle(p) == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

// One exchange of replica `rank`: see the file header.  `grads[r]` is rank r's gradient slab of the slot
// the preceding backward wrote (own entry: the local slab, others: their mappings), `peer_flags[r]` the
// base of rank r's flag array, `state` three zero-initialised words in local device memory whose first
// one must be set to 1 before the first exchange.
extern "C" int dtf_xgmi_allreduce_apply(const uint64_t* grads, const uint64_t* peer_flags, const uint32_t* my_flags,
                                        float* params, int64_t n, const float* lr, float* global_step,
                                        uint32_t* state, int rank, int world, int timeout_ms, void* stream) {
  return dtf_xgmi_allreduce_apply_subset(grads, peer_flags, my_flags, params, n, lr, global_step, state, rank, world, 0,
                                         timeout_ms, stream);
}

// The same with backup replicas: `required` in 1 .. world - 1 sums the gradients of the first `required` replicas to
// arrive (see xgmi_allreduce_apply_subset_kernel; `state` then has four words, the fourth receives the member mask, and
// the flag arrays are at least 1 KB); 0 or `world`: every replica's gradient, the kernel above.
extern "C" int dtf_xgmi_allreduce_apply_subset(const uint64_t* grads, const uint64_t* peer_flags, const uint32_t* my_flags,
                                               float* params, int64_t n, const float* lr, float* global_step,
                                               uint32_t* state, int rank, int world, int required, int timeout_ms,
                                               void* stream) {
  if (!grads || !peer_flags || !my_flags || !params || !lr || !state) return DTF_ERR_NULL;
  if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world || n < 0) return DTF_ERR_SHAPE;
  if (required < 0 || required > world) return DTF_ERR_SHAPE;
  if (required == world) required = 0;
  if (!aligned16(params)) return DTF_ERR_SHAPE;
  XgmiArgs a;
  for (int r = 0; r < kMaxRanks; ++r) {
    a.grad[r] = r < world ? reinterpret_cast<const float*>(grads[r]) : nullptr;
    a.peer_flags[r] = r < world ? reinterpret_cast<uint32_t*>(peer_flags[r]) : nullptr;
    if (r < world && (!a.grad[r] || !a.peer_flags[r] || !aligned16(a.grad[r]))) return DTF_ERR_NULL;
  }
  a.my_flags = my_flags;
  a.params = params;
  a.n = n;
  a.lr = lr;
  a.global_step = global_step;
  a.state = state;
  a.rank = rank;
  a.world = world;
  a.timeout_ticks = (int64_t)(timeout_ms > 0 ? timeout_ms : 5000) * 100000;   // 100 MHz wall clock
  a.required = required;
  int64_t blocks = ((n >> 2) + kBlock - 1) / kBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;      // every block must be resident while it waits
  const dim3 grid((unsigned)blocks), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (required > 0) {
    hipLaunchKernelGGL(xgmi_allreduce_apply_subset_kernel, grid, block, 0, s, a);
    return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
  }
  switch (world) {
    case 1: hipLaunchKernelGGL(xgmi_allreduce_apply_kernel<1>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(xgmi_allreduce_apply_kernel<2>, grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL(xgmi_allreduce_apply_kernel<4>, grid, block, 0, s, a); break;
    case 8: hipLaunchKernelGGL(xgmi_allreduce_apply_kernel<8>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(xgmi_allreduce_apply_kernel<0>, grid, block, 0, s, a); break;
  }
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

static int fill_shards(ShardArgs& a, const uint64_t* shards, int world, int64_t chunk) {
  if (!shards) return DTF_ERR_NULL;
  if (world < 1 || world > kMaxRanks || chunk <= 0 || (chunk & 3) || chunk * world >= (int64_t(1) << 33))
    return DTF_ERR_SHAPE;
  for (int r = 0; r < kMaxRanks; ++r) {
    a.shard[r] = r < world ? reinterpret_cast<float*>(shards[r]) : nullptr;
    if (r < world && (!a.shard[r] || !aligned16(a.shard[r]))) return DTF_ERR_NULL;
  }
  a.chunk = chunk;
  a.n = chunk * world;
  a.world = world;
  return DTF_OK;
}

static dim3 shard_grid(int64_t n) {
  int64_t blocks = ((n >> 2) + kBlock - 1) / kBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  return dim3((unsigned)blocks);
}

static int launch_range_pull(const RangeArgs& a, float* local, hipStream_t stream);

// local[0 .. world*chunk) = the authoritative copy of every shard.  shards: HOST array of `world` device
// addresses, entry k = base of rank k's slab-sized buffer (own entry local, the others mapped).
extern "C" int dtf_xgmi_pull(const uint64_t* shards, int world, int64_t chunk, float* local, void* stream) {
  ShardArgs a;
  const int rc = fill_shards(a, shards, world, chunk);
  if (rc != DTF_OK) return rc;
  if (!local || !aligned16(local)) return DTF_ERR_NULL;
  RangeArgs r;
  for (int k = 0; k < kMaxRanks; ++k) {
    r.base[k] = k < world ? reinterpret_cast<const f32x4*>(a.shard[k]) : nullptr;   // slab-sized buffers: no offset
    r.startv[k] = k < world ? (unsigned)((chunk >> 2) * k) : 0xffffffffu;
  }
  r.nvec = (unsigned)(a.n >> 2);
  r.world = world;
  return launch_range_pull(r, local, (hipStream_t)stream);
}

static int launch_range_pull(const RangeArgs& a, float* local, hipStream_t stream) {
  // (blocks of 128 or 64 threads, i.e. more, smaller workgroups, measured no faster: 14.7 / 17.2 against 14.6 us/step)
  unsigned blocks = (a.nvec + kBlock - 1) / kBlock;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(range_pull_kernel, dim3(blocks), dim3(kBlock), 0, stream, a, reinterpret_cast<f32x4*>(local));
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

// local[0 .. starts[world]) = shard 0 ++ shard 1 ++ ...: shard k holds the slab elements [starts[k], starts[k+1])
// and shards[k] is the device address of ITS first element (HOST arrays; empty shards may have address 0).
// Ranges and addresses must be 16-byte aligned.  One launch whatever the number of shards.
extern "C" int dtf_xgmi_pull_ranges(const uint64_t* shards, const int64_t* starts, int world, float* local,
                                    void* stream) {
  if (!shards || !starts) return DTF_ERR_NULL;
  if (world < 1 || world > kMaxRanks || starts[0] != 0) return DTF_ERR_SHAPE;
  if (!local || !aligned16(local)) return DTF_ERR_NULL;
  const int64_t total = starts[world];
  if (total <= 0 || (total & 3) || total >= (int64_t(1) << 33)) return DTF_ERR_SHAPE;
  RangeArgs a;
  for (int r = 0; r < kMaxRanks; ++r) {
    a.base[r] = nullptr;
    a.startv[r] = 0xffffffffu;
  }
  for (int r = 0; r < world; ++r) {
    const int64_t lo = starts[r], hi = starts[r + 1];
    if (lo < 0 || hi < lo || hi > total || (lo & 3)) return DTF_ERR_SHAPE;
    const float* p = reinterpret_cast<const float*>(shards[r]);
    if (hi > lo && (!p || !aligned16(p))) return DTF_ERR_NULL;
    a.base[r] = reinterpret_cast<const f32x4*>(p - lo);      // only vectors lo/4 .. hi/4-1 are ever formed from it
    a.startv[r] = (unsigned)(lo >> 2);
  }
  a.nvec = (unsigned)(total >> 2);
  a.world = world;
  return launch_range_pull(a, local, (hipStream_t)stream);
}

// Every shard -= lr * (its range of grad), written through the mappings; step_index >= 0: the element
// that holds global_step, which is incremented atomically instead (its gradient is ignored).
extern "C" int dtf_xgmi_push_sgd(const uint64_t* shards, int world, int64_t chunk, const float* grad,
                                 const float* lr, int64_t step_index, void* stream) {
  ShardArgs a;
  const int rc = fill_shards(a, shards, world, chunk);
  if (rc != DTF_OK) return rc;
  if (!grad || !lr || !aligned16(grad)) return DTF_ERR_NULL;
  if (step_index >= a.n) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(shard_push_sgd_kernel, shard_grid(a.n), dim3(kBlock), 0, (hipStream_t)stream, a, grad, lr,
                     step_index);
  return hipGetLastError() == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}
