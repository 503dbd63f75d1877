// Is a grid-wide dependency INSIDE one launch cheaper than a kernel boundary, for the training step's grid?
//
// A training step is a launch of 352 workgroups x 512 threads whose successor needs what ALL of them wrote (every
// forward tile needs the whole logit accumulator and its column of the updated W1).  Consecutive steps are therefore
// separated by a kernel boundary.  The alternative is a persistent launch that loops over steps with a grid barrier
// built from flags: arrival counters (sharded per XCD), a release of the step's stores before the arrival and an
// acquire after the wait (the per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by
// another CU's stores).  This program measures both on the real grid, with a body that stores 2 KB per workgroup
// and reads 2 KB written by another workgroup in the previous step (checked for staleness):
//
//   hipcc --offload-arch=gfx950 -O2 bench/micro/persistent_boundary.hip -o bench/micro/bin/persistent_boundary
//
//   A  N dependent launches in a hipGraph               -> us per step (= body + kernel boundary)
//   B  ONE persistent launch, N steps, barrier variants -> us per step (= body + in-launch barrier)
//        B1 one arrival counter, release + acquire fences
//        B2 arrival counters sharded by blockIdx % 8, release + acquire fences
//        B3 sharded counters, NO fences (what the barrier alone costs; its data hand-off would be wrong)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kBlocks = 352, kThreads = 512, kWords = 512;   // 2 KB per workgroup and step

#define CHECK(x)                                                           \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

// body of one step: read the slab the NEIGHBOUR wrote in the previous step, store this step's own slab
__device__ __forceinline__ void body(unsigned* slabs, int step, unsigned* stale) {
  const int b = blockIdx.x, t = threadIdx.x;
  const unsigned* theirs = slabs + (size_t)((b + 37) % kBlocks) * kWords;
  const unsigned got = theirs[t];
  if (step > 0 && got != (unsigned)(step - 1)) atomicAdd(stale, 1u);
  slabs[(size_t)b * kWords + t] = (unsigned)step;     // (each slab is double-buffered by the caller's pointer swap)
}

__global__ __launch_bounds__(kThreads) void step_kernel(unsigned* rd, unsigned* wr, int step, unsigned* stale) {
  const int b = blockIdx.x, t = threadIdx.x;
  const unsigned got = rd[(size_t)((b + 37) % kBlocks) * kWords + t];
  if (step > 0 && got != (unsigned)(step - 1)) atomicAdd(stale, 1u);
  wr[(size_t)b * kWords + t] = (unsigned)step;
}

// mode: bit 0 = sharded counters, bit 1 = fences
__global__ __launch_bounds__(kThreads) void persistent_kernel(unsigned* buf0, unsigned* buf1, unsigned* counters,
                                                              int steps, int mode, unsigned* stale, unsigned* gave_up) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int shards = (mode & 1) ? 8 : 1;
  __shared__ int bail;
  if (t == 0) bail = 0;
  __syncthreads();
  unsigned my_stale = 0;
  for (int s = 0; s < steps; ++s) {
    unsigned* rd = (s & 1) ? buf1 : buf0;
    unsigned* wr = (s & 1) ? buf0 : buf1;
    const unsigned got = rd[(size_t)((b + 37) % kBlocks) * kWords + t];
    my_stale += (s > 0 && got != (unsigned)(s - 1)) ? 1u : 0u;   // (counted in a register: no atomics in the loop)
    wr[(size_t)b * kWords + t] = (unsigned)s;
    // ---- grid barrier ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      if (mode & 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(&counters[(b % shards) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned per_shard_total = (unsigned)(s + 1);
      const long long t0 = wall_clock64();
      for (int k = 0; k < shards; ++k) {
        const unsigned members = (unsigned)((kBlocks - k + shards - 1) / shards);
        while (__hip_atomic_load(&counters[k * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < members * per_shard_total) {
          if (wall_clock64() - t0 > 200000000LL) {     // 2 s: never hang the GPU
            bail = 1;
            atomicAdd(gave_up, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (bail) break;
      }
      if (mode & 2) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (bail) break;
  }
  if (my_stale) atomicAdd(stale, my_stale);
}

int main() {
  unsigned *buf0, *buf1, *counters, *stale, *gave_up;
  const size_t slab_bytes = (size_t)kBlocks * kWords * 4;
  CHECK(hipMalloc(&buf0, slab_bytes));
  CHECK(hipMalloc(&buf1, slab_bytes));
  CHECK(hipMalloc(&counters, 8 * 32 * 4));
  CHECK(hipMalloc(&stale, 4));
  CHECK(hipMalloc(&gave_up, 4));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int steps = 500, reps = 10;

  // occupancy: every workgroup of the persistent launch must be resident
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent_kernel, kThreads, 0));
  printf("persistent kernel: %d workgroups per CU admitted, grid %d on 256 CUs\n", per_cu, kBlocks);
  if (per_cu * 256 < kBlocks) return 2;

  // A: dependent launches in a graph
  {
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipMemsetAsync(stale, 0, 4, st));
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < steps; ++s)
      hipLaunchKernelGGL(step_kernel, dim3(kBlocks), dim3(kThreads), 0, st, (s & 1) ? buf1 : buf0, (s & 1) ? buf0 : buf1, s, stale);
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(exec, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(exec, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h = 0;
    CHECK(hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost));
    printf("A  %d dependent launches per graph replay:            %.3f us per step (stale reads: %u)\n", steps, ms * 1000 / (reps * steps), h);
  }
  const char* names[4] = {"B1' one counter, no fences", "B3 sharded counters, no fences", "B1 one counter, release + acquire fences",
                          "B2 sharded counters, release + acquire fences"};
  for (int mode = 0; mode < 4; ++mode) {
    float total = 0;
    unsigned h = 0, g = 0;
    for (int r = 0; r < reps + 1; ++r) {
      CHECK(hipMemsetAsync(counters, 0, 8 * 32 * 4, st));
      CHECK(hipMemsetAsync(stale, 0, 4, st));
      CHECK(hipMemsetAsync(gave_up, 0, 4, st));
      CHECK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(persistent_kernel, dim3(kBlocks), dim3(kThreads), 0, st, buf0, buf1, counters, steps, mode, stale, gave_up);
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) total += ms;
      unsigned hh = 0, gg = 0;
      CHECK(hipMemcpy(&hh, stale, 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&gg, gave_up, 4, hipMemcpyDeviceToHost));
      h += hh;
      g += gg;
      if (gg) break;
    }
    printf("%-48s %.3f us per step (stale reads: %u of %d, barriers given up: %u)\n", names[mode], total * 1000 / (reps * steps), h,
           (reps + 1) * steps * kBlocks * kThreads, g);
    fflush(stdout);
  }
  return 0;
}
