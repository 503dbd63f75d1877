// How many streams of ONE process can hold a spinning kernel at once before a kernel on another stream can no
// longer start (streams share a few hardware queues)?  Decides how many service lanes the two-sided asynchronous
// parameter-server transport may keep blocked in ncclRecv at a time.
//
//   hipcc --offload-arch=gfx950 -O2 bench/micro/stream_queues.hip -o bench/micro/bin/stream_queues
//
// For S = 1, 2, 3, 4, 6, 8, 12, 16: S streams each run a one-workgroup kernel that waits (at most 0.3 s) for a
// flag; one more stream then runs the kernel that raises it.  Prints how many waiters timed out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void wait_flag(const unsigned* flag, unsigned* timed_out, long long ticks) {
  if (threadIdx.x) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    if (wall_clock64() - t0 > ticks) {
      atomicAdd(timed_out, 1u);
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

__global__ void raise_flag(unsigned* flag) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  unsigned *flag, *bad;
  if (hipMalloc(&flag, 4) != hipSuccess || hipMalloc(&bad, 4) != hipSuccess) return 1;
  const int sizes[] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
  for (int S : sizes) {
    std::vector<hipStream_t> st(S + 1);
    for (auto& s : st)
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 2;
    hipMemset(flag, 0, 4);
    hipMemset(bad, 0, 4);
    hipDeviceSynchronize();
    for (int i = 0; i < S; ++i) hipLaunchKernelGGL(wait_flag, dim3(1), dim3(64), 0, st[i], flag, bad, 30000000LL);
    hipLaunchKernelGGL(raise_flag, dim3(1), dim3(64), 0, st[S], flag);
    hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("waiting streams %2d: %u timed out\n", S, h);
    fflush(stdout);
    for (auto& s : st) hipStreamDestroy(s);
  }
  return 0;
}
