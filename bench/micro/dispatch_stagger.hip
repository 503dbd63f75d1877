// How long does the chip take to START the workgroups of one launch?
//   hipcc --offload-arch=gfx950 -O3 bench/micro/dispatch_stagger.hip -o bench/micro/bin/dispatch_stagger
// Every workgroup stamps the wall clock (100 MHz) when its first wave starts and spins ~3 us (so that none has left
// before the last one arrives).  Reported: the spread of the start stamps for 247 workgroups (the training step's
// grid) with 512 / 256 / 128 threads and with / without the step's 38 KB of LDS, second launch of each variant.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                         \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                          \
    }                                                                    \
  } while (0)

template <int THREADS, int LDS_FLOATS>
__global__ __launch_bounds__(THREADS) void stamp_kernel(long long* starts, float* sink) {
  __shared__ float lds[LDS_FLOATS > 0 ? LDS_FLOATS : 1];
  const long long t0 = (long long)wall_clock64();
  if (threadIdx.x == 0) starts[blockIdx.x] = t0;
  if (LDS_FLOATS > 0) lds[threadIdx.x] = (float)t0;
  while ((long long)wall_clock64() - t0 < 300) __builtin_amdgcn_s_sleep(4);
  if (LDS_FLOATS > 0 && threadIdx.x == 0 && lds[1] == -1.f) sink[0] = lds[0];
}

template <int THREADS, int LDS_FLOATS>
int run(const char* name, int blocks, long long* d_starts, float* sink) {
  std::vector<long long> h(blocks);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((stamp_kernel<THREADS, LDS_FLOATS>), dim3(blocks), dim3(THREADS), 0, 0, d_starts, sink);
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipMemcpy(h.data(), d_starts, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("%-44s %4d workgroups: half started after %4lld ns, 90%% after %4lld ns, all after %4lld ns\n", name, blocks,
         (h[blocks / 2] - h[0]) * 10, (h[blocks * 9 / 10] - h[0]) * 10, (h[blocks - 1] - h[0]) * 10);
  return 0;
}

int main() {
  long long* d;
  float* sink;
  CHECK(hipMalloc(&d, sizeof(long long) * 1024));
  CHECK(hipMalloc(&sink, 64));
  if (run<512, 0>("512 threads, no LDS", 247, d, sink)) return 1;
  if (run<512, 9728>("512 threads, 38 KB LDS", 247, d, sink)) return 1;
  if (run<256, 0>("256 threads, no LDS", 247, d, sink)) return 1;
  if (run<256, 9728>("256 threads, 38 KB LDS", 247, d, sink)) return 1;
  if (run<128, 0>("128 threads, no LDS", 247, d, sink)) return 1;
  if (run<1024, 0>("1024 threads, no LDS", 247, d, sink)) return 1;
  if (run<512, 0>("512 threads, no LDS", 128, d, sink)) return 1;
  if (run<512, 0>("512 threads, no LDS", 494, d, sink)) return 1;
  return 0;
}
