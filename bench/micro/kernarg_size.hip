// Does the size of a kernel's argument block show in the time of a chain of dependent launches?
//   hipcc --offload-arch=gfx950 -O3 bench/micro/kernarg_size.hip -o bench/micro/bin/kernarg_size
// The training step passes 324 bytes of arguments (81 dwords, more than the scalar registers hold at once).  Each
// variant launches 352 workgroups of 512 threads, 500 launches per hipGraph replay; every thread reads the LAST
// field of its argument block, so the whole block has to arrive before the wave can finish.
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x)                                                         \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                          \
    }                                                                    \
  } while (0)

template <int N>
struct Args {
  unsigned* out;
  int pad[N];
  int last;
};

template <int N>
__global__ __launch_bounds__(512) void touch_kernel(Args<N> a) {
  int s = a.last;
  if (N >= 16) {            // use every dword, as the step does: all of them have to sit in scalar registers at once
#pragma unroll
    for (int i = 0; i < N; ++i) s += a.pad[i] * (i + 1);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) a.out[0] = (unsigned)s;
}

template <int N>
int run(const char* name, unsigned* out, hipStream_t st) {
  const int steps = 500, reps = 20;
  Args<N> a{};
  a.out = out;
  a.last = 1;
  hipGraph_t graph;
  hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(touch_kernel<N>, dim3(247), dim3(512), 0, st, a);
  CHECK(hipStreamEndCapture(st, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CHECK(hipGraphLaunch(exec, st));
  CHECK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(exec, st));
  CHECK(hipEventRecord(e1, st));
  CHECK(hipStreamSynchronize(st));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-40s %.3f us per launch\n", name, ms * 1000 / (reps * steps));
  return 0;
}

int main() {
  unsigned* out;
  CHECK(hipMalloc(&out, 64));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  if (run<1>("16 bytes of arguments", out, st)) return 1;
  if (run<29>("128 bytes of arguments", out, st)) return 1;
  if (run<78>("324 bytes of arguments", out, st)) return 1;
  if (run<1>("16 bytes of arguments (again)", out, st)) return 1;
  return 0;
}
