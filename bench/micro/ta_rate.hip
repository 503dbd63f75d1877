// Micro-benchmark: how long does a 512-thread workgroup take to bring ~40 KB from L2 into
// registers with (a) 4-byte loads, (b) 16-byte loads?  One workgroup per CU, 108 workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 bench/micro/ta_rate.hip -o bench/micro/bin/ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
__global__ __launch_bounds__(512) void narrow(const float* __restrict__ src, float* __restrict__ out, long long* t) {
  const long long t0 = clock64();
  float v[N];
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = src[(blockIdx.x * 13 % 7) * 4096 + j * 512 + threadIdx.x];
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  const long long t2 = clock64();
  if (threadIdx.x == 0) { t[blockIdx.x * 2] = t1 - t0; t[blockIdx.x * 2 + 1] = t2 - t0; }
}

template <int N>
__global__ __launch_bounds__(512) void wide(const float* __restrict__ src, float* __restrict__ out, long long* t) {
  const long long t0 = clock64();
  f32x4 v[N];
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src + (blockIdx.x * 13 % 7) * 4096);
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = s4[j * 512 + threadIdx.x];
  const long long t1 = clock64();
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < N; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  const long long t2 = clock64();
  if (threadIdx.x == 0) { t[blockIdx.x * 2] = t1 - t0; t[blockIdx.x * 2 + 1] = t2 - t0; }
}

template <class F>
void run(const char* name, F launch, long long* dt, int blocks) {
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * 2);
  hipMemcpy(h.data(), dt, sizeof(long long) * blocks * 2, hipMemcpyDeviceToHost);
  long long a = 0, b = 0, mx = 0;
  for (int i = 0; i < blocks; ++i) { a += h[2 * i]; b += h[2 * i + 1]; if (h[2 * i + 1] > mx) mx = h[2 * i + 1]; }
  printf("%-28s issue %6lld cyc   data back %6lld cyc (max %lld)\n", name, a / blocks, b / blocks, mx);
}

int main() {
  const int blocks = 108;
  float *src, *out;
  long long* dt;
  hipMalloc(&src, sizeof(float) * (7 * 4096 + 32 * 2048 * 4));
  hipMalloc(&out, sizeof(float) * blocks * 512);
  hipMalloc(&dt, sizeof(long long) * blocks * 2);
  hipMemset(src, 0, sizeof(float) * (7 * 4096 + 32 * 2048 * 4));
  run("20 x dword  / thread (40KB)", [&] { hipLaunchKernelGGL(narrow<20>, dim3(blocks), dim3(512), 0, 0, src, out, dt); }, dt, blocks);
  run("28 x dword  / thread (56KB)", [&] { hipLaunchKernelGGL(narrow<28>, dim3(blocks), dim3(512), 0, 0, src, out, dt); }, dt, blocks);
  run(" 5 x dwordx4/ thread (40KB)", [&] { hipLaunchKernelGGL(wide<5>, dim3(blocks), dim3(512), 0, 0, src, out, dt); }, dt, blocks);
  run(" 7 x dwordx4/ thread (56KB)", [&] { hipLaunchKernelGGL(wide<7>, dim3(blocks), dim3(512), 0, 0, src, out, dt); }, dt, blocks);
  run("12 x dwordx4/ thread (96KB)", [&] { hipLaunchKernelGGL(wide<12>, dim3(blocks), dim3(512), 0, 0, src, out, dt); }, dt, blocks);
  return 0;
}
