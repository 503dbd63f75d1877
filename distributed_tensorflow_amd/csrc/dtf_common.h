// Shared declarations for the gfx950 (MI355X / CDNA4) kernel library.
//
// Everything here is compiled with `hipcc --offload-arch=gfx950`; the C ABI in
// dtf_api.h is what Python (ctypes) and the C++ runtime call.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DTF_ACT_SIGMOID 0
#define DTF_ACT_RELU 1

// Error codes returned by every extern "C" entry point (0 = success).
#define DTF_OK 0
#define DTF_ERR_SHAPE 1
#define DTF_ERR_NULL 2
#define DTF_ERR_LAUNCH 3
#define DTF_ERR_UNSUPPORTED 4

__device__ __forceinline__ float dtf_act_fwd(float z, int act) {
  if (act == DTF_ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));
  return fmaxf(z, 0.0f);
}

// Derivative of the activation expressed through its OUTPUT a = act(z); the
// pre-activation never has to be kept.
__device__ __forceinline__ float dtf_act_bwd(float a, int act) {
  if (act == DTF_ACT_SIGMOID) return a * (1.0f - a);
  return a > 0.0f ? 1.0f : 0.0f;
}

// Index of the mini-batch a kernel works on.  `cursor` is a device-resident
// count of consumed batches that no kernel of a captured chunk modifies;
// `bias` is the position of the step inside the chunk, baked into the graph
// node.  A null cursor means "batch `bias`".
__device__ __forceinline__ long dtf_batch_index(const int* cursor, int bias, int batches_per_epoch) {
  long c = cursor ? (long)(*cursor) : 0L;
  long b = c + (long)bias;
  if (batches_per_epoch > 0) b %= (long)batches_per_epoch;
  return b;
}
