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
#define DTF_ACT_TANH 2      // dense-layer kernels only (gemm.hip)
#define DTF_ACT_NONE (-1)   // dense-layer kernels only: a linear layer

// Error codes returned by every extern "C" entry point (0 = success).
#define DTF_OK 0
#define DTF_ERR_SHAPE 1
#define DTF_ERR_NULL 2
#define DTF_ERR_LAUNCH 3
#define DTF_ERR_UNSUPPORTED 4

__device__ __forceinline__ float dtf_act_fwd(float z, int act) {
  if (act == DTF_ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));
  if (act == DTF_ACT_TANH) return tanhf(z);
  return fmaxf(z, 0.0f);
}

// Derivative of the activation expressed through its OUTPUT a = act(z); the
// pre-activation never has to be kept.
__device__ __forceinline__ float dtf_act_bwd(float a, int act) {
  if (act == DTF_ACT_SIGMOID) return a * (1.0f - a);
  if (act == DTF_ACT_TANH) return 1.0f - a * a;
  if (act == DTF_ACT_NONE) return 1.0f;
  return a > 0.0f ? 1.0f : 0.0f;
}

// Index of the mini-batch a kernel works on.  `cursor` is a device-resident
// count of consumed batches that no kernel of a captured chunk modifies;
// `bias` is the position of the step inside the chunk, baked into the graph
// node.  A null cursor means "batch `bias`".
// Reductions inside a row of 16 lanes with DPP moves (one VALU instruction per step, no LDS crossbar): xor 1 and xor 2
// as quad permutations; after them quads are uniform, so the half mirror (lane i -> 7 - i) and the row mirror
// (i -> 15 - i) stand in for xor 4 and xor 8.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], lane i <-> 7 - i within 8, lane i <-> 15 - i within 16
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
// argmax with ties resolved to the smallest index (tf.argmax); the comparison is a total order, so every lane of the
// row ends with the same pair
__device__ __forceinline__ int row16_argmax(float v, int idx) {
  argmax_step<kXor1>(v, idx);
  argmax_step<kXor2>(v, idx);
  argmax_step<kHalfMirror>(v, idx);
  argmax_step<kRowMirror>(v, idx);
  return idx;
}

__device__ __forceinline__ long dtf_batch_index(const int* cursor, int bias, int batches_per_epoch) {
  long c = cursor ? (long)(*cursor) : 0L;
  long b = c + (long)bias;
  if (batches_per_epoch > 0) b %= (long)batches_per_epoch;
  return b;
}
