// Step-level entry points: one MLP training step = three launches
//   fwd_gemm (split-K)  ->  mlp_head  ->  dw_gemm
// replacing the ~30 kernels TensorFlow launches for the same graph
// (reference tfdist_between.py:55-70).  The functions only enqueue work on the
// given stream — no allocation, no synchronisation — so a caller can capture
// any number of steps into one hipGraph.
#include "dtf_api.h"
#include "dtf_common.h"

#include "dtf_internal.h"

namespace {

bool g_force_generic = false;   // tests/probes: run the generic three-launch path

int check_desc(const DtfMlpDesc* d, bool need_grads) {
  if (!d) return DTF_ERR_NULL;
  if (d->in_dim <= 0 || d->hidden <= 0 || d->classes <= 0 || d->classes > 16 || d->batch <= 0)
    return DTF_ERR_SHAPE;
  if (!d->x || !d->y || !d->W1 || !d->b1 || !d->W2 || !d->b2 || !d->z2 || !d->dz2) return DTF_ERR_NULL;
  if (need_grads && (!d->gW1 || !d->gb1 || !d->gW2 || !d->gb2)) return DTF_ERR_NULL;
  if (!need_grads && !d->lr) return DTF_ERR_NULL;
  return DTF_OK;
}

__global__ void zero_hist_slot_kernel(const int* cursor, int bias, float* loss_hist, float* acc_hist,
                                      int cap) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && cap > 0) {
    const long step = (cursor ? (long)(*cursor) : 0L) + bias;
    const int slot = (int)(step % cap);
    if (loss_hist) loss_hist[slot] = 0.f;
    if (acc_hist) acc_hist[slot] = 0.f;
  }
}

int run_step(const DtfMlpDesc* d, int bias, int parity, int head_mode, float grad_scale, hipStream_t stream) {
  const int B = d->batch, K = d->in_dim, H = d->hidden, C = d->classes;
  // Small-batch fast path: two launches (see mlp_step2.hip).  It addresses the
  // mini-batch by kernel argument, so it needs the cursor-free form.
  if (!d->cursor && !g_force_generic && dtf_step2_supported(B, K, H, C) &&
      d->zws && (head_mode != DTF_HEAD_FUSED || d->shadow))
    return dtf_launch_step2(d, bias, parity, head_mode, grad_scale, stream);
  const int head_grid = dtf_head_grid(B, H);
  if (head_mode == DTF_HEAD_FUSED && head_grid != 1) return DTF_ERR_UNSUPPORTED;

  // 1. z2 += x W1 (split-K partial sums into the recycled zero buffer), or the
  //    fused-epilogue form when there are enough tiles.
  int ksplit = d->ksplit;
  if (ksplit <= 0) {
    const int tiles = ((B + 31) / 32) * ((H + 31) / 32);
    ksplit = tiles >= 256 ? 1 : 7;
    const int nms = (K + 7) / 8;
    if (ksplit * 4 > nms) ksplit = (nms + 3) / 4 > 0 ? (nms + 3) / 4 : 1;
  }
  int rc = dtf_launch_fwd_gemm(d->x, d->cursor, bias, d->batches_per_epoch, d->W1, d->b1, d->z2, B, K, H,
                               d->act, ksplit, stream);
  if (rc) return rc;

  // 2. head: activation, logits, loss/accuracy, dz3, dW2/db2/db1, dz2.
  if (head_grid > 1) {
    if (d->loss_hist)
      hipLaunchKernelGGL(zero_hist_slot_kernel, dim3(1), dim3(64), 0, stream, d->cursor, bias,
                         d->loss_hist, d->acc_hist, d->hist_cap);
    if (dtf_launch_zero(d->gW2, (long)H * C, stream) || dtf_launch_zero(d->gb2, C, stream) ||
        dtf_launch_zero(d->gb1, H, stream))
      return DTF_ERR_LAUNCH;
  }
  DtfHeadLaunch h;
  h.zin = d->z2;
  h.pre_act = (ksplit == 1);
  h.zero_zin = (ksplit != 1);
  h.b1 = d->b1; h.W2 = d->W2; h.b2 = d->b2; h.ybase = d->y;
  h.cursor = d->cursor; h.bias = bias; h.bpe = d->batches_per_epoch;
  h.dz2 = d->dz2;
  if (head_mode == DTF_HEAD_FUSED) {
    h.oW2 = d->W2; h.ob2 = d->b2; h.ob1 = d->b1;
  } else {
    h.oW2 = d->gW2; h.ob2 = d->gb2; h.ob1 = d->gb1;
  }
  h.grad_scale = grad_scale;
  h.lr = d->lr;
  h.loss_sum = nullptr; h.correct = nullptr;
  h.loss_hist = d->loss_hist; h.acc_hist = d->acc_hist; h.hist_cap = d->hist_cap;
  h.global_step = d->global_step;
  h.M = B; h.H = H; h.C = C; h.act = d->act; h.mode = head_mode;
  rc = dtf_launch_head(h, stream);
  if (rc) return rc;

  // 3. dW1 = x^T dz2, stored to the gradient slab or applied in place.
  if (head_mode == DTF_HEAD_FUSED) {
    rc = dtf_launch_dw_gemm(d->x, d->cursor, bias, d->batches_per_epoch, d->dz2, d->W1, B, K, H, 1, 1.0f,
                            d->lr, stream);
  } else {
    rc = dtf_launch_dw_gemm(d->x, d->cursor, bias, d->batches_per_epoch, d->dz2, d->gW1, B, K, H, 0,
                            grad_scale, nullptr, stream);
  }
  return rc;
}

}  // namespace

extern "C" int dtf_mlp_step_fused(const DtfMlpDesc* d, int batch_index, int parity, void* stream) {
  int rc = check_desc(d, false);
  if (rc) return rc;
  return run_step(d, batch_index, parity, DTF_HEAD_FUSED, 1.0f, (hipStream_t)stream);
}

extern "C" int dtf_mlp_step_grad(const DtfMlpDesc* d, int batch_index, int parity, float grad_scale,
                                 void* stream) {
  int rc = check_desc(d, true);
  if (rc) return rc;
  return run_step(d, batch_index, parity, DTF_HEAD_GRAD, grad_scale, (hipStream_t)stream);
}

static int g_eval_fused = 1;
// 0: evaluation as first-layer GEMM + head kernel (the path for shapes the one-kernel evaluation does not take)
extern "C" void dtf_debug_set_eval_fused(int on) { g_eval_fused = on; }

extern "C" int dtf_mlp_eval(const float* x, const float* y, int M, int K, int H, int C, int act,
                            const float* W1, const float* b1, const float* W2, const float* b2,
                            float* a2_ws, float* loss_sum, int32_t* correct, void* stream) {
  if (!x || !y || !W1 || !b1 || !W2 || !b2 || !a2_ws || !loss_sum || !correct) return DTF_ERR_NULL;
  if (M <= 0 || K <= 0 || H <= 0 || C <= 0 || C > 16) return DTF_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (g_eval_fused && dtf_eval_fused_supported(M, K, H, C) && (reinterpret_cast<uintptr_t>(x) & 15u) == 0)
    return dtf_launch_eval_fused(x, y, M, K, H, C, act, W1, b1, W2, b2, loss_sum, correct, s);
  int rc = dtf_launch_fwd_gemm(x, nullptr, 0, 0, W1, b1, a2_ws, M, K, H, act, 1, s);
  if (rc) return rc;
  DtfHeadLaunch h;
  h.zin = a2_ws; h.pre_act = 1; h.zero_zin = 0;
  h.b1 = b1; h.W2 = W2; h.b2 = b2; h.ybase = y;
  h.cursor = nullptr; h.bias = 0; h.bpe = 0;
  h.dz2 = nullptr; h.oW2 = nullptr; h.ob2 = nullptr; h.ob1 = nullptr;
  h.grad_scale = 1.0f; h.lr = nullptr;
  h.loss_sum = loss_sum; h.correct = correct;
  h.loss_hist = nullptr; h.acc_hist = nullptr; h.hist_cap = 0;
  h.global_step = nullptr;
  h.M = M; h.H = H; h.C = C; h.act = act; h.mode = DTF_HEAD_EVAL;
  return dtf_launch_head(h, s);
}

// Sizes of the descriptor structs, for bindings that mirror them (0: mlp, 1: pipelined chunk, 2: cross-replica).
extern "C" int dtf_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(DtfMlpDesc);
    case 1: return (int)sizeof(DtfPipeDesc);
    case 2: return (int)sizeof(DtfXgDesc);
    default: return -1;
  }
}

extern "C" const char* dtf_hip_build_info(void) {
  return "libdtf_hip gfx950 fp32-mfma " __DATE__ " " __TIME__;
}

extern "C" void dtf_debug_force_generic_path(int on) { g_force_generic = on != 0; }
