// Launcher prototypes shared between the translation units of libdtf_hip.so.
#pragma once
#include "dtf_common.h"

#define DTF_HEAD_EVAL 0
#define DTF_HEAD_GRAD 1
#define DTF_HEAD_FUSED 2

struct DtfHeadLaunch {
  float* zin;
  int pre_act, zero_zin;
  const float *b1, *W2, *b2, *ybase;
  const int* cursor;
  int bias, bpe;
  float* dz2;
  float *oW2, *ob2, *ob1;
  float grad_scale;
  const float* lr;
  float* loss_sum;
  int* correct;
  float *loss_hist, *acc_hist;
  int hist_cap;
  float* global_step;
  int M, H, C, act, mode;
};

int dtf_launch_head(const DtfHeadLaunch& a, hipStream_t stream);
// One-kernel evaluation (eval.hip): loss_sum += sum of row losses, correct += rows classified right.
int dtf_eval_fused_supported(int M, int K, int H, int C);
int dtf_launch_eval_fused(const float* x, const float* y, int M, int K, int H, int C, int act, const float* W1,
                          const float* b1, const float* W2, const float* b2, float* loss_sum, int* correct,
                          hipStream_t stream);
int dtf_head_grid(int M, int H);
int dtf_launch_fwd_gemm(const float* xbase, const int* cursor, int bias, int bpe, const float* W,
                        const float* b, float* out, int M, int K, int H, int act, int ksplit,
                        hipStream_t stream);
int dtf_launch_dw_gemm(const float* xbase, const int* cursor, int bias, int bpe, const float* dZ,
                       float* out, int M, int K, int H, int mode, float scale, const float* lr,
                       hipStream_t stream);

struct DtfMlpDesc;
bool dtf_step2_supported(int B, int K, int H, int C);
int dtf_launch_step2(const DtfMlpDesc* d, long batch_index, int parity, int mode, float grad_scale,
                     hipStream_t stream);

// p[0..n) = 0 with a kernel launch (captured faithfully by stream captures, unlike hipMemsetAsync here).
int dtf_launch_zero(float* p, long n, hipStream_t stream);
