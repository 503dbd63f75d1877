// C ABI of libdtf_hip.so — the hand-written gfx950 kernels behind the MLP hot
// path, the sharded SGD apply, gradient accumulation and the Philox initialiser.
// Mirrored field-for-field by ctypes structures in ops/hip_backend.py.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// One training/eval problem instance: a 2-layer MLP
//   a2 = act(x W1 + b1); y = softmax(a2 W2 + b2); loss = mean(-sum(y_ log y))
// (the model of reference tfdist_between.py:55-62).  All pointers are device
// pointers; parameter and gradient pointers normally alias ranges of the flat
// per-shard slabs.
typedef struct DtfMlpDesc {
  int32_t in_dim;             // K (784)
  int32_t hidden;             // H (100)
  int32_t classes;            // C (10), at most 16
  int32_t batch;              // B rows per step
  int32_t act;                // DTF_ACT_*
  int32_t ksplit;             // split-K factor of the first GEMM, 0 = auto
  int32_t batches_per_epoch;  // cursor modulus, 0 = no wrap
  int32_t hist_cap;           // length of loss_hist / acc_hist rings
  const float* x;             // [batches_per_epoch*B, K] resident samples
  const float* y;             // [batches_per_epoch*B, C] resident one-hot/soft labels
  float* W1;                  // [K, H]
  float* b1;                  // [H]
  float* W2;                  // [H, C]
  float* b2;                  // [C]
  float* gW1;                 // gradient outputs (may be null in fused mode)
  float* gb1;
  float* gW2;
  float* gb2;
  float* z2;                  // [B, H] workspace, must start zeroed
  float* dz2;                 // [B, H] workspace
  int32_t* cursor;            // device int: batches consumed so far (may be null)
  const float* lr;            // device float: learning rate
  float* loss_hist;           // ring of per-step mean loss (may be null)
  float* acc_hist;            // ring of per-step batch accuracy (may be null)
  float* global_step;         // device float scalar bumped by fused steps (may be null)
  float* shadow;              // [H*C + C] workspace: read-side copy of W2/b2 for fused steps
  float* zws;                 // [2][2048] zero-initialised logit ping-pong accumulators
} DtfMlpDesc;

// Extra state of the one-launch-per-step ("pipelined") chunk, see dtf_mlp_chunk_pipelined.
typedef struct DtfPipeDesc {
  float* tW1;                 // twin parameter set (same shapes as W1, b1, W2, b2): the steps of a
  float* tb1;                 //   chunk alternate between the caller's tensors and these
  float* tW2;
  float* tb2;
  float* a2b;                 // second [B, H] activation buffer (the first one is DtfMlpDesc.z2)
  float* zws3;                // [3][2048] logit accumulators
  const float* gram;          // [batches_per_epoch - 1][B][gram_ld]: gram[j][m][i] = 1 + x_{j+1}[m] . x_j[i],
                              //   zero for i >= B
  int32_t gram_ld;            // multiple of 16, B <= gram_ld <= 128
  int32_t reserved;
  float* w1t;                 // [2][hidden][in_dim]: W1 transposed, one image per parameter set (kept by the chunk's
                              //   own launches; contents between chunks are scratch)
} DtfPipeDesc;

// `batch_index` = mini-batch of the resident set (added to *cursor when a
// device cursor is used).  `parity` must alternate 0,1,0,1,... over consecutive
// steps of one descriptor (it selects the logit accumulator half).
// fwd + loss + bwd + in-place SGD; two launches for small batches, three otherwise.
int dtf_mlp_step_fused(const DtfMlpDesc* d, int batch_index, int parity, void* stream);
// fwd + loss + bwd writing gradients (scaled by grad_scale) to g*; no update.
int dtf_mlp_step_grad(const DtfMlpDesc* d, int batch_index, int parity, float grad_scale, void* stream);
// n fused steps on mini-batches first_batch .. first_batch + n - 1 of the current epoch with ONE
// launch per step: step t's backward and step t+1's forward share a launch (the forward is taken
// from the pre-update parameters plus a rank-B correction through the Gram blocks of consecutive
// batches).  Same results as n calls of dtf_mlp_step_fused up to fp32 rounding.
int dtf_mlp_chunk_pipelined_supported(int B, int K, int H, int C);
int dtf_mlp_chunk_pipelined(const DtfMlpDesc* d, const DtfPipeDesc* q, int first_batch, int n, void* stream);
// The same with chunks that chain: see the flags at the definition.
#define DTF_PIPE_CARRY_IN 1
#define DTF_PIPE_CARRY_OUT 2
int dtf_mlp_chunk_pipelined_ex(const DtfMlpDesc* d, const DtfPipeDesc* q, int first_batch, int n, int flags, void* stream);
// cursor += n  (one thread).
int dtf_advance_cursor(int32_t* cursor, int n, void* stream);

// Forward-only evaluation of M rows: accumulates sum of per-row losses into
// loss_sum[0] and the number of correct argmax predictions into correct[0]
// (both must be zeroed by the caller).  a2_ws is an [M, H] workspace.
int dtf_mlp_eval(const float* x, const float* y, int M, int K, int H, int C, int act,
                 const float* W1, const float* b1, const float* W2, const float* b2,
                 float* a2_ws, float* loss_sum, int32_t* correct, void* stream);

// Individual kernels (unit-tested against fp32 PyTorch oracles).
// out[M,H] = act(x[M,K] W[K,H] + b[H])
int dtf_linear_bias_act(const float* x, const float* W, const float* b, float* out,
                        int M, int K, int H, int act, int ksplit, void* stream);
// dW[K,H] = x[M,K]^T dz[M,H] * scale   (mode 0 store, 1: W -= lr*dW, 2: atomic add)
int dtf_dw_gemm(const float* x, const float* dz, float* out, int M, int K, int H,
                int mode, float scale, const float* lr, void* stream);

// p[i] -= lr * scale * g[i]; optionally zero g afterwards; optionally
// global_step += 1 (single thread).
int dtf_sgd_apply(float* p, float* g, int64_t n, const float* lr, float scale,
                  int zero_grad, float* global_step, void* stream);
// The same update with one float atomic per element: concurrent applies to one shard lose no update
// (use_locking=True for shards that several service streams update at once).
int dtf_sgd_apply_atomic(float* p, const float* g, int64_t n, const float* lr, float scale, float* global_step,
                         void* stream);
// p[i] -= lr * scale * (g_0[i] + ... + g_{count-1}[i]) in one launch (sum in index order); global_step += 1.
// grads: HOST array of `count` <= 16 device addresses.
int dtf_sum_apply(float* p, const uint64_t* grads, int count, int64_t n, const float* lr, float scale,
                  float* global_step, void* stream);
// Whole rounds of the chief of synchronous replicas on the shared-memory board (arrivals, dtf_sum_apply per shard, wait,
// tokens) for as long as the arrivals are one per worker, in step and of one mailbox row; see optim.hip.
int dtf_sync_rounds(void* take_fn, void* release_fn, void* board, int workers, int64_t round_step, int timeout_ms,
                    int n_shards, float* const* params, const uint64_t* const* tabs, const int64_t* numel,
                    float* const* global_step, const float* lr, int copies, void* stream, int max_rounds, int64_t* out,
                    int* got, int* rounds);
// acc[i] += g[i]
int dtf_accumulate(float* acc, const float* g, int64_t n, void* stream);
// Philox4x32-10 + Box-Muller normal fill: out[i] = mean + std * N(0,1), a pure
// function of (seed, offset + i).
int dtf_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, float mean,
                      float stddev, void* stream);
// dst[i, :] = src[idx[i], :]   (epoch shuffle of the resident dataset).  Indices outside [0, src_rows) are clamped
// and, with err != null (two ints on the device), counted in err[0]; err[1] = the first such position of idx.
int dtf_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int cols,
                    int64_t src_rows, int* err, void* stream);

// Epoch refresh in one launch (gram.hip): dst[r] = src[perm[r]] for r < nb * B (dst may be null) and
// gram[j][m][i] = 1 + x_{j+1}[m] . x_j[i] over the epoch-ordered rows (zero for B <= i < ld), fp32 MFMA.
int dtf_gather_gram_supported(int B, int K);
// ysrc / ydst (both or neither): labels [rows][C] gathered alike.
int dtf_gather_gram(const float* src, const int64_t* perm, float* dst, float* gram, int rows_src, int B, int K,
                    int nb, int ld, const float* ysrc, float* ydst, int C, void* stream);

// --- one-shot gradient all-reduce + SGD apply over peer-mapped HBM (xgmi.hip) -----------------
#define DTF_XGMI_MAX_RANKS 16
int dtf_xgmi_max_ranks(void);
// Kernels on `dev` may dereference memory of `peer` afterwards (DTF_ERR_UNSUPPORTED: no direct path).
int dtf_enable_peer_access(int dev, int peer);
// 1 when a kernel on stream_b can run while a kernel on stream_a (same device) waits for it, 0 when the two streams
// share a hardware queue (the waiting kernel gave up after timeout_us), < 0 on a HIP error.
int dtf_streams_independent(void* stream_a, void* stream_b, int timeout_us);
// Zero-filled device memory that other processes may map; uncached != 0 for words that one GPU
// polls while another writes them.
int dtf_xgmi_alloc(void** out, int64_t bytes, int uncached);
int dtf_xgmi_free(void* p);
int dtf_xgmi_export(void* p, void* handle64);
int dtf_xgmi_open(const void* handle64, void** out);
int dtf_xgmi_close(void* p);
// params -= lr * sum over ranks r of grads[r]; global_step += 1.  grads / peer_flags: HOST arrays of `world`
// device addresses (own entries local, the others mapped).  See xgmi.hip.
int dtf_xgmi_allreduce_apply(const uint64_t* grads, const uint64_t* peer_flags, const uint32_t* my_flags,
                             float* params, int64_t n, const float* lr, float* global_step, uint32_t* state,
                             int rank, int world, int timeout_ms, void* stream);
// The same with backup replicas (SyncReplicasOptimizer with replicas_to_aggregate < total_num_replicas): the sum runs
// over the first `required` replicas to arrive in this exchange, every replica applies it.  required = 0 or world:
// the call above.  state: four words (the fourth receives the mask of the replicas that counted).
int dtf_xgmi_allreduce_apply_subset(const uint64_t* grads, const uint64_t* peer_flags, const uint32_t* my_flags,
                                    float* params, int64_t n, const float* lr, float* global_step, uint32_t* state,
                                    int rank, int world, int required, int timeout_ms, void* stream);
// Two-launch step of a synchronous replica whose backward exchanges per element (mlp_step2.hip, XgParams).
typedef struct DtfXgDesc {
  uint64_t inbox[DTF_XGMI_MAX_RANKS];   // device address of every rank's inbox: 2 slots (even / odd exchange
                                        //   numbers) of 8-byte words [source rank * n + element], uncached
                                        //   memory, zero-initialised
  uint32_t* state;                      // dtf_mlp_step_xg: 4 zero-initialised words in local memory; dtf_mlp_step_async:
                                        //   null, or one word that receives global_step (a float) as it was before this
                                        //   step's push
  int64_t n;                            // elements per source row
  int64_t offW1, offb1, offW2, offb2;   // element offsets of the gradients inside a row
  int32_t rank, world, timeout_ms;
  int32_t compact;                      // dtf_mlp_step_xg: 1 = inbox words are bare 4-byte values, unwritten ones hold
                                        // 0xffffffff (the inbox must be filled with that pattern beforehand);
                                        // 2 = two-phase exchange (8-byte words): partials go to the element's owner,
                                        // which returns the new parameter value to every peer;
                                        // dtf_mlp_step_async: 1 = the forward reads the variables from the shards
                                        // themselves (no pull launch before the step; DTF_ERR_UNSUPPORTED when
                                        // shards inside the variables are shorter than 32 rows of W1)
  int64_t step_index;                   // dtf_mlp_step_async only: element that holds global_step (< 0: none)
  int64_t shard_start[DTF_XGMI_MAX_RANKS + 1];   // dtf_mlp_step_async with n == 0: slab range of every shard
  float* w1t;                           // dtf_mlp_step_xg only, may be null: [hidden][in_dim] image of W1 TRANSPOSED that
                                        // the step reads and updates INSTEAD of W1 (the caller transposes W1 into it
                                        // before the first step of a chunk and back after the last: dtf_transpose)
} DtfXgDesc;
// Asynchronous worker step: the backward pushes `shard_owner(e)[e] -= lr * g` with float atomics through the
// mappings and counts the push.  n > 0: equal shards of n floats, inbox[k] = base of rank k's slab-sized buffer;
// n == 0: shard k = slab elements [shard_start[k], shard_start[k+1]), inbox[k] = address of its first element.
int dtf_mlp_step_async(const DtfMlpDesc* d, int batch_index, int parity, const DtfXgDesc* x, void* stream);
// Gradient step of a worker whose variables live in shards it has mapped: the forward reads them from the shards (no
// pull in front), the backward writes grad_scale * gradient to d->gW1 .. d->gb2.  x: the shard table as above (state,
// step_index and compact are not used).
int dtf_mlp_step_grad_shards(const DtfMlpDesc* d, int batch_index, int parity, float grad_scale, const DtfXgDesc* x,
                             void* stream);
int dtf_transpose(const float* src, float* dst, int rows, int cols, void* stream);
int dtf_mlp_step_xg(const DtfMlpDesc* d, int batch_index, int parity, float grad_scale, const DtfXgDesc* x,
                    void* stream);

// One-sided asynchronous PS traffic over the same mappings: the slab is cut into `world` shards of `chunk`
// floats, shard k authoritative in rank k's buffer (shards[k], slab-sized).  See xgmi.hip.
int dtf_xgmi_pull(const uint64_t* shards, int world, int64_t chunk, float* local, void* stream);
int dtf_xgmi_push_sgd(const uint64_t* shards, int world, int64_t chunk, const float* grad, const float* lr,
                      int64_t step_index, void* stream);
// Shards of different sizes: shard k = slab elements [starts[k], starts[k+1]), shards[k] = address of its first element.
// Fills `bytes` bytes at p with `byte` and waits for it (initial pattern of a compact inbox).
int dtf_xgmi_fill(void* p, int byte, int64_t bytes);
int dtf_xgmi_pull_ranges(const uint64_t* shards, const int64_t* starts, int world, float* local, void* stream);

// --- dense layers of any shape (gemm.hip) ----------------------------------------------------------------------
// C[M][N] = epilogue(sum_k opA(m,k) opB(k,n)) on the fp32 matrix cores; leading dimensions in floats.
#define DTF_GEMM_NN 0        // A[m*lda + k], B[k*ldb + n]
#define DTF_GEMM_NT 1        // A[m*lda + k], B[n*ldb + k]
#define DTF_GEMM_TN 2        // A[k*lda + m], B[k*ldb + n]
#define DTF_EPI_NONE 0       // C = scale * acc
#define DTF_EPI_BIAS_ACT 1   // C = act(acc + bias[n])      (bias may be null; act DTF_ACT_* or DTF_ACT_NONE)
#define DTF_EPI_ACT_BWD 2    // C = acc * act'(aux[m*ldaux + n])   with act' written through the activation's OUTPUT
#define DTF_EPI_SGD 3        // C -= lr[0] * scale * acc
// tile: 0 = chosen from the shape, 64 or 128 = forced (tuning and tests).
int dtf_gemm_f32(int layout, int epi, const float* A, const float* B, float* C, int M, int N, int K, int64_t lda,
                 int64_t ldb, int64_t ldc, int act, const float* bias, const float* aux, int64_t ldaux,
                 const float* lr, float scale, int tile, void* stream);
// The same with a workspace (zero-filled once by the caller, then only used by these calls, one stream at a time):
// shapes with few tiles and a long K are then computed by several workgroups per tile, the last of which adds the
// partial sums in a fixed order and runs the epilogue.  splits: 0 = chosen from the shape, n = at most n.
int dtf_gemm_f32_ws(int layout, int epi, const float* A, const float* B, float* C, int M, int N, int K, int64_t lda,
                    int64_t ldb, int64_t ldc, int act, const float* bias, const float* aux, int64_t ldaux,
                    const float* lr, float scale, int tile, float* workspace, int64_t workspace_floats, int splits,
                    void* stream);
// out[n] = scale * sum_m z[m*ldz + n]  (mode DTF_EPI_NONE)  or  out[n] -= lr[0] * scale * sum  (mode DTF_EPI_SGD)
int dtf_colsum(const float* z, int64_t ldz, int M, int N, float* out, int mode, float scale, const float* lr,
               void* stream);
// Column sums of up to 16 matrices in one launch (HOST arrays of device addresses and shapes; no row segments: meant for
// batches of up to a few thousand rows).
int dtf_colsum_multi(int count, const uint64_t* z, const int64_t* ldz, const int32_t* M, const int32_t* N,
                     const uint64_t* out, int mode, float scale, const float* lr, void* stream);
// The head of a training step in one launch (M <= 8192 rows, one workgroup): *loss_out = sum of the row losses,
// *correct_out = rows classified like their labels (both STORED, in a fixed order of additions), dz as above,
// *global_step += 1 (each output may be null).
int dtf_softmax_xent_step(const float* logits, const float* labels, int M, int C, float* dz, float grad_scale,
                          float* loss_out, int32_t* correct_out, float* global_step, void* stream);
// The same head over ceil(M / 4) workgroups with nothing to clear beforehand: per-workgroup partial sums meet in `scratch`
// (>= 2 * ceil(M / 4) + 1 floats owned by the caller; the last word is a ticket that must be zero before the first launch
// and is zero again after every launch), the last workgroup to arrive adds them in a fixed order, STORES *loss_out and
// *correct_out and bumps *global_step (each output may be null).
int dtf_softmax_xent_head(const float* logits, const float* labels, int M, int C, float* dz, float grad_scale,
                          float* scratch, int64_t scratch_floats, float* loss_out, int32_t* correct_out,
                          float* global_step, void* stream);
// Start of a training step of a dense stack: *loss_sum = 0, *correct = 0, *global_step += 1 (each may be null).
int dtf_dense_step_begin(float* loss_sum, int32_t* correct, float* global_step, void* stream);
// Softmax cross-entropy over logits [M][C] against (soft) labels [M][C]: loss_sum += sum of the row losses, correct +=
// rows whose argmax matches the labels' (both accumulated, either may be null), dz = grad_scale * dLoss_row/dlogits
// (null: forward only), row_loss[m] (may be null).
int dtf_softmax_xent(const float* logits, const float* labels, int M, int C, float* dz, float grad_scale,
                     float* loss_sum, int32_t* correct, float* row_loss, void* stream);

int dtf_abi_sizeof(int which);   // sizeof of DtfMlpDesc (0), DtfPipeDesc (1), DtfXgDesc (2)
const char* dtf_hip_build_info(void);

#ifdef __cplusplus
}
#endif
