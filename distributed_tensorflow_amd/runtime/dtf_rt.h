// C ABI of libdtf_rt.so — the host-side native runtime: task server / parameter
// store, TFRecord event-file writer, tensor-bundle checkpoint writer.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// status codes
#define DTF_RT_OK 0
#define DTF_RT_TIMEOUT 1
#define DTF_RT_BAD_REQUEST 2
#define DTF_RT_IO_ERROR 3
#define DTF_RT_TOO_SMALL 4

// request flags
#define DTF_FLAG_LOCK 1
#define DTF_FLAG_RELEASE 4  /* ACC_TAKE_APPLY: also bump the step counter and deal the round's tokens */
#define DTF_FLAG_THEN_POP 8 /* ACC_APPLY: key = accumulator \x01 queue; then block for an item of that queue */
#define DTF_FLAG_FANOUT 2   /* QPUSH: key is a prefix, the payload goes to the queues key/0 .. key/(a1-1) */
#define DTF_FLAG_BUMP 16    /* BUF_AXPY: then add 1 to element a1 of the buffer (the step counter); its value is returned */

// operations
#define DTF_OP_PING 0
#define DTF_OP_SET 1
#define DTF_OP_GET 2
#define DTF_OP_ADD 3
#define DTF_OP_WAIT_GE 4
#define DTF_OP_QPUSH 5
#define DTF_OP_QPOP 6
#define DTF_OP_DEL 7
#define DTF_OP_SHUTDOWN 8
#define DTF_OP_QLEN 9
#define DTF_OP_BUF_CREATE 10
#define DTF_OP_BUF_WRITE 11
#define DTF_OP_BUF_READ 12
#define DTF_OP_BUF_AXPY 13
#define DTF_OP_ACC_APPLY 14
#define DTF_OP_ACC_TAKE_APPLY 15
#define DTF_OP_ACC_SET_STEP 16
#define DTF_OP_BUF_ADD_SCALAR 17
#define DTF_OP_ACC_INFO 18
// One round trip instead of two: push the payload's item onto queue `key`, then block (a0 ms) for an item of a
// SECOND queue.  Payload = u16 length of the second queue's name, that name, the item to push.
#define DTF_OP_QPUSH_POP 19
// Pop a0 items at once: blocks (a1 ms) until a0 are queued; on expiry returns the ones that are (status TIMEOUT
// if none).  Reply payload = the items back to back (callers use fixed-size items); value = items returned.
#define DTF_OP_QPOP_N 20

void* dtf_server_start(const char* host, int port, int* out_port);
int dtf_server_port(void* server);
int dtf_server_wait(void* server, int timeout_ms);
void dtf_server_stats(void* server, uint64_t* out3);
int dtf_server_connections(void* server);
void dtf_server_stop(void* server);

void* dtf_client_connect(const char* host, int port, int timeout_ms);
void dtf_client_close(void* client);
int dtf_client_call(void* client, int op, const char* key, int flags, int64_t a0, int64_t a1, double f0,
                    const void* payload, uint64_t payload_len, int64_t* out_value, void* out_buf,
                    uint64_t out_cap, uint64_t* out_len);

// --- checksums / record framing -------------------------------------------------
uint32_t dtf_crc32c(const void* data, uint64_t n);
uint32_t dtf_crc32c_masked(const void* data, uint64_t n);

// --- TensorBoard event files ------------------------------------------------------
// Opens <logdir>/events.out.tfevents.<secs>.<host>[.<suffix>] and writes the
// file-version record.  Returns a handle or null.
void* dtf_events_open(const char* logdir, const char* suffix);
const char* dtf_events_path(void* writer);
// One Event{wall_time, step, summary{value{tag, simple_value}...}} record.
int dtf_events_write_scalars(void* writer, double wall_time, int64_t step, int n, const char* const* tags,
                             const float* values);
// n_events records, each with the same tags: values is [n_events][n_tags], steps [n_events].
int dtf_events_write_scalar_series(void* writer, double wall_time, int n_events, const int64_t* steps,
                                   int n_tags, const char* const* tags, const float* values);
// Raw record (already-serialised Event proto), e.g. a graph description.
int dtf_events_write_raw(void* writer, const void* data, uint64_t n);
int dtf_events_flush(void* writer);
void dtf_events_close(void* writer);

// --- tensor-bundle checkpoints (<prefix>.index + <prefix>.data-00000-of-00001) -------
void* dtf_bundle_begin(const char* prefix);
// float32 tensor; names may be added in any order (the index is written sorted).
int dtf_bundle_add_f32(void* bundle, const char* name, const float* data, const int64_t* shape, int ndim);
int dtf_bundle_finish(void* bundle);   // writes both files atomically (tmp + rename); frees the handle

// --- same-node control plane of the synchronous barrier (shm_board.cpp): arrival queue + per-worker token queues in
// POSIX shared memory, futexes instead of sockets -------------------------------------------------------------------
void* dtf_board_open(const char* name, int workers, int create);     // name: "/dtf_board_..."; null on failure
void dtf_board_close(void* board);
int dtf_board_unlink(void* board);
int dtf_board_arrive(void* board, int64_t worker, int64_t step, int64_t slot);
int dtf_board_take(void* board, int want, int timeout_ms, int64_t* out, int* got);    // out: 3 x int64 per arrival
int dtf_board_release(void* board, int64_t worker, int64_t step, int copies);         // worker < 0: every worker
int dtf_board_wait_token(void* board, int64_t worker, int timeout_ms, int64_t* step);
int dtf_board_attached(void* board);

#ifdef __cplusplus
}
#endif
