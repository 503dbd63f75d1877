// Native task server: the control plane and the host-memory parameter store of
// a cluster task.  It plays the role TensorFlow's in-process gRPC server plays
// for the reference (tf.train.Server, reference tfdist_between.py:17, 27-29):
//
//   * rendezvous / control: key-value blobs, monotonic counters with blocking
//     waits, blocking FIFO queues (sync tokens), shutdown;
//   * parameter-server state: named float32 buffers (the variables of a shard)
//     with server-side  buf += alpha * g  (ApplyGradientDescent executed ON the
//     parameter server, locked or lock-free = use_locking True/False) and
//     conditional accumulators with stale-gradient dropping and an atomic
//     "take N gradients, apply their mean" (SyncReplicasOptimizer,
//     tfdist_between_sync.py:65-68).
//
// One thread per connection, blocking sockets: a cluster has (workers x shards)
// connections, a handful.  Because every connection is served by its own
// thread, unlocked AXPY requests from different workers really do interleave on
// the shared buffer — the asynchronous, lock-free update semantics of the
// reference's async mode.
//
// GPU-resident shards do not move tensor bytes through this server: their data
// plane is RCCL / HIP-IPC; they use only the control operations.
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <list>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dtf_rt.h"

namespace {

constexpr uint32_t kMagic = 0x44544631;  // "DTF1"

struct ReqHeader {
  uint32_t magic;
  uint32_t op;
  uint32_t key_len;
  uint32_t flags;
  int64_t a0;
  int64_t a1;
  double f0;
  uint64_t payload_len;
};
struct RespHeader {
  int32_t status;
  uint32_t pad;
  int64_t value;
  uint64_t payload_len;
};

bool read_full(int fd, void* buf, size_t n) {
  char* p = static_cast<char*>(buf);
  while (n > 0) {
    ssize_t r = ::recv(fd, p, n, 0);
    if (r == 0) return false;
    if (r < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += r;
    n -= static_cast<size_t>(r);
  }
  return true;
}
// Header, key and payload of a message leave in ONE system call (and, small as the control messages are, in one
// TCP segment): with TCP_NODELAY three separate sends are three segments and three wake-ups of the receiver.
bool write_parts(int fd, const void* a, size_t na, const void* b, size_t nb, const void* c, size_t nc) {
  struct iovec iov[3];
  int cnt = 0;
  auto add = [&](const void* p, size_t n) {
    if (n) {
      iov[cnt].iov_base = const_cast<void*>(p);
      iov[cnt].iov_len = n;
      ++cnt;
    }
  };
  add(a, na);
  add(b, nb);
  add(c, nc);
  int first = 0;
  while (first < cnt) {
    struct msghdr msg;
    std::memset(&msg, 0, sizeof(msg));
    msg.msg_iov = iov + first;
    msg.msg_iovlen = static_cast<size_t>(cnt - first);
    ssize_t r = ::sendmsg(fd, &msg, MSG_NOSIGNAL);
    if (r < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    size_t done = static_cast<size_t>(r);
    while (first < cnt && done >= iov[first].iov_len) {
      done -= iov[first].iov_len;
      ++first;
    }
    if (first < cnt && done) {
      iov[first].iov_base = static_cast<char*>(iov[first].iov_base) + done;
      iov[first].iov_len -= done;
    }
  }
  return true;
}

struct Entry {
  std::mutex mu;
  std::condition_variable cv;
  bool has_blob = false;
  std::vector<char> blob;
  int64_t counter = 0;
  std::deque<std::vector<char>> queue;
  // float buffer (variable storage)
  std::vector<float> buf;
  // accumulator
  std::vector<float> acc;
  int64_t acc_count = 0;
  int64_t acc_step = 0;
  int64_t acc_dropped = 0;
};

struct Server {
  int listen_fd = -1;
  int port = 0;
  std::atomic<bool> stopping{false};
  std::atomic<bool> shutdown_requested{false};
  std::thread accept_thread;
  std::mutex map_mu;
  std::map<std::string, std::shared_ptr<Entry>> entries;
  // one record per accepted connection; the accept loop joins and drops the finished ones,
  // so a long-lived server does not keep a thread handle per client that ever connected
  struct Conn {
    std::thread thread;
    int fd = -1;
    std::atomic<bool> done{false};
  };
  std::mutex conn_mu;
  std::list<Conn> conns;
  std::mutex done_mu;
  std::condition_variable done_cv;
  std::atomic<uint64_t> n_requests{0}, bytes_in{0}, bytes_out{0};

  std::shared_ptr<Entry> get(const std::string& key, bool create) {
    std::lock_guard<std::mutex> g(map_mu);
    auto it = entries.find(key);
    if (it != entries.end()) return it->second;
    if (!create) return nullptr;
    auto e = std::make_shared<Entry>();
    entries[key] = e;
    return e;
  }
};

using Clock = std::chrono::steady_clock;

// The lock-free variable paths (use_locking=False): several connections update and read one
// buffer at the same time ON PURPOSE.  Aligned 4-byte loads and stores do not tear, so a
// reader sees a mix of old and new elements and a concurrent update may be lost, nothing
// worse.  They live in functions of their own, excluded from ThreadSanitizer, so that a
// sanitizer build reports every OTHER race in this file.
#if defined(__clang__) || defined(__GNUC__)
#define DTF_BENIGN_RACE __attribute__((no_sanitize("thread"), noinline))
#else
#define DTF_BENIGN_RACE
#endif

DTF_BENIGN_RACE void axpy_unlocked(float* p, const float* g, float alpha, size_t n) {
  for (size_t i = 0; i < n; ++i) p[i] += alpha * g[i];
}

DTF_BENIGN_RACE void copy_unlocked(float* dst, const float* src, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = src[i];
}

// Waits on e.cv until pred() or timeout_ms (<0 = forever) or server stop.
template <class Pred>
bool wait_until(Server* s, Entry& e, std::unique_lock<std::mutex>& lk, int64_t timeout_ms, Pred pred) {
  const auto deadline = Clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
  while (!pred()) {
    if (s->stopping.load()) return false;
    if (timeout_ms < 0) {
      e.cv.wait_for(lk, std::chrono::milliseconds(200));
    } else {
      if (Clock::now() >= deadline) return pred();
      auto left = deadline - Clock::now();
      auto slice = std::chrono::milliseconds(200);
      e.cv.wait_for(lk, left < slice ? std::chrono::duration_cast<std::chrono::milliseconds>(left) : slice);
    }
  }
  return true;
}

constexpr uint64_t kMaxBlobBytes = 64ull << 20;      // control-plane blobs: keys, handles, queue items
constexpr uint64_t kMaxBufFloats = 1ull << 28;       // parameter / accumulator buffers: 1 GiB of floats

void serve_connection(Server* s, Server::Conn* self) {
  const int fd = self->fd;
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  std::vector<char> payload, out;
  std::string key;
  // An allocation that fails inside a request (a huge but admissible size on a small machine) ends THIS
  // connection; an exception leaving a connection thread would abort the whole task.
  try {
  for (;;) {
    ReqHeader h;
    if (!read_full(fd, &h, sizeof(h))) break;
    // Sizes come from the peer: bound them per operation BEFORE anything is allocated.  Tensor payloads (writes,
    // axpy, accumulator pushes) may be as large as the largest buffer, control blobs (keys, queue items) 64 MiB.
    const bool tensor_op = h.op == DTF_OP_BUF_WRITE || h.op == DTF_OP_BUF_AXPY || h.op == DTF_OP_ACC_APPLY;
    if (h.magic != kMagic || h.key_len > (1u << 16) ||
        h.payload_len > (tensor_op ? kMaxBufFloats * sizeof(float) : kMaxBlobBytes))
      break;
    key.resize(h.key_len);
    if (h.key_len && !read_full(fd, &key[0], h.key_len)) break;
    payload.resize(h.payload_len);
    if (h.payload_len && !read_full(fd, payload.data(), h.payload_len)) break;
    s->n_requests++;
    s->bytes_in += sizeof(h) + h.key_len + h.payload_len;

    RespHeader r{DTF_RT_OK, 0, 0, 0};
    out.clear();
    switch (h.op) {
      case DTF_OP_PING:
        break;
      case DTF_OP_SET: {
        auto e = s->get(key, true);
        std::lock_guard<std::mutex> g(e->mu);
        e->blob = payload;
        e->has_blob = true;
        e->cv.notify_all();
        break;
      }
      case DTF_OP_GET: {
        auto e = s->get(key, true);
        std::unique_lock<std::mutex> lk(e->mu);
        if (wait_until(s, *e, lk, h.a0, [&] { return e->has_blob; })) {
          out = e->blob;
        } else {
          r.status = DTF_RT_TIMEOUT;
        }
        break;
      }
      case DTF_OP_DEL: {
        std::lock_guard<std::mutex> g(s->map_mu);
        r.value = static_cast<int64_t>(s->entries.erase(key));
        break;
      }
      case DTF_OP_ADD: {
        auto e = s->get(key, true);
        std::lock_guard<std::mutex> g(e->mu);
        e->counter += h.a0;
        r.value = e->counter;
        e->cv.notify_all();
        break;
      }
      case DTF_OP_WAIT_GE: {
        auto e = s->get(key, true);
        std::unique_lock<std::mutex> lk(e->mu);
        if (!wait_until(s, *e, lk, h.a1, [&] { return e->counter >= h.a0; })) r.status = DTF_RT_TIMEOUT;
        r.value = e->counter;
        break;
      }
      case DTF_OP_QPUSH: {
        if (h.flags & DTF_FLAG_FANOUT) {
          // one request feeds the queue of every worker (the chief's token release): a0 copies each into
          // key/0 .. key/(a1-1)
          const int64_t copies = h.a0 > 0 ? h.a0 : 1;
          if (h.a1 < 0 || h.a1 > 65536) {
            r.status = DTF_RT_BAD_REQUEST;
            break;
          }
          for (int64_t w = 0; w < h.a1; ++w) {
            auto e = s->get(key + "/" + std::to_string(w), true);
            std::lock_guard<std::mutex> g(e->mu);
            for (int64_t i = 0; i < copies; ++i) e->queue.push_back(payload);
            e->cv.notify_all();
          }
          r.value = h.a1;
          break;
        }
        auto e = s->get(key, true);
        std::lock_guard<std::mutex> g(e->mu);
        const int64_t copies = h.a0 > 0 ? h.a0 : 1;
        for (int64_t i = 0; i < copies; ++i) e->queue.push_back(payload);
        r.value = static_cast<int64_t>(e->queue.size());
        e->cv.notify_all();
        break;
      }
      case DTF_OP_QPOP: {
        auto e = s->get(key, true);
        std::unique_lock<std::mutex> lk(e->mu);
        if (wait_until(s, *e, lk, h.a0, [&] { return !e->queue.empty(); })) {
          out = std::move(e->queue.front());
          e->queue.pop_front();
          r.value = static_cast<int64_t>(e->queue.size());
        } else {
          r.status = DTF_RT_TIMEOUT;
        }
        break;
      }
      case DTF_OP_QPUSH_POP: {
        // deposit + wait in one request (a synchronous replica's "here is my gradient, wake me with my token")
        if (payload.size() < 2) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        uint16_t klen = 0;
        std::memcpy(&klen, payload.data(), 2);
        if (payload.size() < 2u + klen) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        const std::string pop_key(payload.data() + 2, klen);
        {
          auto e = s->get(key, true);
          std::lock_guard<std::mutex> g(e->mu);
          e->queue.emplace_back(payload.begin() + 2 + klen, payload.end());
          e->cv.notify_all();
        }
        auto e2 = s->get(pop_key, true);
        std::unique_lock<std::mutex> lk(e2->mu);
        if (wait_until(s, *e2, lk, h.a0, [&] { return !e2->queue.empty(); })) {
          out = std::move(e2->queue.front());
          e2->queue.pop_front();
          r.value = static_cast<int64_t>(e2->queue.size());
        } else {
          r.status = DTF_RT_TIMEOUT;
        }
        break;
      }
      case DTF_OP_QPOP_N: {
        if (h.a0 < 1 || h.a0 > 65536) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        auto e = s->get(key, true);
        std::unique_lock<std::mutex> lk(e->mu);
        const size_t want = static_cast<size_t>(h.a0);
        (void)wait_until(s, *e, lk, h.a1, [&] { return e->queue.size() >= want; });
        size_t n = 0;
        while (n < want && !e->queue.empty()) {
          out.insert(out.end(), e->queue.front().begin(), e->queue.front().end());
          e->queue.pop_front();
          ++n;
        }
        r.value = static_cast<int64_t>(n);
        if (n == 0) r.status = DTF_RT_TIMEOUT;
        break;
      }
      case DTF_OP_QLEN: {
        auto e = s->get(key, true);
        std::lock_guard<std::mutex> g(e->mu);
        r.value = static_cast<int64_t>(e->queue.size());
        break;
      }
      case DTF_OP_BUF_CREATE: {
        auto e = s->get(key, true);
        std::lock_guard<std::mutex> g(e->mu);
        if (h.a0 <= 0 || static_cast<uint64_t>(h.a0) > kMaxBufFloats) {
          r.status = DTF_RT_BAD_REQUEST;     // (a negative or absurd size would throw in a connection thread)
        } else if (e->buf.empty()) {
          try {
            e->buf.assign(static_cast<size_t>(h.a0), 0.0f);
            r.value = 1;
          } catch (const std::bad_alloc&) {
            r.status = DTF_RT_BAD_REQUEST;
          }
        } else if (static_cast<int64_t>(e->buf.size()) != h.a0) {
          r.status = DTF_RT_BAD_REQUEST;
        }
        e->cv.notify_all();
        break;
      }
      case DTF_OP_BUF_WRITE: {
        auto e = s->get(key, false);
        const size_t n = payload.size() / sizeof(float);
        if (!e || h.a0 < 0 || static_cast<size_t>(h.a0) + n > e->buf.size()) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        std::lock_guard<std::mutex> g(e->mu);
        std::memcpy(e->buf.data() + h.a0, payload.data(), n * sizeof(float));
        break;
      }
      case DTF_OP_BUF_READ: {
        auto e = s->get(key, false);
        if (!e || h.a0 < 0 || h.a1 < 0 || static_cast<size_t>(h.a0 + h.a1) > e->buf.size()) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        out.resize(static_cast<size_t>(h.a1) * sizeof(float));
        if (h.flags & DTF_FLAG_LOCK) {
          std::lock_guard<std::mutex> g(e->mu);
          std::memcpy(out.data(), e->buf.data() + h.a0, out.size());
        } else {
          // unlocked pull: may observe a concurrent update half applied (tensor-level
          // tearing, never element-level on aligned floats) — use_locking=False semantics
          copy_unlocked(reinterpret_cast<float*>(out.data()), e->buf.data() + h.a0, static_cast<size_t>(h.a1));
        }
        break;
      }
      case DTF_OP_BUF_AXPY: {
        auto e = s->get(key, false);
        const size_t n = payload.size() / sizeof(float);
        if (!e || h.a0 < 0 || static_cast<size_t>(h.a0) + n > e->buf.size()) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        const float alpha = static_cast<float>(h.f0);
        const float* g = reinterpret_cast<const float*>(payload.data());
        float* p = e->buf.data() + h.a0;
        if ((h.flags & DTF_FLAG_BUMP) && (h.a1 < 0 || static_cast<size_t>(h.a1) >= e->buf.size())) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        if (h.flags & DTF_FLAG_LOCK) {
          std::lock_guard<std::mutex> lk(e->mu);
          for (size_t i = 0; i < n; ++i) p[i] += alpha * g[i];
        } else {
          axpy_unlocked(p, g, alpha, n);
        }
        if (h.flags & DTF_FLAG_BUMP) {   // the step counter lives in this buffer: count the update in the same request
          std::lock_guard<std::mutex> lk(e->mu);
          e->buf[h.a1] += 1.0f;
          r.value = static_cast<int64_t>(e->buf[h.a1]);
        }
        break;
      }
      case DTF_OP_BUF_ADD_SCALAR: {
        auto e = s->get(key, false);
        if (!e || h.a0 < 0 || static_cast<size_t>(h.a0) >= e->buf.size()) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        std::lock_guard<std::mutex> g(e->mu);
        e->buf[h.a0] += static_cast<float>(h.f0);
        r.value = static_cast<int64_t>(e->buf[h.a0]);
        break;
      }
      case DTF_OP_ACC_APPLY: {
        // a0 = local_step of the pushing worker; stale gradients are dropped.
        // With DTF_FLAG_THEN_POP the key is  accumulator \x01 queue  and the request then blocks (a1 ms) for an
        // item of that queue, which is the reply payload: a synchronous replica's push and its wait for the
        // token in one round trip.  (value: bit 0 = gradient accepted.)
        std::string acc_key = key, pop_key;
        const bool then_pop = (h.flags & DTF_FLAG_THEN_POP) != 0;
        if (then_pop) {
          const size_t cut = key.find('\x01');
          if (cut == std::string::npos) {
            r.status = DTF_RT_BAD_REQUEST;
            break;
          }
          acc_key = key.substr(0, cut);
          pop_key = key.substr(cut + 1);
        }
        auto e = s->get(acc_key, true);
        const size_t n = payload.size() / sizeof(float);
        {
          std::lock_guard<std::mutex> g(e->mu);
          if (e->acc.empty()) e->acc.assign(n, 0.0f);
          if (e->acc.size() != n) {
            r.status = DTF_RT_BAD_REQUEST;
            break;
          }
          if (h.a0 < e->acc_step) {
            e->acc_dropped++;
            r.value = 0;
          } else {
            const float* gsrc = reinterpret_cast<const float*>(payload.data());
            for (size_t i = 0; i < n; ++i) e->acc[i] += gsrc[i];
            e->acc_count++;
            r.value = 1;
            e->cv.notify_all();
          }
        }
        if (then_pop) {
          auto q = s->get(pop_key, true);
          std::unique_lock<std::mutex> lk(q->mu);
          if (wait_until(s, *q, lk, h.a1, [&] { return !q->queue.empty(); })) {
            out = std::move(q->queue.front());
            q->queue.pop_front();
          } else {
            r.status = DTF_RT_TIMEOUT;
          }
        }
        break;
      }
      case DTF_OP_ACC_TAKE_APPLY: {
        // key = accumulator; payload = name of the variable buffer; a0 = required
        // gradients; a1 = timeout ms; f0 = learning rate.
        // On success: var[off..] -= lr * mean(acc); accumulator reset; its step += 1.
        // With DTF_FLAG_RELEASE the payload is  var name \0 token prefix \0 {int64 step offset, fanout, copies}
        // and the same request also bumps the step counter var[step offset] and deals `copies` tokens carrying
        // the new step into prefix/0 .. prefix/(fanout-1): the chief's whole round (take_grad, apply, step,
        // enqueue_many of SyncReplicasOptimizer) in one request.
        auto acc = s->get(key, true);
        std::string var_key(payload.begin(), payload.end());
        std::string token_prefix;
        int64_t rel[3] = {0, 0, 0};
        const bool release = (h.flags & DTF_FLAG_RELEASE) != 0;
        if (release) {
          const char* p0 = payload.data();
          const char* pe = p0 + payload.size();
          const char* z1 = static_cast<const char*>(std::memchr(p0, 0, payload.size()));
          const char* z2 = z1 ? static_cast<const char*>(std::memchr(z1 + 1, 0, static_cast<size_t>(pe - z1 - 1))) : nullptr;
          if (!z1 || !z2 || pe - (z2 + 1) != static_cast<long>(sizeof(rel))) {
            r.status = DTF_RT_BAD_REQUEST;
            break;
          }
          var_key.assign(p0, z1);
          token_prefix.assign(z1 + 1, z2);
          std::memcpy(rel, z2 + 1, sizeof(rel));
          if (rel[0] < 0 || rel[1] < 0 || rel[1] > 65536 || rel[2] < 1 || rel[2] > 65536) {
            r.status = DTF_RT_BAD_REQUEST;
            break;
          }
        }
        auto var = s->get(var_key, false);
        if (!var) {
          r.status = DTF_RT_BAD_REQUEST;
          break;
        }
        int64_t new_step = 0;
        {
          std::unique_lock<std::mutex> lk(acc->mu);
          if (!wait_until(s, *acc, lk, h.a1, [&] { return acc->acc_count >= h.a0 && h.a0 > 0; })) {
            r.status = DTF_RT_TIMEOUT;
            r.value = acc->acc_count;
            break;
          }
          const float scale = -static_cast<float>(h.f0) / static_cast<float>(acc->acc_count);
          {
            std::lock_guard<std::mutex> vg(var->mu);
            if (release && static_cast<size_t>(rel[0]) >= var->buf.size()) {
              r.status = DTF_RT_BAD_REQUEST;
              break;
            }
            const size_t n = acc->acc.size() < var->buf.size() ? acc->acc.size() : var->buf.size();
            for (size_t i = 0; i < n; ++i) var->buf[i] += scale * acc->acc[i];
            if (release) {
              var->buf[rel[0]] += 1.0f;
              new_step = static_cast<int64_t>(var->buf[rel[0]] + 0.5f);
            }
          }
          r.value = release ? new_step : acc->acc_count;
          std::fill(acc->acc.begin(), acc->acc.end(), 0.0f);
          acc->acc_count = 0;
          acc->acc_step += 1;
        }
        if (release) {
          std::vector<char> tok(sizeof(int64_t));
          std::memcpy(tok.data(), &new_step, sizeof(int64_t));
          for (int64_t w = 0; w < rel[1]; ++w) {
            auto q = s->get(token_prefix + "/" + std::to_string(w), true);
            std::lock_guard<std::mutex> g(q->mu);
            for (int64_t i = 0; i < rel[2]; ++i) q->queue.push_back(tok);
            q->cv.notify_all();
          }
        }
        break;
      }
      case DTF_OP_ACC_SET_STEP: {
        auto e = s->get(key, true);
        std::lock_guard<std::mutex> g(e->mu);
        e->acc_step = h.a0;
        break;
      }
      case DTF_OP_ACC_INFO: {
        auto e = s->get(key, true);
        std::lock_guard<std::mutex> g(e->mu);
        int64_t info[3] = {e->acc_count, e->acc_step, e->acc_dropped};
        out.assign(reinterpret_cast<char*>(info), reinterpret_cast<char*>(info) + sizeof(info));
        break;
      }
      case DTF_OP_SHUTDOWN: {
        s->shutdown_requested = true;
        std::lock_guard<std::mutex> g(s->done_mu);
        s->done_cv.notify_all();
        break;
      }
      default:
        r.status = DTF_RT_BAD_REQUEST;
    }
    r.payload_len = out.size();
    if (!write_parts(fd, &r, sizeof(r), out.data(), out.size(), nullptr, 0)) break;
    s->bytes_out += sizeof(r) + out.size();
  }
  } catch (const std::exception&) {
  }
  {
    // forget the descriptor before closing it so that stop() never touches a recycled number
    std::lock_guard<std::mutex> g(s->conn_mu);
    self->fd = -1;
  }
  ::close(fd);
  self->done.store(true);
}

void reap_finished(Server* s) {
  std::lock_guard<std::mutex> g(s->conn_mu);
  for (auto it = s->conns.begin(); it != s->conns.end();) {
    if (it->done.load()) {
      it->thread.join();
      it = s->conns.erase(it);
    } else {
      ++it;
    }
  }
}

void accept_loop(Server* s) {
  while (!s->stopping.load()) {
    pollfd pfd{s->listen_fd, POLLIN, 0};
    int pr = ::poll(&pfd, 1, 200);
    reap_finished(s);
    if (pr <= 0) continue;
    int fd = ::accept(s->listen_fd, nullptr, nullptr);
    if (fd < 0) continue;
    std::lock_guard<std::mutex> g(s->conn_mu);
    s->conns.emplace_back();
    Server::Conn* c = &s->conns.back();
    c->fd = fd;
    c->thread = std::thread(serve_connection, s, c);
  }
}

struct Client {
  int fd = -1;
  std::mutex mu;
};

int connect_once(const char* host, int port) {
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  char portstr[16];
  snprintf(portstr, sizeof(portstr), "%d", port);
  if (getaddrinfo(host, portstr, &hints, &res) != 0 || !res) return -1;
  int fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
  if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) != 0) {
    ::close(fd);
    fd = -1;
  }
  freeaddrinfo(res);
  if (fd >= 0) {
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  }
  return fd;
}

}  // namespace

extern "C" {

void* dtf_server_start(const char* host, int port, int* out_port) {
  auto* s = new Server();
  s->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
  if (s->listen_fd < 0) {
    delete s;
    return nullptr;
  }
  int one = 1;
  setsockopt(s->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_port = htons(static_cast<uint16_t>(port));
  addr.sin_addr.s_addr = htonl(INADDR_ANY);
  if (host && *host && strcmp(host, "0.0.0.0") != 0) {
    in_addr a{};
    if (inet_pton(AF_INET, host, &a) == 1) addr.sin_addr = a;  // names bind to ANY
  }
  if (::bind(s->listen_fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0 ||
      ::listen(s->listen_fd, 128) != 0) {
    ::close(s->listen_fd);
    delete s;
    return nullptr;
  }
  socklen_t len = sizeof(addr);
  getsockname(s->listen_fd, reinterpret_cast<sockaddr*>(&addr), &len);
  s->port = ntohs(addr.sin_port);
  if (out_port) *out_port = s->port;
  s->accept_thread = std::thread(accept_loop, s);
  return s;
}

int dtf_server_port(void* h) { return h ? static_cast<Server*>(h)->port : -1; }

// Blocks until a client sent SHUTDOWN (returns 1) or timeout_ms elapsed (returns 0); <0 = forever.
int dtf_server_wait(void* h, int timeout_ms) {
  auto* s = static_cast<Server*>(h);
  std::unique_lock<std::mutex> lk(s->done_mu);
  if (timeout_ms < 0) {
    s->done_cv.wait(lk, [&] { return s->shutdown_requested.load() || s->stopping.load(); });
  } else {
    s->done_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms),
                        [&] { return s->shutdown_requested.load() || s->stopping.load(); });
  }
  return s->shutdown_requested.load() ? 1 : 0;
}

void dtf_server_stats(void* h, uint64_t* out3) {
  auto* s = static_cast<Server*>(h);
  out3[0] = s->n_requests.load();
  out3[1] = s->bytes_in.load();
  out3[2] = s->bytes_out.load();
}

// Connections the server still holds a thread for (finished ones are reaped by the accept loop).
int dtf_server_connections(void* h) {
  auto* s = static_cast<Server*>(h);
  std::lock_guard<std::mutex> g(s->conn_mu);
  return static_cast<int>(s->conns.size());
}

void dtf_server_stop(void* h) {
  auto* s = static_cast<Server*>(h);
  if (!s) return;
  s->stopping = true;
  {
    std::lock_guard<std::mutex> g(s->done_mu);
    s->done_cv.notify_all();
  }
  if (s->accept_thread.joinable()) s->accept_thread.join();
  ::close(s->listen_fd);
  {
    std::lock_guard<std::mutex> g(s->conn_mu);
    for (auto& c : s->conns)
      if (c.fd >= 0) ::shutdown(c.fd, SHUT_RDWR);
  }
  {
    std::lock_guard<std::mutex> g(s->map_mu);
    for (auto& kv : s->entries) kv.second->cv.notify_all();
  }
  // the accept thread is gone, so the list no longer changes; the connection threads take
  // conn_mu on their way out, hence no lock is held while joining them
  for (auto& c : s->conns)
    if (c.thread.joinable()) c.thread.join();
  delete s;
}

void* dtf_client_connect(const char* host, int port, int timeout_ms) {
  const auto deadline = Clock::now() + std::chrono::milliseconds(timeout_ms);
  for (;;) {
    int fd = connect_once(host, port);
    if (fd >= 0) {
      auto* c = new Client();
      c->fd = fd;
      return c;
    }
    if (Clock::now() >= deadline) return nullptr;
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

void dtf_client_close(void* h) {
  auto* c = static_cast<Client*>(h);
  if (!c) return;
  if (c->fd >= 0) ::close(c->fd);
  delete c;
}

// One request/response.  Returns DTF_RT_* ; *out_len receives the response payload
// size even when it did not fit out_cap (then the status is DTF_RT_TOO_SMALL and
// the connection stays usable).
int dtf_client_call(void* h, int op, const char* key, int flags, int64_t a0, int64_t a1, double f0,
                    const void* payload, uint64_t payload_len, int64_t* out_value, void* out_buf,
                    uint64_t out_cap, uint64_t* out_len) {
  auto* c = static_cast<Client*>(h);
  if (!c || c->fd < 0) return DTF_RT_IO_ERROR;
  std::lock_guard<std::mutex> g(c->mu);
  ReqHeader req{kMagic, static_cast<uint32_t>(op), static_cast<uint32_t>(key ? strlen(key) : 0),
                static_cast<uint32_t>(flags), a0, a1, f0, payload_len};
  if (!write_parts(c->fd, &req, sizeof(req), key, req.key_len, payload, payload_len)) return DTF_RT_IO_ERROR;
  RespHeader r;
  if (!read_full(c->fd, &r, sizeof(r))) return DTF_RT_IO_ERROR;
  if (out_value) *out_value = r.value;
  if (out_len) *out_len = r.payload_len;
  if (r.payload_len) {
    if (r.payload_len <= out_cap && out_buf) {
      if (!read_full(c->fd, out_buf, r.payload_len)) return DTF_RT_IO_ERROR;
    } else {
      std::vector<char> sink(r.payload_len);
      if (!read_full(c->fd, sink.data(), sink.size())) return DTF_RT_IO_ERROR;
      return r.status == DTF_RT_OK ? DTF_RT_TOO_SMALL : r.status;
    }
  }
  return r.status;
}

}  // extern "C"
