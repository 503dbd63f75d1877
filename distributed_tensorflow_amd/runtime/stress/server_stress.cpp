// Concurrency stress of the task server (ps_server.cpp), meant to be built with
// -fsanitize=thread (or address) and run on the CPU:
//
//   * W worker threads, each with a connection of its own, run the synchronous-replica
//     protocol in miniature against one server: locked and lock-free variable updates,
//     accumulator pushes with a local step, a per-worker token queue, shared counters;
//   * a chief thread aggregates W gradients per round and releases the tokens;
//   * a churn thread keeps opening and closing connections;
//   * a blob reader blocks in GET before the blob exists; keys are deleted while in use.
//
// Exit code 0 and "OK" on stdout when every invariant holds; the sanitizer's own
// report (and exit code) covers the rest.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../dtf_rt.h"

namespace {

std::atomic<int> failures{0};

#define CHECK(cond, ...)                      \
  do {                                        \
    if (!(cond)) {                            \
      std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);      \
      std::fprintf(stderr, "\n");             \
      failures++;                             \
    }                                         \
  } while (0)

struct Conn {
  void* c;
  explicit Conn(int port) : c(dtf_client_connect("127.0.0.1", port, 5000)) {
    if (!c) {
      std::fprintf(stderr, "connect failed\n");
      std::exit(2);
    }
  }
  ~Conn() { dtf_client_close(c); }
  int call(int op, const std::string& key, int flags = 0, int64_t a0 = 0, int64_t a1 = 0, double f0 = 0.0,
           const void* payload = nullptr, uint64_t n = 0, int64_t* value = nullptr, void* out = nullptr,
           uint64_t cap = 0, uint64_t* out_len = nullptr) {
    return dtf_client_call(c, op, key.c_str(), flags, a0, a1, f0, payload, n, value, out, cap, out_len);
  }
};

constexpr int kN = 2048;     // floats per variable

}  // namespace

int main(int argc, char** argv) {
  const int workers = argc > 1 ? std::atoi(argv[1]) : 6;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 60;
  const float lr = 0.5f;

  int port = 0;
  void* server = dtf_server_start("127.0.0.1", 0, &port);
  if (!server) {
    std::fprintf(stderr, "server did not start\n");
    return 2;
  }

  {
    Conn boot(port);
    for (const char* k : {"var/locked", "var/free", "var/sync"}) {
      int64_t created = 0;
      CHECK(boot.call(DTF_OP_BUF_CREATE, k, 0, kN, 0, 0.0, nullptr, 0, &created) == DTF_RT_OK && created == 1,
            "create %s", k);
    }
    CHECK(boot.call(DTF_OP_ACC_SET_STEP, "acc/sync", 0, 0) == DTF_RT_OK, "acc step");
  }

  // blob reader: blocks until the blob appears
  std::thread blob_reader([&] {
    Conn c(port);
    char buf[64] = {0};
    uint64_t len = 0;
    int rc = c.call(DTF_OP_GET, "blob/init", 0, 20000, 0, 0.0, nullptr, 0, nullptr, buf, sizeof(buf), &len);
    CHECK(rc == DTF_RT_OK && len == 5 && std::memcmp(buf, "ready", 5) == 0, "blob get rc=%d len=%llu", rc,
          (unsigned long long)len);
  });

  std::atomic<bool> churn_stop{false};
  std::thread churn([&] {
    int n = 0;
    while (!churn_stop.load() && n < 400) {
      Conn c(port);
      CHECK(c.call(DTF_OP_PING, "") == DTF_RT_OK, "ping");
      c.call(DTF_OP_SET, "scratch/" + std::to_string(n % 7), 0, 0, 0, 0.0, "x", 1);
      c.call(DTF_OP_DEL, "scratch/" + std::to_string((n + 3) % 7));
      ++n;
    }
  });

  std::thread chief([&] {
    Conn c(port);
    for (int r = 0; r < rounds; ++r) {
      int64_t taken = 0;
      int rc;
      do {
        rc = c.call(DTF_OP_ACC_TAKE_APPLY, "acc/sync", 0, workers, 500, lr, "var/sync", 8, &taken);
      } while (rc == DTF_RT_TIMEOUT);
      CHECK(rc == DTF_RT_OK && taken == workers, "take_apply rc=%d taken=%lld", rc, (long long)taken);
      int64_t step = r + 1;
      if (r & 1) {   // odd rounds: one fan-out request for all workers' queues, even rounds: one request per queue
        c.call(DTF_OP_QPUSH, "tok", DTF_FLAG_FANOUT, 1, workers, 0.0, &step, sizeof(step));
      } else {
        for (int w = 0; w < workers; ++w)
          c.call(DTF_OP_QPUSH, "tok/" + std::to_string(w), 0, 1, 0, 0.0, &step, sizeof(step));
      }
    }
  });

  std::vector<std::thread> pool;
  for (int w = 0; w < workers; ++w) {
    pool.emplace_back([&, w] {
      Conn c(port);
      std::vector<float> g(kN, 1.0f), out(kN);
      int64_t local_step = 0;
      for (int r = 0; r < rounds; ++r) {
        uint64_t len = 0;
        // lock-free pull next to everybody's lock-free updates
        CHECK(c.call(DTF_OP_BUF_READ, "var/free", 0, 0, kN, 0.0, nullptr, 0, nullptr, out.data(),
                     kN * sizeof(float), &len) == DTF_RT_OK && len == kN * sizeof(float), "read free");
        CHECK(c.call(DTF_OP_BUF_AXPY, "var/free", 0, 0, 0, -0.25, g.data(), kN * sizeof(float)) == DTF_RT_OK,
              "axpy free");
        CHECK(c.call(DTF_OP_BUF_AXPY, "var/locked", DTF_FLAG_LOCK, 0, 0, -0.25, g.data(), kN * sizeof(float)) ==
                  DTF_RT_OK, "axpy locked");
        CHECK(c.call(DTF_OP_BUF_READ, "var/locked", DTF_FLAG_LOCK, 0, kN, 0.0, nullptr, 0, nullptr, out.data(),
                     kN * sizeof(float), &len) == DTF_RT_OK, "read locked");
        for (int i = 1; i < kN; i += 257) CHECK(out[i] == out[0], "locked read saw a torn tensor");
        int64_t v = 0;
        c.call(DTF_OP_ADD, "counter", 0, 1, 0, 0.0, nullptr, 0, &v);
        CHECK(v >= 1, "counter");
        // synchronous round: push with the local step, wait for this worker's token
        int64_t accepted = 0;
        CHECK(c.call(DTF_OP_ACC_APPLY, "acc/sync", 0, local_step, 0, 0.0, g.data(), kN * sizeof(float),
                     &accepted) == DTF_RT_OK && accepted == 1, "acc apply (step %lld)", (long long)local_step);
        int64_t token = -1;
        CHECK(c.call(DTF_OP_QPOP, "tok/" + std::to_string(w), 0, 20000, 0, 0.0, nullptr, 0, nullptr, &token,
                     sizeof(token), &len) == DTF_RT_OK && len == sizeof(token), "token");
        CHECK(token == r + 1, "token %lld in round %d", (long long)token, r);
        local_step = token;
      }
      // a gradient computed against an old step is dropped
      int64_t accepted = 1;
      c.call(DTF_OP_ACC_APPLY, "acc/sync", 0, 0, 0, 0.0, g.data(), kN * sizeof(float), &accepted);
      CHECK(accepted == 0, "stale gradient was accepted");
    });
  }

  {
    Conn c(port);
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    c.call(DTF_OP_SET, "blob/init", 0, 0, 0, 0.0, "ready", 5);
    int64_t v = 0;
    CHECK(c.call(DTF_OP_WAIT_GE, "counter", 0, (int64_t)workers * rounds, 60000, 0.0, nullptr, 0, &v) == DTF_RT_OK,
          "wait_ge");
    CHECK(v == (int64_t)workers * rounds, "counter %lld", (long long)v);
  }
  for (auto& t : pool) t.join();
  chief.join();
  churn_stop = true;
  churn.join();
  blob_reader.join();

  {
    Conn c(port);
    std::vector<float> out(kN);
    uint64_t len = 0;
    c.call(DTF_OP_BUF_READ, "var/locked", DTF_FLAG_LOCK, 0, kN, 0.0, nullptr, 0, nullptr, out.data(),
           kN * sizeof(float), &len);
    const float want_locked = -0.25f * workers * rounds;
    for (int i = 0; i < kN; ++i) CHECK(out[i] == want_locked, "locked[%d]=%f want %f", i, out[i], want_locked);
    c.call(DTF_OP_BUF_READ, "var/sync", DTF_FLAG_LOCK, 0, kN, 0.0, nullptr, 0, nullptr, out.data(),
           kN * sizeof(float), &len);
    const float want_sync = -lr * rounds;          // mean of W unit gradients per round
    for (int i = 0; i < kN; ++i) CHECK(std::fabs(out[i] - want_sync) < 1e-4f, "sync[%d]=%f want %f", i, out[i], want_sync);
    c.call(DTF_OP_BUF_READ, "var/free", 0, 0, kN, 0.0, nullptr, 0, nullptr, out.data(), kN * sizeof(float), &len);
    // lock-free updates may lose increments but never produce garbage
    for (int i = 0; i < kN; ++i)
      CHECK(out[i] <= -0.25f && out[i] >= want_locked && out[i] * 4.0f == std::floor(out[i] * 4.0f), "free[%d]=%f", i,
            out[i]);
    int64_t info[3] = {0, 0, 0};
    c.call(DTF_OP_ACC_INFO, "acc/sync", 0, 0, 0, 0.0, nullptr, 0, nullptr, info, sizeof(info), &len);
    CHECK(info[0] == 0 && info[1] == rounds && info[2] == workers, "acc info %lld %lld %lld", (long long)info[0],
          (long long)info[1], (long long)info[2]);
    c.call(DTF_OP_SHUTDOWN, "");
  }
  CHECK(dtf_server_wait(server, 5000) == 1, "shutdown was not seen");
  // every client above has disconnected: the accept loop must have let go of their threads
  int held = -1;
  for (int i = 0; i < 50 && (held = dtf_server_connections(server)) > 0; ++i)
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  CHECK(held == 0, "%d finished connections are still held", held);
  uint64_t stats[3];
  dtf_server_stats(server, stats);
  dtf_server_stop(server);
  if (failures.load()) {
    std::fprintf(stderr, "%d check(s) failed\n", failures.load());
    return 1;
  }
  std::printf("OK workers=%d rounds=%d requests=%llu\n", workers, rounds, (unsigned long long)stats[0]);
  return 0;
}
