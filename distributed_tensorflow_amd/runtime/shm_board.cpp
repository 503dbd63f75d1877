// Control plane of the synchronous barrier for tasks that share a NODE: the arrival queue and the per-worker token
// queues of SyncReplicasOptimizer (reference tfdist_between_sync.py:65-68, 86-89) in one POSIX shared-memory segment,
// with futexes instead of sockets.
//
// Over the task server a round costs every replica two TCP round trips on its critical path (announce the gradient and
// block for the token; the chief's aggregator blocks for the arrivals, then deals the tokens): 130-160 us of a 300 us
// step.  Here an arrival is a cell of a ring that the aggregator polls / sleeps on, and a token is a cell of the
// worker's own ring: one cache-line hand-off each way, a futex wake only when the other side actually sleeps.
//
// Layout: Header | arrival ring (cells of 3 x int64) | one token ring per worker (cells of 1 x int64 in the same cell
// type).  Rings are bounded multi-producer / multi-consumer queues (a sequence number per cell orders producers and
// consumers; Vyukov's scheme); `published` counts pushes and is the futex word consumers sleep on.
// The segment is created by the chief (exclusive create, unique name), opened by the others by name, unlinked by the
// creator; a mapping stays valid after the unlink.
#include <atomic>
#include <cerrno>
#include <climits>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <new>

#include <fcntl.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "dtf_rt.h"

namespace {

constexpr uint32_t kMagic = 0x44544642;     // "DTFB"
constexpr uint32_t kCells = 256;            // per ring, a power of two: far more than rounds in flight
constexpr int kSpinNs = 40000;              // a consumer polls this long before it sleeps on the futex

struct Cell {
  std::atomic<uint64_t> seq;
  int64_t v[3];
};

struct Ring {
  alignas(64) std::atomic<uint64_t> head;        // next ticket to push
  alignas(64) std::atomic<uint64_t> tail;        // next ticket to pop
  alignas(64) std::atomic<uint32_t> published;   // pushes so far (mod 2^32): futex word
  std::atomic<uint32_t> sleepers;                // consumers inside futex_wait
  Cell cells[kCells];
};

struct Header {
  uint32_t magic, version, workers, cells;
  std::atomic<uint32_t> attached;
};

struct Board {
  void* base;
  size_t bytes;
  Header* hdr;
  Ring* arrivals;
  Ring* tokens;       // [workers]
  char name[96];
  bool creator;
};

size_t board_bytes(uint32_t workers) { return sizeof(Header) + 64 + sizeof(Ring) * (1 + (size_t)workers); }

Ring* ring_at(void* base, size_t i) {
  uintptr_t p = (reinterpret_cast<uintptr_t>(base) + sizeof(Header) + 63) & ~uintptr_t(63);
  return reinterpret_cast<Ring*>(p) + i;
}

void ring_init(Ring* r) {
  r->head.store(0, std::memory_order_relaxed);
  r->tail.store(0, std::memory_order_relaxed);
  r->published.store(0, std::memory_order_relaxed);
  r->sleepers.store(0, std::memory_order_relaxed);
  for (uint32_t i = 0; i < kCells; ++i) r->cells[i].seq.store(i, std::memory_order_relaxed);
}

long futex(std::atomic<uint32_t>* addr, int op, uint32_t val, const timespec* ts) {
  return syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), op, val, ts, nullptr, 0);
}

int64_t now_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

// false when the ring is full (kCells items nobody has popped: the consumer is gone)
bool ring_push(Ring* r, const int64_t* v, int n) {
  uint64_t pos = r->head.load(std::memory_order_relaxed);
  for (;;) {
    Cell& c = r->cells[pos & (kCells - 1)];
    const uint64_t seq = c.seq.load(std::memory_order_acquire);
    const int64_t dif = (int64_t)seq - (int64_t)pos;
    if (dif == 0) {
      if (r->head.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
        for (int i = 0; i < 3; ++i) c.v[i] = i < n ? v[i] : 0;
        c.seq.store(pos + 1, std::memory_order_release);
        break;
      }
    } else if (dif < 0) {
      return false;
    } else {
      pos = r->head.load(std::memory_order_relaxed);
    }
  }
  r->published.fetch_add(1, std::memory_order_seq_cst);
  if (r->sleepers.load(std::memory_order_seq_cst) != 0) futex(&r->published, FUTEX_WAKE, INT_MAX, nullptr);
  return true;
}

bool ring_try_pop(Ring* r, int64_t* v) {
  uint64_t pos = r->tail.load(std::memory_order_relaxed);
  for (;;) {
    Cell& c = r->cells[pos & (kCells - 1)];
    const uint64_t seq = c.seq.load(std::memory_order_acquire);
    const int64_t dif = (int64_t)seq - (int64_t)(pos + 1);
    if (dif == 0) {
      if (r->tail.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
        for (int i = 0; i < 3; ++i) v[i] = c.v[i];
        c.seq.store(pos + kCells, std::memory_order_release);
        return true;
      }
    } else if (dif < 0) {
      return false;
    } else {
      pos = r->tail.load(std::memory_order_relaxed);
    }
  }
}

// Blocks up to timeout_ms (< 0: for ever) for one item: polls for a short while, then sleeps on the futex.
bool ring_pop(Ring* r, int64_t* v, int timeout_ms) {
  if (ring_try_pop(r, v)) return true;
  const int64_t t0 = now_ns();
  const int64_t deadline = timeout_ms < 0 ? INT64_MAX : t0 + (int64_t)timeout_ms * 1000000LL;
  for (;;) {
    const uint32_t seen = r->published.load(std::memory_order_seq_cst);
    if (ring_try_pop(r, v)) return true;
    const int64_t now = now_ns();
    if (now >= deadline) return false;
    if (now - t0 < kSpinNs) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      continue;
    }
    r->sleepers.fetch_add(1, std::memory_order_seq_cst);
    if (r->published.load(std::memory_order_seq_cst) == seen) {
      int64_t left = deadline == INT64_MAX ? 50000000LL : deadline - now;     // re-check at least every 50 ms
      if (left > 50000000LL) left = 50000000LL;
      timespec ts{(time_t)(left / 1000000000LL), (long)(left % 1000000000LL)};
      futex(&r->published, FUTEX_WAIT, seen, &ts);
    }
    r->sleepers.fetch_sub(1, std::memory_order_seq_cst);
  }
}

}  // namespace

// Creates (create != 0: exclusive, fails if the name exists) or opens the board `name` ("/dtf_board_...").
extern "C" void* dtf_board_open(const char* name, int workers, int create) {
  if (!name || name[0] != '/' || std::strlen(name) >= sizeof(Board::name) || workers < 1 || workers > 4096) return nullptr;
  const size_t bytes = board_bytes((uint32_t)workers);
  int fd = create ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
  if (fd < 0) return nullptr;
  if (create && ftruncate(fd, (off_t)bytes) != 0) {
    close(fd);
    shm_unlink(name);
    return nullptr;
  }
  if (!create) {
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) {
      close(fd);
      return nullptr;
    }
  }
  void* base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (base == MAP_FAILED) {
    if (create) shm_unlink(name);
    return nullptr;
  }
  Header* hdr = reinterpret_cast<Header*>(base);
  if (create) {
    hdr->version = 1;
    hdr->workers = (uint32_t)workers;
    hdr->cells = kCells;
    hdr->attached.store(0, std::memory_order_relaxed);
    for (int i = 0; i <= workers; ++i) ring_init(ring_at(base, (size_t)i));
    std::atomic_thread_fence(std::memory_order_seq_cst);
    hdr->magic = kMagic;        // (others open the board only after its name was published: they see it complete)
  } else if (hdr->magic != kMagic || hdr->workers != (uint32_t)workers || hdr->cells != kCells) {
    munmap(base, bytes);
    return nullptr;
  }
  hdr->attached.fetch_add(1, std::memory_order_relaxed);
  Board* b = new (std::nothrow) Board;
  if (!b) {
    munmap(base, bytes);
    return nullptr;
  }
  b->base = base;
  b->bytes = bytes;
  b->hdr = hdr;
  b->arrivals = ring_at(base, 0);
  b->tokens = ring_at(base, 1);
  std::strncpy(b->name, name, sizeof(b->name) - 1);
  b->name[sizeof(b->name) - 1] = 0;
  b->creator = create != 0;
  return b;
}

extern "C" void dtf_board_close(void* board) {
  Board* b = static_cast<Board*>(board);
  if (!b) return;
  if (b->creator) shm_unlink(b->name);
  munmap(b->base, b->bytes);
  delete b;
}

// Removes the name (the mappings stay valid); the creator calls it once every task has opened the board.
extern "C" int dtf_board_unlink(void* board) {
  Board* b = static_cast<Board*>(board);
  if (!b) return DTF_RT_BAD_REQUEST;
  return shm_unlink(b->name) == 0 ? DTF_RT_OK : DTF_RT_IO_ERROR;
}

// A replica announces a gradient: {worker, local step, mailbox slot}.
extern "C" int dtf_board_arrive(void* board, int64_t worker, int64_t step, int64_t slot) {
  Board* b = static_cast<Board*>(board);
  if (!b) return DTF_RT_BAD_REQUEST;
  const int64_t v[3] = {worker, step, slot};
  return ring_push(b->arrivals, v, 3) ? DTF_RT_OK : DTF_RT_TOO_SMALL;
}

// The aggregator takes up to `want` arrivals: blocks (timeout_ms) for the first one, then takes what else is there.
// out: 3 x int64 per arrival; *got = how many.  DTF_RT_TIMEOUT when none came.
extern "C" int dtf_board_take(void* board, int want, int timeout_ms, int64_t* out, int* got) {
  Board* b = static_cast<Board*>(board);
  if (!b || !out || !got || want < 1) return DTF_RT_BAD_REQUEST;
  *got = 0;
  if (!ring_pop(b->arrivals, out, timeout_ms)) return DTF_RT_TIMEOUT;
  int n = 1;
  while (n < want && ring_try_pop(b->arrivals, out + 3 * n)) ++n;
  *got = n;
  return DTF_RT_OK;
}

// `copies` tokens carrying `step` for worker `worker` (worker < 0: for every worker).
extern "C" int dtf_board_release(void* board, int64_t worker, int64_t step, int copies) {
  Board* b = static_cast<Board*>(board);
  if (!b || copies < 0 || worker >= (int64_t)b->hdr->workers) return DTF_RT_BAD_REQUEST;
  const int64_t first = worker < 0 ? 0 : worker, last = worker < 0 ? (int64_t)b->hdr->workers - 1 : worker;
  int rc = DTF_RT_OK;
  for (int64_t w = first; w <= last; ++w)
    for (int c = 0; c < copies; ++c)
      if (!ring_push(b->tokens + w, &step, 1)) rc = DTF_RT_TOO_SMALL;
  return rc;
}

// Worker `worker` waits for its next token; *step = the value it carries.
extern "C" int dtf_board_wait_token(void* board, int64_t worker, int timeout_ms, int64_t* step) {
  Board* b = static_cast<Board*>(board);
  if (!b || !step || worker < 0 || worker >= (int64_t)b->hdr->workers) return DTF_RT_BAD_REQUEST;
  int64_t v[3];
  if (!ring_pop(b->tokens + worker, v, timeout_ms)) return DTF_RT_TIMEOUT;
  *step = v[0];
  return DTF_RT_OK;
}

extern "C" int dtf_board_attached(void* board) {
  Board* b = static_cast<Board*>(board);
  return b ? (int)b->hdr->attached.load(std::memory_order_relaxed) : -1;
}
