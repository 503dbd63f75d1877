// Stress of the shared-memory barrier board (shm_board.cpp) for the sanitizers: `workers` replica threads and one
// aggregator thread run `rounds` lock-step rounds through two mappings of one board (as two processes would);
// every token must carry the round's number, a stale arrival must be visible as such, nothing may be lost.
//   board_stress WORKERS ROUNDS
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include <unistd.h>

#include "dtf_rt.h"

int main(int argc, char** argv) {
  const int workers = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 200;
  char name[96];
  snprintf(name, sizeof(name), "/dtf_board_stress_%d", (int)getpid());
  void* chief = dtf_board_open(name, workers, 1);
  if (!chief) { printf("create failed\n"); return 1; }
  if (dtf_board_open(name, workers, 1)) { printf("second exclusive create succeeded\n"); return 1; }
  void* other = dtf_board_open(name, workers, 0);
  if (!other || dtf_board_attached(other) != 2) { printf("open failed\n"); return 1; }
  if (dtf_board_open(name, workers + 1, 0)) { printf("opened with the wrong worker count\n"); return 1; }
  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (int w = 0; w < workers; ++w)
    th.emplace_back([&, w] {
      void* b = (w & 1) ? other : chief;
      int64_t local = 0;
      for (int r = 0; r < rounds; ++r) {
        if (dtf_board_arrive(b, w, local, r & 1) != DTF_RT_OK) ++bad;
        int64_t tok = -1;
        if (dtf_board_wait_token(b, w, 20000, &tok) != DTF_RT_OK || tok != local + 1) ++bad;
        local = tok;
      }
      int64_t none;
      if (dtf_board_wait_token(b, w, 5, &none) != DTF_RT_TIMEOUT) ++bad;      // no token is left over
    });
  th.emplace_back([&] {
    std::vector<int64_t> recs(3 * workers);
    for (int r = 0; r < rounds; ++r) {
      int have = 0;
      std::vector<char> seen(workers, 0);
      while (have < workers) {
        int got = 0;
        const int rc = dtf_board_take(chief, workers - have, 20000, recs.data(), &got);
        if (rc != DTF_RT_OK) { ++bad; break; }
        for (int i = 0; i < got; ++i) {
          const int64_t w = recs[3 * i], step = recs[3 * i + 1], slot = recs[3 * i + 2];
          if (w < 0 || w >= workers || step != r || slot != (r & 1) || seen[w]) ++bad;
          else seen[w] = 1;
        }
        have += got;
      }
      if (dtf_board_release(chief, -1, r + 1, 1) != DTF_RT_OK) ++bad;
    }
  });
  for (auto& t : th) t.join();
  int got = 0;
  int64_t rec[3];
  if (dtf_board_take(chief, 1, 5, rec, &got) != DTF_RT_TIMEOUT) ++bad;
  dtf_board_close(other);
  dtf_board_close(chief);
  if (dtf_board_open(name, workers, 0)) { printf("name survived the creator's close\n"); return 1; }
  printf("workers=%d rounds=%d bad=%d %s\n", workers, rounds, bad.load(), bad.load() ? "FAILED" : "OK");
  return bad.load() ? 1 : 0;
}
