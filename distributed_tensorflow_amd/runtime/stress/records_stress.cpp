// Sanitizer run of the record writers (records.cpp): TensorBoard event files written by
// several threads, each with a writer of its own in ONE directory (what the workers of a
// cluster do with ./logs), and tensor-bundle checkpoints written side by side.  Every
// record is read back and its framing checked: length, masked CRC of the length, payload,
// masked CRC of the payload.
//
//   records_stress <directory> [threads] [events]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../dtf_rt.h"

namespace {

int failures = 0;

#define CHECK(cond, ...)                                        \
  do {                                                          \
    if (!(cond)) {                                              \
      std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);                        \
      std::fprintf(stderr, "\n");                               \
      __atomic_add_fetch(&failures, 1, __ATOMIC_RELAXED);       \
    }                                                           \
  } while (0)

// Number of well-formed records in an event file, -1 on a framing error.
int count_records(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return -1;
  int n = 0;
  std::vector<char> payload;
  for (;;) {
    uint64_t len;
    uint32_t crc;
    size_t got = std::fread(&len, 1, sizeof(len), f);
    if (got == 0) break;
    if (got != sizeof(len) || std::fread(&crc, 1, 4, f) != 4 || crc != dtf_crc32c_masked(&len, sizeof(len)) ||
        len > (1u << 24)) {
      n = -1;
      break;
    }
    payload.resize(len);
    if (std::fread(payload.data(), 1, len, f) != len || std::fread(&crc, 1, 4, f) != 4 ||
        crc != dtf_crc32c_masked(payload.data(), len)) {
      n = -1;
      break;
    }
    ++n;
  }
  std::fclose(f);
  return n;
}

long file_size(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return -1;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fclose(f);
  return n;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <directory> [threads] [events]\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  const int threads = argc > 2 ? std::atoi(argv[2]) : 4;
  const int events = argc > 3 ? std::atoi(argv[3]) : 300;

  CHECK(dtf_crc32c("123456789", 9) == 0xE3069283u, "crc32c check value");

  std::vector<std::string> paths(threads);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t] {
      const std::string suffix = "w" + std::to_string(t);
      void* w = dtf_events_open(dir.c_str(), suffix.c_str());
      CHECK(w != nullptr, "events_open");
      if (!w) return;
      paths[t] = dtf_events_path(w);
      const char* tags[2] = {"cost", "accuracy"};
      std::vector<int64_t> steps(events);
      std::vector<float> values(2 * events);
      for (int i = 0; i < events; ++i) {
        steps[i] = i;
        values[2 * i] = 2.3f / (1 + i);
        values[2 * i + 1] = 0.1f + 0.001f * i;
        CHECK(dtf_events_write_scalars(w, 1.5e9 + i, i, 2, tags, &values[2 * i]) == DTF_RT_OK, "write_scalars");
      }
      CHECK(dtf_events_write_scalar_series(w, 1.5e9, events, steps.data(), 2, tags, values.data()) == DTF_RT_OK,
            "write_scalar_series");
      const char raw[] = "\x09\x00\x00\x00\x00\x00\x00\xf0\x3f";     // Event{wall_time = 1.0}
      CHECK(dtf_events_write_raw(w, raw, sizeof(raw) - 1) == DTF_RT_OK, "write_raw");
      CHECK(dtf_events_flush(w) == DTF_RT_OK, "flush");
      dtf_events_close(w);

      // a checkpoint per thread, odd sizes on purpose
      const std::string prefix = dir + "/model.ckpt-" + std::to_string(t);
      void* b = dtf_bundle_begin(prefix.c_str());
      CHECK(b != nullptr, "bundle_begin");
      std::vector<float> W(784 * 100, 0.5f), bias(100, -1.0f), one(1, 42.0f);
      const int64_t sw[2] = {784, 100}, sb[1] = {100};
      CHECK(dtf_bundle_add_f32(b, "weights/Variable", W.data(), sw, 2) == DTF_RT_OK, "add W");
      CHECK(dtf_bundle_add_f32(b, "biases/Variable", bias.data(), sb, 1) == DTF_RT_OK, "add b");
      CHECK(dtf_bundle_add_f32(b, "global_step", one.data(), nullptr, 0) == DTF_RT_OK, "add scalar");
      CHECK(dtf_bundle_finish(b) == DTF_RT_OK, "bundle_finish");
      CHECK(file_size(prefix + ".data-00000-of-00001") == (long)((784 * 100 + 100 + 1) * sizeof(float)),
            "bundle data size %ld", file_size(prefix + ".data-00000-of-00001"));
      CHECK(file_size(prefix + ".index") > 0, "bundle index");
    });
  }
  for (auto& t : pool) t.join();

  for (int t = 0; t < threads; ++t) {
    for (int u = 0; u < t; ++u) CHECK(paths[t] != paths[u], "two writers share %s", paths[t].c_str());
    const int n = count_records(paths[t]);
    // file version + one per write_scalars + one per series element + the raw record
    CHECK(n == 1 + events + events + 1, "%s holds %d records", paths[t].c_str(), n);
  }
  if (failures) {
    std::fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  std::printf("OK threads=%d events=%d\n", threads, events);
  return 0;
}
