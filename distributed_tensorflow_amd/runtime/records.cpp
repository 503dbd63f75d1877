// CRC32C, TFRecord framing, TensorBoard event writer and tensor-bundle
// checkpoint writer — the native equivalents of TensorFlow's EventsWriter and
// BundleWriter that the reference reaches through tf.summary.FileWriter
// (tfdist_between.py:84, 96) and that a Supervisor with a logdir would reach
// through tf.train.Saver.
//
// Formats:
//   TFRecord   : u64 len | u32 masked_crc(len) | bytes | u32 masked_crc(bytes)
//   Event      : proto { 1: double wall_time, 2: int64 step, 3: string file_version,
//                        5: Summary { 1: repeated Value { 1: string tag, 2: float simple_value } } }
//   Bundle     : <prefix>.data-00000-of-00001 = concatenated raw tensors;
//                <prefix>.index = LevelDB-format table: key "" -> BundleHeaderProto,
//                key <tensor name> -> BundleEntryProto{dtype, shape, shard_id, offset, size, crc32c}
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>

#include "dtf_rt.h"

namespace {

uint32_t g_table[8][256];
bool g_table_ready = false;

void init_table() {
  if (g_table_ready) return;
  const uint32_t poly = 0x82F63B78u;  // reflected Castagnoli
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ poly : c >> 1;
    g_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_table[t][i] = (g_table[t - 1][i] >> 8) ^ g_table[0][g_table[t - 1][i] & 0xFF];
  g_table_ready = true;
}

uint32_t crc32c(const void* data, uint64_t n) {
  init_table();
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {  // slicing-by-8
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_table[7][lo & 0xFF] ^ g_table[6][(lo >> 8) & 0xFF] ^ g_table[5][(lo >> 16) & 0xFF] ^
        g_table[4][lo >> 24] ^ g_table[3][hi & 0xFF] ^ g_table[2][(hi >> 8) & 0xFF] ^
        g_table[1][(hi >> 16) & 0xFF] ^ g_table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = g_table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

uint32_t mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xA282EAD8u; }

// ---- minimal protobuf encoding ----------------------------------------------------
void put_varint(std::string& s, uint64_t v) {
  while (v >= 0x80) {
    s.push_back(static_cast<char>(v | 0x80));
    v >>= 7;
  }
  s.push_back(static_cast<char>(v));
}
void put_tag(std::string& s, int field, int wire) { put_varint(s, (static_cast<uint64_t>(field) << 3) | wire); }
void put_bytes(std::string& s, int field, const std::string& b) {
  put_tag(s, field, 2);
  put_varint(s, b.size());
  s += b;
}
void put_double(std::string& s, int field, double d) {
  put_tag(s, field, 1);
  s.append(reinterpret_cast<const char*>(&d), 8);
}
void put_float(std::string& s, int field, float f) {
  put_tag(s, field, 5);
  s.append(reinterpret_cast<const char*>(&f), 4);
}
void put_fixed32(std::string& s, int field, uint32_t v) {
  put_tag(s, field, 5);
  s.append(reinterpret_cast<const char*>(&v), 4);
}
void put_int(std::string& s, int field, int64_t v) {
  put_tag(s, field, 0);
  put_varint(s, static_cast<uint64_t>(v));
}

struct EventsWriter {
  FILE* f = nullptr;
  std::string path;
};

bool write_record(FILE* f, const void* data, uint64_t n) {
  char head[12];
  std::memcpy(head, &n, 8);
  const uint32_t hc = mask(crc32c(head, 8));
  std::memcpy(head + 8, &hc, 4);
  const uint32_t dc = mask(crc32c(data, n));
  return fwrite(head, 1, 12, f) == 12 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(&dc, 1, 4, f) == 4;
}

std::string scalar_event(double wall, int64_t step, int n, const char* const* tags, const float* values) {
  std::string summary;
  for (int i = 0; i < n; ++i) {
    std::string v;
    put_bytes(v, 1, tags[i]);
    put_float(v, 2, values[i]);
    put_bytes(summary, 1, v);
  }
  std::string ev;
  put_double(ev, 1, wall);
  put_int(ev, 2, step);
  put_bytes(ev, 5, summary);
  return ev;
}

// ---- LevelDB-format table builder (single uncompressed data block per 4 KiB) -------------
struct BlockBuilder {
  std::string buf;
  std::vector<uint32_t> restarts{0};
  std::string last_key;
  int counter = 0;
  void add(const std::string& key, const std::string& value) {
    size_t shared = 0;
    if (counter < 16) {
      const size_t lim = std::min(last_key.size(), key.size());
      while (shared < lim && last_key[shared] == key[shared]) ++shared;
    } else {
      restarts.push_back(static_cast<uint32_t>(buf.size()));
      counter = 0;
    }
    put_varint(buf, shared);
    put_varint(buf, key.size() - shared);
    put_varint(buf, value.size());
    buf.append(key, shared, std::string::npos);
    buf += value;
    last_key = key;
    ++counter;
  }
  std::string finish() {
    std::string out = buf;
    for (uint32_t r : restarts) out.append(reinterpret_cast<const char*>(&r), 4);
    const uint32_t n = static_cast<uint32_t>(restarts.size());
    out.append(reinterpret_cast<const char*>(&n), 4);
    return out;
  }
  bool empty() const { return buf.empty(); }
};

struct Handle {
  uint64_t offset, size;
};

Handle emit_block(std::string& file, const std::string& contents) {
  Handle h{file.size(), contents.size()};
  file += contents;
  char trailer[5];
  trailer[0] = 0;  // no compression
  std::string crc_in = contents;
  crc_in.push_back(0);
  const uint32_t c = mask(crc32c(crc_in.data(), crc_in.size()));
  std::memcpy(trailer + 1, &c, 4);
  file.append(trailer, 5);
  return h;
}

std::string encode_handle(const Handle& h) {
  std::string s;
  put_varint(s, h.offset);
  put_varint(s, h.size);
  return s;
}

std::string build_table(const std::map<std::string, std::string>& kv) {
  std::string file;
  BlockBuilder data, index;
  std::string pending_last;
  auto flush = [&]() {
    if (data.empty()) return;
    const Handle h = emit_block(file, data.finish());
    index.add(pending_last, encode_handle(h));   // separator = last key of the block
    data = BlockBuilder();
  };
  for (const auto& e : kv) {
    data.add(e.first, e.second);
    pending_last = e.first;
    if (data.buf.size() >= 4096) flush();
  }
  flush();
  BlockBuilder meta;
  const Handle mh = emit_block(file, meta.finish());
  const Handle ih = emit_block(file, index.finish());
  std::string footer = encode_handle(mh) + encode_handle(ih);
  footer.resize(40, 0);
  const uint64_t magic = 0xdb4775248b80fb57ull;
  footer.append(reinterpret_cast<const char*>(&magic), 8);
  file += footer;
  return file;
}

struct Bundle {
  std::string prefix;
  std::string data;
  std::map<std::string, std::string> entries;
};

bool write_file_atomic(const std::string& path, const std::string& bytes) {
  const std::string tmp = path + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes.empty() || fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
  fclose(f);
  return ok && rename(tmp.c_str(), path.c_str()) == 0;
}

}  // namespace

extern "C" {

uint32_t dtf_crc32c(const void* data, uint64_t n) { return crc32c(data, n); }
uint32_t dtf_crc32c_masked(const void* data, uint64_t n) { return mask(crc32c(data, n)); }

void* dtf_events_open(const char* logdir, const char* suffix) {
  char host[256] = "localhost";
  gethostname(host, sizeof(host) - 1);
  const long secs = static_cast<long>(time(nullptr));
  std::string path = std::string(logdir) + "/events.out.tfevents." + std::to_string(secs) + "." + host;
  if (suffix && *suffix) path += std::string(".") + suffix;
  FILE* f = fopen(path.c_str(), "ab");
  if (!f) return nullptr;
  auto* w = new EventsWriter();
  w->f = f;
  w->path = path;
  std::string ev;
  put_double(ev, 1, static_cast<double>(secs));
  put_bytes(ev, 3, "brain.Event:2");
  write_record(f, ev.data(), ev.size());
  fflush(f);
  return w;
}

const char* dtf_events_path(void* h) { return static_cast<EventsWriter*>(h)->path.c_str(); }

int dtf_events_write_scalars(void* h, double wall, int64_t step, int n, const char* const* tags,
                             const float* values) {
  auto* w = static_cast<EventsWriter*>(h);
  const std::string ev = scalar_event(wall, step, n, tags, values);
  return write_record(w->f, ev.data(), ev.size()) ? DTF_RT_OK : DTF_RT_IO_ERROR;
}

int dtf_events_write_scalar_series(void* h, double wall, int n_events, const int64_t* steps, int n_tags,
                                   const char* const* tags, const float* values) {
  auto* w = static_cast<EventsWriter*>(h);
  for (int i = 0; i < n_events; ++i) {
    const std::string ev = scalar_event(wall, steps[i], n_tags, tags, values + static_cast<size_t>(i) * n_tags);
    if (!write_record(w->f, ev.data(), ev.size())) return DTF_RT_IO_ERROR;
  }
  return DTF_RT_OK;
}

int dtf_events_write_raw(void* h, const void* data, uint64_t n) {
  auto* w = static_cast<EventsWriter*>(h);
  return write_record(w->f, data, n) ? DTF_RT_OK : DTF_RT_IO_ERROR;
}

int dtf_events_flush(void* h) { return fflush(static_cast<EventsWriter*>(h)->f) == 0 ? DTF_RT_OK : DTF_RT_IO_ERROR; }

void dtf_events_close(void* h) {
  auto* w = static_cast<EventsWriter*>(h);
  if (!w) return;
  fclose(w->f);
  delete w;
}

void* dtf_bundle_begin(const char* prefix) {
  auto* b = new Bundle();
  b->prefix = prefix;
  return b;
}

int dtf_bundle_add_f32(void* h, const char* name, const float* data, const int64_t* shape, int ndim) {
  auto* b = static_cast<Bundle*>(h);
  int64_t numel = 1;
  std::string shape_pb;
  for (int i = 0; i < ndim; ++i) {
    numel *= shape[i];
    std::string dim;
    put_int(dim, 1, shape[i]);
    put_bytes(shape_pb, 2, dim);
  }
  const uint64_t nbytes = static_cast<uint64_t>(numel) * 4;
  std::string e;
  put_int(e, 1, 1);  // DT_FLOAT
  put_bytes(e, 2, shape_pb);
  // shard_id 0 is the proto default and omitted
  if (!b->data.empty()) put_int(e, 4, static_cast<int64_t>(b->data.size()));
  put_int(e, 5, static_cast<int64_t>(nbytes));
  put_fixed32(e, 6, mask(crc32c(data, nbytes)));
  b->data.append(reinterpret_cast<const char*>(data), nbytes);
  b->entries[name] = e;
  return DTF_RT_OK;
}

int dtf_bundle_finish(void* h) {
  auto* b = static_cast<Bundle*>(h);
  std::string header;
  put_int(header, 1, 1);  // num_shards; endianness LITTLE (0) is the default
  std::string version;
  put_int(version, 1, 1);  // producer
  put_bytes(header, 3, version);
  std::map<std::string, std::string> kv = b->entries;
  kv[""] = header;
  const bool ok = write_file_atomic(b->prefix + ".data-00000-of-00001", b->data) &&
                  write_file_atomic(b->prefix + ".index", build_table(kv));
  delete b;
  return ok ? DTF_RT_OK : DTF_RT_IO_ERROR;
}

}  // extern "C"
