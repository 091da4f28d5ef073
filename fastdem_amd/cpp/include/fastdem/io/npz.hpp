// fastdem/io/npz.hpp — ElevationMap <-> NumPy .npz checkpoint
// (API: fastdem/include/fastdem/io/npz.hpp:27-35; format: fastdem/src/io_npz.cpp).
//
// File format (interchangeable with the reference's files and with numpy.load):
//   * ZIP archive, every member STORED (method 0), no data descriptors;
//   * "<layer>.npy" : NPY 1.0, descr '<f4', fortran_order True, shape (rows, cols) — the raw
//     column-major MatrixXf bytes;
//   * "meta.npy"    : NPY 1.0 scalar '|S<n>' holding the JSON
//     {"version": 1, "resolution": r, "position": [x, y], "frame_id": "..", "size": [rows, cols],
//      "start_index": [r, c]}.
// Saving pulls each layer out of HBM once (one strided-gather kernel + one D2H per layer); loading
// uploads each layer once.  The reader walks the central directory (so archives written by
// numpy.savez load too, including zip64 size fields); compressed members are refused.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "fastdem/elevation_map.hpp"

namespace fastdem {
namespace io {
namespace detail {

inline uint32_t crc32(const void* data, size_t len) {  // IEEE 802.3, reflected, poly 0xEDB88320
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t n = 0; n < 256; ++n) {
      uint32_t c = n;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
      table[n] = c;
    }
    ready = true;
  }
  uint32_t crc = ~0u;
  const uint8_t* p = static_cast<const uint8_t*>(data);
  for (size_t i = 0; i < len; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return ~crc;
}

struct ByteSink {  // little-endian serialiser
  std::string b;
  void u16(uint32_t v) { b.push_back(char(v & 0xFF)); b.push_back(char((v >> 8) & 0xFF)); }
  void u32(uint32_t v) { u16(v & 0xFFFF); u16(v >> 16); }
  void raw(const void* p, size_t n) { b.append(static_cast<const char*>(p), n); }
};

// NPY 1.0 preamble: magic, version, header length, python-dict header padded with spaces so that
// the data starts on a 64-byte boundary, terminated by '\n'.
inline std::string npyPreamble(const std::string& descr, bool fortran, const std::string& shape) {
  std::string dict = "{'descr': '" + descr + "', 'fortran_order': " + (fortran ? "True" : "False") +
                     ", 'shape': " + shape + ", }";
  const size_t unpadded = 10 + dict.size() + 1;
  dict.append((64 - unpadded % 64) % 64, ' ');
  dict.push_back('\n');
  ByteSink s;
  s.raw("\x93NUMPY\x01\x00", 8);
  s.u16(uint32_t(dict.size()));
  s.raw(dict.data(), dict.size());
  return s.b;
}

class ZipStoreWriter {
 public:
  explicit ZipStoreWriter(std::ostream& os) : os_(os) {}
  void add(const std::string& name, const std::string& head, const void* body, size_t body_len) {
    Member m;
    m.name = name;
    m.size = uint32_t(head.size() + body_len);
    uint32_t crc = crc32(head.data(), head.size());
    if (body_len) {  // crc of the concatenation: continue the running value
      std::string whole = head;
      whole.append(static_cast<const char*>(body), body_len);
      crc = crc32(whole.data(), whole.size());
    }
    m.crc = crc;
    m.offset = uint32_t(os_.tellp());
    ByteSink h;
    h.u32(0x04034b50u); h.u16(20); h.u16(0); h.u16(0); h.u16(0); h.u16(0);
    h.u32(m.crc); h.u32(m.size); h.u32(m.size); h.u16(uint32_t(name.size())); h.u16(0);
    h.raw(name.data(), name.size());
    os_.write(h.b.data(), std::streamsize(h.b.size()));
    os_.write(head.data(), std::streamsize(head.size()));
    if (body_len) os_.write(static_cast<const char*>(body), std::streamsize(body_len));
    members_.push_back(m);
  }
  void finish() {
    const uint32_t cd_at = uint32_t(os_.tellp());
    ByteSink c;
    for (const Member& m : members_) {
      c.u32(0x02014b50u); c.u16(20); c.u16(20); c.u16(0); c.u16(0); c.u16(0); c.u16(0);
      c.u32(m.crc); c.u32(m.size); c.u32(m.size); c.u16(uint32_t(m.name.size()));
      c.u16(0); c.u16(0); c.u16(0); c.u16(0); c.u32(0); c.u32(m.offset);
      c.raw(m.name.data(), m.name.size());
    }
    const uint32_t cd_len = uint32_t(c.b.size());
    c.u32(0x06054b50u); c.u16(0); c.u16(0); c.u16(uint32_t(members_.size())); c.u16(uint32_t(members_.size()));
    c.u32(cd_len); c.u32(cd_at); c.u16(0);
    os_.write(c.b.data(), std::streamsize(c.b.size()));
  }

 private:
  struct Member { std::string name; uint32_t crc = 0, size = 0, offset = 0; };
  std::ostream& os_;
  std::vector<Member> members_;
};

inline std::string jsonEscape(const std::string& s) {
  std::string o;
  for (char ch : s) {
    if (ch == '"' || ch == '\\') o.push_back('\\');
    o.push_back(ch);
  }
  return o;
}
constexpr int kMetadataVersion = 1;
inline std::string metadataJson(const ElevationMap& map) {
  std::ostringstream j;  // default stream formatting (6 significant digits), as the reference writes it
  const auto p = map.getPosition();
  const auto sz = map.getSize();
  const auto st = map.getStartIndex();
  j << "{\"version\": " << kMetadataVersion << ", \"resolution\": " << map.getResolution() << ", \"position\": ["
    << p(0) << ", " << p(1) << "], \"frame_id\": \"" << jsonEscape(map.getFrameId()) << "\", \"size\": [" << sz(0)
    << ", " << sz(1) << "], \"start_index\": [" << st(0) << ", " << st(1) << "]}";
  return j.str();
}

// -------- reading --------
struct Reader {
  const std::string& d;
  bool ok(size_t at, size_t n) const { return at <= d.size() && n <= d.size() - at; }
  uint32_t u16(size_t at) const { return uint8_t(d[at]) | (uint32_t(uint8_t(d[at + 1])) << 8); }
  uint32_t u32(size_t at) const { return u16(at) | (u16(at + 2) << 16); }
  uint64_t u64(size_t at) const { return uint64_t(u32(at)) | (uint64_t(u32(at + 4)) << 32); }
};
struct Member { std::string name; size_t at = 0, size = 0; };

// Central-directory walk; false on a malformed or compressed archive.
inline bool listMembers(const std::string& file, std::vector<Member>& out) {
  const Reader r{file};
  if (file.size() < 22) return false;
  size_t eocd = std::string::npos;
  for (size_t k = file.size() - 22;; --k) {  // the comment (if any) follows the record
    if (r.u32(k) == 0x06054b50u) { eocd = k; break; }
    if (k == 0 || file.size() - k > 22 + 65535) break;
  }
  if (eocd == std::string::npos) return false;
  uint64_t count = r.u16(eocd + 10), cd_at = r.u32(eocd + 16);
  if ((count == 0xFFFF || cd_at == 0xFFFFFFFFu) && eocd >= 20 && r.u32(eocd - 20) == 0x07064b50u) {
    const uint64_t z64 = r.u64(eocd - 20 + 8);  // zip64 end-of-central-directory record
    if (!r.ok(z64, 56) || r.u32(z64) != 0x06064b50u) return false;
    count = r.u64(z64 + 32);
    cd_at = r.u64(z64 + 48);
  }
  size_t at = size_t(cd_at);
  for (uint64_t k = 0; k < count; ++k) {
    if (!r.ok(at, 46) || r.u32(at) != 0x02014b50u) return false;
    const uint32_t method = r.u16(at + 10);
    uint64_t csize = r.u32(at + 20), usize = r.u32(at + 24), local = r.u32(at + 42);
    const size_t nlen = r.u16(at + 28), xlen = r.u16(at + 30), clen = r.u16(at + 32);
    if (!r.ok(at + 46, nlen + xlen + clen)) return false;
    Member m;
    m.name = file.substr(at + 46, nlen);
    for (size_t x = at + 46 + nlen, xe = x + xlen; x + 4 <= xe;) {  // zip64 extended information
      const uint32_t id = r.u16(x), len = r.u16(x + 2);
      if (id == 0x0001) {
        size_t f = x + 4;
        if (usize == 0xFFFFFFFFu && f + 8 <= xe) { usize = r.u64(f); f += 8; }
        if (csize == 0xFFFFFFFFu && f + 8 <= xe) { csize = r.u64(f); f += 8; }
        if (local == 0xFFFFFFFFu && f + 8 <= xe) { local = r.u64(f); f += 8; }
      }
      x += 4 + len;
    }
    if (method != 0 || csize != usize) return false;  // STORE only
    if (!r.ok(size_t(local), 30) || r.u32(size_t(local)) != 0x04034b50u) return false;
    m.at = size_t(local) + 30 + r.u16(size_t(local) + 26) + r.u16(size_t(local) + 28);
    m.size = size_t(usize);
    if (!r.ok(m.at, m.size)) return false;
    out.push_back(m);
    at += 46 + nlen + xlen + clen;
  }
  return !out.empty();
}

struct NpyView {
  bool f4_matrix = false, fortran = false, bytes = false;
  long rows = 0, cols = 0;
  size_t data_at = 0, str_len = 0;
};
inline bool parseNpy(const std::string& file, const Member& m, NpyView& v) {
  const Reader r{file};
  if (m.size < 10 || std::memcmp(file.data() + m.at, "\x93NUMPY", 6) != 0) return false;
  const int major = uint8_t(file[m.at + 6]);
  size_t hlen, hat;
  if (major == 1) { hlen = r.u16(m.at + 8); hat = m.at + 10; }
  else { if (m.size < 12) return false; hlen = r.u32(m.at + 8); hat = m.at + 12; }
  if (hat + hlen > m.at + m.size) return false;
  const std::string dict = file.substr(hat, hlen);
  v.data_at = hat + hlen;
  v.fortran = dict.find("'fortran_order': True") != std::string::npos;
  const size_t sp = dict.find("'shape'");
  if (sp == std::string::npos) return false;
  const size_t lp = dict.find('(', sp), rp = dict.find(')', sp);
  if (lp == std::string::npos || rp == std::string::npos || rp < lp) return false;
  const std::string shape = dict.substr(lp + 1, rp - lp - 1);
  if (dict.find("'<f4'") != std::string::npos) {
    long a = 0, b = 0;
    if (std::sscanf(shape.c_str(), " %ld , %ld", &a, &b) != 2) return false;
    v.f4_matrix = true;
    v.rows = a;
    v.cols = b;
    return true;
  }
  const size_t s = dict.find("'|S");
  if (s != std::string::npos) {
    v.bytes = true;
    v.str_len = size_t(std::strtoul(dict.c_str() + s + 3, nullptr, 10));
    return true;
  }
  return false;
}

// tolerant "key": value extraction (the reference reads its own JSON the same way)
inline bool afterKey(const std::string& j, const std::string& key, char open, size_t& at) {
  size_t p = j.find("\"" + key + "\"");
  if (p == std::string::npos) return false;
  p = j.find(open, p);
  if (p == std::string::npos) return false;
  at = p + 1;
  return true;
}
inline bool jsonNumber(const std::string& j, const std::string& key, double& v) {
  size_t at;
  if (!afterKey(j, key, ':', at)) return false;
  char* end = nullptr;
  v = std::strtod(j.c_str() + at, &end);
  return end != j.c_str() + at;
}
inline bool jsonPair(const std::string& j, const std::string& key, double& a, double& b) {
  size_t at;
  if (!afterKey(j, key, '[', at)) return false;
  return std::sscanf(j.c_str() + at, " %lf , %lf", &a, &b) == 2;
}
inline bool jsonText(const std::string& j, const std::string& key, std::string& out) {
  size_t at;
  if (!afterKey(j, key, ':', at)) return false;
  const size_t q1 = j.find('"', at);
  if (q1 == std::string::npos) return false;
  const size_t q2 = j.find('"', q1 + 1);
  if (q2 == std::string::npos) return false;
  out = j.substr(q1 + 1, q2 - q1 - 1);
  return true;
}
inline void logError(const char* what, const std::string& file) {
  std::fprintf(stderr, "[npz_io] %s: %s\n", what, file.c_str());
}
}  // namespace detail

/// Save specific layers + metadata as NumPy .npz archive (io_npz.cpp:390-438).
inline bool saveNpz(const std::string& filename, const ElevationMap& map, const std::vector<std::string>& layer_names) {
  std::ofstream fs(filename, std::ios::binary);
  if (!fs.is_open()) {
    detail::logError("cannot create", filename);
    return false;
  }
  detail::ZipStoreWriter zip(fs);
  const int rows = map.getSize()(0), cols = map.getSize()(1);
  const std::string head =
      detail::npyPreamble("<f4", true, "(" + std::to_string(rows) + ", " + std::to_string(cols) + ")");
  for (const auto& name : layer_names) {
    if (!map.exists(name)) {
      std::fprintf(stderr, "[npz_io] layer '%s' does not exist, skipping\n", name.c_str());
      continue;
    }
    const auto& m = map.get(name);
    zip.add(name + ".npy", head, m.data(), size_t(rows) * cols * sizeof(float));
  }
  const std::string meta = detail::metadataJson(map);
  zip.add("meta.npy", detail::npyPreamble("|S" + std::to_string(meta.size()), false, "()") + meta, nullptr, 0);
  zip.finish();
  if (fs.fail()) {
    detail::logError("write failed", filename);
    return false;
  }
  return true;
}
/// Save all layers + metadata.
inline bool saveNpz(const std::string& filename, const ElevationMap& map) {
  return saveNpz(filename, map, map.getLayers());
}

/// Load ElevationMap from .npz archive (io_npz.cpp:442-620).
inline bool loadNpz(const std::string& filename, ElevationMap& map) {
  std::ifstream fs(filename, std::ios::binary);
  if (!fs.is_open()) {
    detail::logError("cannot open", filename);
    return false;
  }
  const std::string file((std::istreambuf_iterator<char>(fs)), std::istreambuf_iterator<char>());
  std::vector<detail::Member> members;
  if (!detail::listMembers(file, members)) {
    detail::logError("not a stored .npz archive", filename);
    return false;
  }
  std::string meta;
  for (const auto& m : members)
    if (m.name == "meta.npy") {
      detail::NpyView v;
      if (!detail::parseNpy(file, m, v) || !v.bytes || v.data_at + v.str_len > m.at + m.size) {
        detail::logError("invalid meta.npy", filename);
        return false;
      }
      meta = file.substr(v.data_at, v.str_len);
    }
  if (meta.empty()) {
    detail::logError("no meta.npy entry", filename);
    return false;
  }
  double version = 0, res = 0, px = 0, py = 0, nr = 0, nc = 0, sr = 0, sc = 0;
  if (detail::jsonNumber(meta, "version", version) && int(version) > detail::kMetadataVersion) {
    detail::logError("unsupported metadata version", filename);
    return false;
  }
  if (!detail::jsonNumber(meta, "resolution", res) || !detail::jsonPair(meta, "position", px, py) ||
      !detail::jsonPair(meta, "size", nr, nc)) {
    detail::logError("incomplete metadata", filename);
    return false;
  }
  std::string frame;
  detail::jsonText(meta, "frame_id", frame);
  detail::jsonPair(meta, "start_index", sr, sc);
  const int rows = int(nr), cols = int(nc);
  const float resolution = float(res);
  if (rows <= 0 || cols <= 0 || !(resolution > 0.0f)) {
    detail::logError("invalid map dimensions", filename);
    return false;
  }
  map.setFrameId(frame);
  map.setGeometry(resolution * float(rows), resolution * float(cols), resolution);
  map.setPosition(nanogrid::Position(double(float(px)), double(float(py))));
  map.setStartIndex(nanogrid::Index(int(sr), int(sc)));

  int loaded = 0;
  nanogrid::Matrix tmp(rows, cols);
  for (const auto& m : members) {
    if (m.name == "meta.npy" || m.name.size() <= 4 || m.name.compare(m.name.size() - 4, 4, ".npy") != 0) continue;
    const std::string layer = m.name.substr(0, m.name.size() - 4);
    detail::NpyView v;
    if (!detail::parseNpy(file, m, v) || !v.f4_matrix) continue;  // non-float entry
    if (v.rows != rows || v.cols != cols) continue;               // shape mismatch
    const size_t bytes = size_t(rows) * cols * sizeof(float);
    if (v.data_at + bytes > m.at + m.size) continue;              // truncated
    if (v.fortran) {
      std::memcpy(tmp.data(), file.data() + v.data_at, bytes);
    } else {  // C-order arrays written by numpy: transpose into the column-major layer
      const float* src = reinterpret_cast<const float*>(file.data() + v.data_at);
      for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
          float f;
          std::memcpy(&f, src + size_t(r) * cols + c, sizeof(float));
          tmp(r, c) = f;
        }
    }
    map.add(layer, tmp);
    ++loaded;
  }
  if (loaded == 0) {
    detail::logError("no layer data", filename);
    return false;
  }
  return true;
}

}  // namespace io
}  // namespace fastdem
