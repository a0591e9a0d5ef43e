// la3d_json.cpp - host-side writer of the reference's per-scene box file (3dbbox.json / 3dbbox_ground.json):
//   json.dump([{"obj_id": .., "category_name": .., "center_cam": [3], "R_cam": [3][3], "dimensions": [3], "bbox3D_cam": [8][3]}, ..], f)
// reference src/util_3dbox.py:283-292 (key order :284-290, default separators ", " / ": ").  The text is produced straight from the
// packed (n, 39) float64 records of la3d_fit_* - no per-record Python objects - and is byte-identical to what Python's json module
// writes for the same numbers: floats as float.__repr__ prints them (shortest digits that round-trip; fixed notation for
// 1e-4 <= |x| < 1e16, else d.ddde+XX with at least two exponent digits).  Plain C++ (no device code): part of libla3d.so.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "la3d.h"

namespace {

// repr(float) into p (at least 32 bytes free); returns the end
char* py_float(char* p, double v) {
  if (v != v) { memcpy(p, "NaN", 3); return p + 3; }                      // json.dump(allow_nan=True)
  if (std::isinf(v)) { if (v < 0) *p++ = '-'; memcpy(p, "Infinity", 8); return p + 8; }
  if (std::signbit(v)) { *p++ = '-'; v = -v; }
  if (v == 0.0) { memcpy(p, "0.0", 3); return p + 3; }
  char b[40];
  const auto r = std::to_chars(b, b + sizeof(b), v, std::chars_format::scientific);   // d[.ddd]e+XX, shortest round-trip digits
  char* e = b;
  while (*e != 'e') ++e;
  char dig[24];
  int nd = 0;
  for (char* q = b; q < e; ++q)
    if (*q != '.') dig[nd++] = *q;
  int ex = 0;
  for (char* q = e + 2; q < r.ptr; ++q) ex = ex * 10 + (*q - '0');
  if (e[1] == '-') ex = -ex;
  const int decpt = ex + 1;                       // value = 0.d1d2.. * 10^decpt
  if (decpt <= -4 || decpt > 16) {                 // exponent form: d[.ddd]e+XX (at least two exponent digits)
    *p++ = dig[0];
    if (nd > 1) { *p++ = '.'; memcpy(p, dig + 1, (size_t)nd - 1); p += nd - 1; }
    *p++ = 'e';
    *p++ = ex < 0 ? '-' : '+';
    int a = ex < 0 ? -ex : ex;
    char t[8];
    int nt = 0;
    while (a) { t[nt++] = (char)('0' + a % 10); a /= 10; }
    while (nt < 2) t[nt++] = '0';
    while (nt) *p++ = t[--nt];
    return p;
  }
  if (decpt <= 0) {
    *p++ = '0'; *p++ = '.';
    for (int i = 0; i < -decpt; ++i) *p++ = '0';
    memcpy(p, dig, (size_t)nd); p += nd;
  } else if (decpt >= nd) {
    memcpy(p, dig, (size_t)nd); p += nd;
    for (int i = nd; i < decpt; ++i) *p++ = '0';
    *p++ = '.'; *p++ = '0';
  } else {
    memcpy(p, dig, (size_t)decpt); p += decpt;
    *p++ = '.';
    memcpy(p, dig + decpt, (size_t)(nd - decpt)); p += nd - decpt;
  }
  return p;
}

inline char* lit(char* p, const char* s) { const size_t n = strlen(s); memcpy(p, s, n); return p + n; }

char* vec(char* p, const double* v, int n) {
  *p++ = '[';
  for (int i = 0; i < n; ++i) {
    if (i) { *p++ = ','; *p++ = ' '; }
    p = py_float(p, v[i]);
  }
  *p++ = ']';
  return p;
}

char* mat(char* p, const double* v, int rows, int cols) {
  *p++ = '[';
  for (int r = 0; r < rows; ++r) {
    if (r) { *p++ = ','; *p++ = ' '; }
    p = vec(p, v + r * cols, cols);
  }
  *p++ = ']';
  return p;
}

}  // namespace

extern "C" {

// Upper bound of the text of n records whose (escaped, quoted) category names take name_bytes bytes in total, in S scenes.
int64_t la3d_3dbbox_json_bound(int64_t n, int64_t name_bytes, int64_t S) {
  return n * (39 * 27 + 160) + name_bytes + S * 4 + 64;
}

// S scenes' files in one call.  records: host f64 [*][39]; scene s owns entries [scene_off[s], scene_off[s+1]) of rows / obj_ids /
// name_ids: the record row, the object's id (decimal string in the file: the index among the kept instances of the image,
// reference :252-253 / fit_scenes) and the index of its category name in names_json (UTF-8, already JSON-escaped and quoted).
// out: host buffer of `cap` bytes (la3d_3dbbox_json_bound); text_off [S+1]: scene s's text = out[text_off[s] .. text_off[s+1]).
// Returns the bytes written, or -1 if cap is too small / an argument is missing.
int64_t la3d_format_3dbbox_json(const double* records, const int64_t* rows, const int32_t* obj_ids, const int32_t* name_ids,
                                const int64_t* scene_off, int32_t S, const char* const* names_json, char* out, int64_t cap,
                                int64_t* text_off) {
  if (S < 0 || !scene_off || !out || !text_off || (scene_off[S] > 0 && (!records || !rows || !obj_ids || !name_ids || !names_json))) return -1;
  char* p = out;
  for (int s = 0; s < S; ++s) {
    text_off[s] = p - out;
    if (out + cap - p < 8) return -1;
    *p++ = '[';
    for (int64_t k = scene_off[s]; k < scene_off[s + 1]; ++k) {
      const char* name = names_json[name_ids[k]];
      const size_t nl = strlen(name);
      if ((int64_t)(out + cap - p) < (int64_t)(39 * 27 + 160 + nl)) return -1;
      const double* r = records + rows[k] * LA3D_REC;
      if (k > scene_off[s]) { *p++ = ','; *p++ = ' '; }
      p = lit(p, "{\"obj_id\": \"");
      const auto c = std::to_chars(p, p + 12, obj_ids[k]);
      p = c.ptr;
      p = lit(p, "\", \"category_name\": ");
      memcpy(p, name, nl); p += nl;
      p = lit(p, ", \"center_cam\": ");  p = vec(p, r + 0, 3);
      p = lit(p, ", \"R_cam\": ");       p = mat(p, r + 6, 3, 3);
      p = lit(p, ", \"dimensions\": ");  p = vec(p, r + 3, 3);
      p = lit(p, ", \"bbox3D_cam\": ");  p = mat(p, r + 15, 8, 3);
      *p++ = '}';
    }
    *p++ = ']';
  }
  text_off[S] = p - out;
  return p - out;
}

// Host-side staging helper of the scene pipeline: n source planes (pageable host memory, `bytes_each` bytes each) copied into
// consecutive slots of one (pinned) destination buffer by `threads` native threads.  One foreign call instead of one Python task per
// loader thread: the copies neither hold nor fight for the interpreter lock (round 5: sixteen Python loader threads moved 21-40 GB/s
// in aggregate and slowed every short torch call of the other threads).  Returns 0, -1 on a bad argument.
int la3d_gather_planes_host(const void* const* src, int64_t n, int64_t bytes_each, void* dst, int threads) {
  if (n < 0 || bytes_each < 0 || (n > 0 && (!src || !dst))) return -1;
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  if ((int64_t)threads > n) threads = (int)(n > 0 ? n : 1);
  auto work = [&](int t) {
    for (int64_t i = n * t / threads; i < n * (t + 1) / threads; ++i)
      memcpy(static_cast<char*>(dst) + i * bytes_each, src[i], (size_t)bytes_each);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  return 0;
}

}  // extern "C"

// Identity of this binary: "LA3D_BUILD_INFO:<sha256 of the sources and headers>:<sha256 of the compile command>", put in by the build
// recipe (labelany3d_amd/_build.py) - what a measurement quotes to say which code produced it.
#ifndef LA3D_BUILD_SOURCES
#define LA3D_BUILD_SOURCES "unknown"
#define LA3D_BUILD_CMD "unknown"
#endif
extern "C" const char* la3d_build_info(void) { return "LA3D_BUILD_INFO:" LA3D_BUILD_SOURCES ":" LA3D_BUILD_CMD; }
