// Host-side ASCII serialisation in the reference's exact PLY byte layout (multi-threaded).
//
// Layout restated from genply() (camera_to_world.py:112-134 == transfer_T_icp.py:46-68 ==
// pixel_to_camera.py:98-124): the header lines carry the 4-space indentation of the reference's
// triple-quoted template, every vertex row is "%.4f %.4f %.4f \n" (trailing space), the first row
// is indented 4 spaces and the file ends with "\n    ".
//
// "%.4f" must be the correctly rounded (half-to-even on the exact binary value) 4-decimal
// expansion, as CPython and glibc both produce.  Fast path: |x| < 2^40 is scaled exactly with
// 128-bit integer arithmetic (x = m*2^e, m*10^4 < 2^67); everything else goes through snprintf.
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "r3d.h"
#include "r3d_hostpool.h"
#include "r3d_pow10_table.h"

void r3d_set_error(const char* fmt, ...);

namespace {

// A chunk of formatted text.  Not a std::string: resize() would zero-fill the worst-case size (64-80 bytes per point, twice
// what the text takes) and a fresh string per slab would map, fault in and unmap that memory every time; this one hands out
// uninitialised room and keeps it for the next slab.
struct TextBuf {
  std::unique_ptr<char[]> mem;
  size_t cap = 0, len = 0;
  char* room(size_t n) {   // at least n bytes, contents undefined
    if (n > cap) {
      mem.reset(new char[n]);
      cap = n;
    }
    len = 0;
    return mem.get();
  }
  char* grow(size_t n, size_t used) {   // at least n bytes, the first `used` kept
    if (n > cap) {
      std::unique_ptr<char[]> bigger(new char[n]);
      memcpy(bigger.get(), mem.get(), used);
      mem = std::move(bigger);
      cap = n;
    }
    return mem.get();
  }
  const char* data() const { return mem.get(); }
  size_t size() const { return len; }
};

constexpr char kDigitPairs[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263"
    "646566676869707172737475767778798081828384858687888990919293949596979899";

inline char* put_uint(char* p, uint64_t v) {
  if (v < 10) {
    *p = (char)('0' + v);
    return p + 1;
  }
  char tmp[24];
  int n = 24;
  while (v >= 100) {
    const unsigned two = (unsigned)(v % 100);
    v /= 100;
    n -= 2;
    memcpy(tmp + n, kDigitPairs + 2 * two, 2);
  }
  if (v >= 10) {
    n -= 2;
    memcpy(tmp + n, kDigitPairs + 2 * v, 2);
  } else {
    tmp[--n] = (char)('0' + v);
  }
  memcpy(p, tmp + n, 24 - n);
  return p + (24 - n);
}

// writes "%.4f" of x at p, returns the new end
inline char* fmt4(char* p, double x) {
  if (std::isnan(x)) {
    memcpy(p, "nan", 3);
    return p + 3;
  }
  const bool neg = std::signbit(x);
  const double ax = std::fabs(x);
  if (!(ax < 1099511627776.0)) {  // >= 2^40 or inf: rare, take the libc path
    if (std::isinf(ax)) {
      if (neg) *p++ = '-';
      memcpy(p, "inf", 3);
      return p + 3;
    }
    return p + snprintf(p, 400, "%.4f", x);
  }
  uint64_t scaled;  // round_half_even(ax * 10^4) of the EXACT product, which is what printf's "%.4f" prints
  // Fast way: t = fl(ax * 10^4) is within half an ulp of the product; when t is further from a rounding boundary (.5) than
  // that -- delta = t * 2^-52 is twice the bound -- the product rounds the way t does.  Large t (no fraction bits left) and
  // real ties fail the test and take the exact way below.
  const double t = ax * 10000.0;
  const double r = (t + 0x1p52) - 0x1p52;   // t to the nearest integer, ties to even, for t < 2^51 (default rounding mode)
  if (t < 0x1p51 && std::fabs(t - r) < 0.5 - t * 0x1p-52) {
    scaled = (uint64_t)r;
  } else if (ax == 0.0) {
    scaled = 0;
  } else {
    uint64_t bits;
    memcpy(&bits, &ax, 8);
    const int biased = (int)(bits >> 52);
    const uint64_t m = biased ? ((bits & 0xfffffffffffffull) | (1ull << 52)) : (bits & 0xfffffffffffffull);   // ax = m * 2^e
    const int e = (biased ? biased : 1) - 1075;
    const unsigned __int128 prod = (unsigned __int128)m * 10000u;
    if (e >= 0) {
      scaled = (uint64_t)(prod << e);  // ax < 2^40 keeps this below 2^54
    } else {
      const int sh = -e;
      if (sh >= 120) {
        scaled = 0;  // far below half a unit
      } else {
        const unsigned __int128 q = prod >> sh;
        const unsigned __int128 rem = prod - (q << sh);
        const unsigned __int128 half = (unsigned __int128)1 << (sh - 1);
        scaled = (uint64_t)q;
        if (rem > half || (rem == half && (scaled & 1))) ++scaled;
      }
    }
  }
  if (neg) *p++ = '-';
  p = put_uint(p, scaled / 10000u);
  *p++ = '.';
  const unsigned frac = (unsigned)(scaled % 10000u);
  p[0] = (char)('0' + frac / 1000);
  p[1] = (char)('0' + (frac / 100) % 10);
  p[2] = (char)('0' + (frac / 10) % 10);
  p[3] = (char)('0' + frac % 10);
  return p + 4;
}

template <typename T>
void format_rows(const T* xyz, int64_t lo, int64_t hi, TextBuf* out) {
  // worst case per value below 2^40: sign + 13 digits + '.' + 4 = 19; plus separators
  char* base = out->room((size_t)(hi - lo) * 64 + 1300);
  char* p = base;
  for (int64_t i = lo; i < hi; ++i) {
    if ((size_t)(p - base) + 1300 > out->cap) {  // only after snprintf-path giants
      const size_t used = p - base;
      base = out->grow(out->cap * 2 + 1300, used);
      p = base + used;
    }
    p = fmt4(p, (double)xyz[i * 3 + 0]);
    *p++ = ' ';
    p = fmt4(p, (double)xyz[i * 3 + 1]);
    *p++ = ' ';
    p = fmt4(p, (double)xyz[i * 3 + 2]);
    *p++ = ' ';
    *p++ = '\n';
  }
  out->len = p - base;
}

// exactly eight digits of v < 10^8, leading zeros included
inline void put_8_digits(char* p, uint32_t v) {
  const uint32_t hi = v / 10000, lo = v % 10000;
  memcpy(p, kDigitPairs + 2 * (hi / 100), 2);
  memcpy(p + 2, kDigitPairs + 2 * (hi % 100), 2);
  memcpy(p + 4, kDigitPairs + 2 * (lo / 100), 2);
  memcpy(p + 6, kDigitPairs + 2 * (lo % 100), 2);
}

// decimal digits of v, 1 <= v < 10^17
inline int count_digits(uint64_t v) {
  static constexpr uint64_t kPow10[18] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull,
                                          1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull,
                                          100000000000000ull, 1000000000000000ull, 10000000000000000ull, 100000000000000000ull};
  const int t = ((64 - __builtin_clzll(v | 1)) * 1233) >> 12;   // floor(log10 v) or one more
  return t + (v >= kPow10[t] ? 1 : 0);
}

// ---- shortest decimal that reads back as the same double --------------------------------------------------------------
// What repr(float) prints (CPython: David Gay's dtoa mode 0) is the shortest digit string in the double's rounding interval,
// the closest to the double among those of that length.  Computed here by the Schubfach algorithm (R. Giulietti, "The
// Schubfach way to render doubles", 2020; also behind Java's Double.toString since JDK 19): pick k with 10^k <= 2^q <
// 10^(k+1) so that the interval holds one or two multiples of 10^k, and decide with three 64 x 128-bit products against
// g = ceil(10^-k 2^...) (r3d_pow10_table.h; round-to-odd keeps every comparison with the even candidates exact).  One
// integer routine, ~4x cheaper than std::to_chars + re-parsing its text (80 -> 20 ns per number on the build host).
// Pinned against Python's repr and std::to_chars: tests/test_host_logic.py (every table entry recomputed from its definition,
// the exponent formulas for every q, boundary doubles, random bit patterns) and tests/c/host_fuzz.cpp.
struct ShortestDecimal {
  uint64_t digits;   // value = digits * 10^exp10
  int exp10;
};

inline uint64_t round_to_odd(r3d_pow10::U128 g, uint64_t cp) {
  // bits 128..191 of the 192-bit product g * cp, with bit 0 set when anything non-zero lies below (g is rounded UP by less
  // than one unit, the product by less than cp < 2^64: a low word of 0 or 1 is an exact integer)
  const unsigned __int128 x = (unsigned __int128)cp * g.lo;
  const unsigned __int128 y = (unsigned __int128)cp * g.hi + (uint64_t)(x >> 64);
  const uint64_t y0 = (uint64_t)y, y1 = (uint64_t)(y >> 64);
  return y1 | (uint64_t)(y0 > 1);
}

inline int floor_log10_pow2(int e) { return (e * 1262611) >> 22; }                      // floor(e log10 2), |e| <= 1500
inline int floor_log10_three_quarters_pow2(int e) { return (e * 1262611 - 524031) >> 22; }   // floor(log10(3/4 2^e))
inline int floor_log2_pow10(int e) { return (e * 1741647) >> 19; }                      // floor(e log2 10), |e| <= 1233

inline ShortestDecimal shortest_decimal(uint64_t significand, int biased_exponent) {
  uint64_t c;
  int q;
  if (biased_exponent != 0) {
    c = (1ull << 52) | significand;
    q = biased_exponent - 1075;
    if (0 <= -q && -q < 53 && (c & ((1ull << -q) - 1)) == 0) return {c >> -q, 0};   // an integer below 2^53: its own digits
  } else {
    c = significand;
    q = -1074;
  }
  const bool even = (c & 1) == 0;                                      // the interval's ends round to c only for even c
  const bool lower_is_closer = significand == 0 && biased_exponent > 1;   // a power of two: the gap below is half the gap above
  const uint64_t cbl = 4 * c - 2 + (lower_is_closer ? 1 : 0), cb = 4 * c, cbr = 4 * c + 2;   // the interval and c in quarter units
  const int k = lower_is_closer ? floor_log10_three_quarters_pow2(q) : floor_log10_pow2(q);
  const int h = q + floor_log2_pow10(-k) + 1;
  const r3d_pow10::U128 g = r3d_pow10::kG[-k - r3d_pow10::kMin];
  const uint64_t vbl = round_to_odd(g, cbl << h), vb = round_to_odd(g, cb << h), vbr = round_to_odd(g, cbr << h);   // 4 x / 10^k
  const uint64_t lower = vbl + (even ? 0 : 1), upper = vbr - (even ? 0 : 1);
  const uint64_t s = vb / 4;
  if (s >= 10) {   // a multiple of 10^(k+1) inside the interval is one digit shorter
    const uint64_t sp = s / 10;
    const bool up_inside = lower <= 40 * sp, wp_inside = 40 * sp + 40 <= upper;
    if (up_inside != wp_inside) return {sp + (wp_inside ? 1 : 0), k + 1};
  }
  const bool u_inside = lower <= 4 * s, w_inside = 4 * s + 4 <= upper;
  if (u_inside != w_inside) return {s + (w_inside ? 1 : 0), k};
  const uint64_t mid = 4 * s + 2;   // both inside: the closer one, the even one on a tie
  const bool round_up = vb > mid || (vb == mid && (s & 1) != 0);
  return {s + (round_up ? 1 : 0), k};
}

// Python's repr(float) ("short" float_repr_style): shortest round-trip digits; fixed notation when
// -4 < decpt <= 16, else d[.ddd]e+XX with at least two exponent digits; ".0" appended to integers.
inline char* fmt_repr(char* p, double x) {
  if (std::isnan(x)) {
    memcpy(p, "nan", 3);
    return p + 3;
  }
  if (std::isinf(x)) {
    if (x < 0) *p++ = '-';
    memcpy(p, "inf", 3);
    return p + 3;
  }
  if (std::signbit(x)) {
    *p++ = '-';
    x = -x;
  }
  if (x == 0.0) {
    memcpy(p, "0.0", 3);
    return p + 3;
  }
  uint64_t bits;
  memcpy(&bits, &x, 8);
  ShortestDecimal d = shortest_decimal(bits & 0xfffffffffffffull, (int)(bits >> 52) & 0x7ff);
  while (d.digits % 10 == 0) {   // the algorithm stops at the shortest LENGTH; zeros at its end belong to the exponent
    d.digits /= 10;
    ++d.exp10;
  }
  // the digits, 17 places zero-padded, then the start moved past the padding: no loop over the digits, no length-dependent
  // copy (every copy below is a fixed 24 bytes -- it runs past the number's end into space the next number overwrites; the
  // callers leave 64 bytes of slack behind the last one)
  char pad[48];
  const uint64_t top = d.digits / 100000000u;
  put_8_digits(pad + 9, (uint32_t)(d.digits % 100000000u));
  put_8_digits(pad + 1, (uint32_t)(top % 100000000u));
  pad[0] = (char)('0' + top / 100000000u);
  const int nd = count_digits(d.digits);
  const char* digits = pad + 17 - nd;
  const int decpt = d.exp10 + nd;
  if (decpt <= -4 || decpt > 16) {
    p[0] = digits[0];
    if (nd > 1) {
      p[1] = '.';
      memcpy(p + 2, digits + 1, 24);
      p += nd + 1;
    } else {
      ++p;
    }
    *p++ = 'e';
    int e = decpt - 1;
    *p++ = e < 0 ? '-' : '+';
    if (e < 0) e = -e;
    if (e < 10) *p++ = '0';
    return put_uint(p, (uint64_t)e);
  }
  if (decpt <= 0) {
    memcpy(p, "0.000", 5);
    memcpy(p + 2 - decpt, digits, 24);
    return p + 2 - decpt + nd;
  }
  if (decpt >= nd) {
    memcpy(p, digits, 24);
    memcpy(p + nd, "0000000000000000", 16);
    p[decpt] = '.';
    p[decpt + 1] = '0';
    return p + decpt + 2;
  }
  memcpy(p, digits, 24);
  memcpy(p + decpt + 1, digits + decpt, 24);
  p[decpt] = '.';
  return p + nd + 1;
}

template <typename T>
void txt_rows(const T* xyz, const void* z_raw, int z_dtype, int64_t lo, int64_t hi, TextBuf* out) {
  char* base = out->room((size_t)(hi - lo) * 80 + 64);  // 3 x (sign + 17 digits + point + e-308) + 2 commas + newline < 80; fmt_repr's slack
  char* p = base;
  for (int64_t i = lo; i < hi; ++i) {
    p = fmt_repr(p, (double)xyz[i * 3 + 0]);
    *p++ = ',';
    p = fmt_repr(p, (double)xyz[i * 3 + 1]);
    *p++ = ',';
    if (z_raw)
      p = put_uint(p, z_dtype == R3D_DEPTH_U8 ? static_cast<const uint8_t*>(z_raw)[i]
                                              : static_cast<const uint16_t*>(z_raw)[i]);
    else
      p = fmt_repr(p, (double)xyz[i * 3 + 2]);
    *p++ = '\n';
  }
  out->len = p - base;
}

template <typename T>
void format_rows_rgb(const T* xyz, const unsigned char* rgb, int stride, int64_t lo, int64_t hi, TextBuf* out) {
  char* base = out->room((size_t)(hi - lo) * 80 + 1300);
  char* p = base;
  for (int64_t i = lo; i < hi; ++i) {
    if ((size_t)(p - base) + 1300 > out->cap) {
      const size_t used = p - base;
      base = out->grow(out->cap * 2 + 1300, used);
      p = base + used;
    }
    for (int a = 0; a < 3; ++a) {
      p = fmt4(p, (double)xyz[i * 3 + a]);
      *p++ = ' ';
    }
    for (int a = 0; a < 3; ++a) {
      p = put_uint(p, rgb[i * stride + a]);
      *p++ = ' ';
    }
    *p++ = '0';
    *p++ = '\n';
  }
  out->len = p - base;
}

// rows [lo, lo + cnt) as text, one chunk per pool thread: rows(a, b, TextBuf*) formats rows [a, b).  Buffers the vector
// already holds are reused.  May throw (no memory, no thread).
template <typename Rows>
void format_parallel(int64_t lo, int64_t cnt, int64_t min_rows_per_chunk, const Rows& rows, std::vector<TextBuf>* chunks) {
  const int64_t n_chunks = std::min<int64_t>(std::max(1u, r3d_host::cpu_budget()), std::max<int64_t>(1, cnt / min_rows_per_chunk));
  chunks->resize((size_t)n_chunks);
  if (n_chunks == 1) {
    rows(lo, lo + cnt, &(*chunks)[0]);
    return;
  }
  const r3d_host::Spread spread;
  std::vector<std::thread> pool;
  std::exception_ptr failed;
  std::atomic<bool> have_failure{false};
  try {
    for (int64_t c = 0; c < n_chunks; ++c) {
      const int64_t a = lo + cnt * c / n_chunks, b = lo + cnt * (c + 1) / n_chunks;
      pool.emplace_back([&, a, b, c]() {
        spread.place((unsigned)c);   // the creator only waits: chunk 0 may have its CPU
        try {
          rows(a, b, &(*chunks)[(size_t)c]);
        } catch (...) {
          if (!have_failure.exchange(true)) failed = std::current_exception();
        }
      });
    }
  } catch (...) {   // no thread to be had: the ones that started must finish before the buffers go away
    for (auto& t : pool) t.join();
    throw;
  }
  for (auto& t : pool) t.join();
  if (have_failure.load()) std::rethrow_exception(failed);
}

// The body of a text file, slab by slab: slab k goes to the file on a helper thread while slab k + 1 is being formatted
// (two sets of chunk buffers).  Returns false on a short write; may throw (no memory, no thread) with the helper joined.
template <typename Rows>
bool write_slabs(FILE* f, int64_t n_rows, int64_t slab, int64_t min_rows_per_chunk, const Rows& rows) {
  std::vector<TextBuf> buf[2];
  std::thread writer;
  bool wrote_ok = true;
  struct Join {
    std::thread& t;
    ~Join() {
      if (t.joinable()) t.join();
    }
  } join{writer};
  int cur = 0;
  for (int64_t lo = 0; lo < n_rows; lo += slab, cur ^= 1) {
    format_parallel(lo, std::min(slab, n_rows - lo), min_rows_per_chunk, rows, &buf[cur]);
    if (writer.joinable()) writer.join();   // the other set is free again once its slab is in the file
    if (!wrote_ok) return false;
    const std::vector<TextBuf>* src = &buf[cur];
    writer = std::thread([src, f, &wrote_ok]() {
      for (const auto& c : *src) wrote_ok = wrote_ok && fwrite(c.data(), 1, c.size(), f) == c.size();
    });
  }
  if (writer.joinable()) writer.join();
  return wrote_ok;
}

template <typename F32, typename F64>
auto by_dtype(int dtype, F32 f32, F64 f64) {
  return [=](int64_t a, int64_t b, TextBuf* out) {
    if (dtype == R3D_F32)
      f32(a, b, out);
    else
      f64(a, b, out);
  };
}

int ply_header(char* head, size_t cap, int64_t n, bool colour) {
  return snprintf(head, cap,
                  "ply\n    format ascii 1.0\n    element vertex %lld\n    property float x\n    property float y\n"
                  "    property float z\n%s    end_header\n    ",
                  (long long)n,
                  colour ? "    property uchar red\n    property uchar green\n    property uchar blue\n    property uchar alpha\n" : "");
}

}  // namespace

extern "C" {

int r3d_format_ply(const void* h_xyz, int dtype, int64_t n_points, char* h_buf, size_t buf_cap,
                   size_t* n_bytes_out) {
  if (n_points < 0 || (n_points > 0 && !h_xyz) || (dtype != R3D_F32 && dtype != R3D_F64) || !n_bytes_out) {
    r3d_set_error("r3d_format_ply: bad argument");
    return R3D_ERR_INVALID;
  }
  char head[256];
  ply_header(head, sizeof(head), n_points, false);
  const std::string header = head;
  std::vector<TextBuf> chunks;
  try {
    format_parallel(0, n_points, 65536,
                    by_dtype(dtype, [=](int64_t a, int64_t b, TextBuf* o) { format_rows(static_cast<const float*>(h_xyz), a, b, o); },
                             [=](int64_t a, int64_t b, TextBuf* o) { format_rows(static_cast<const double*>(h_xyz), a, b, o); }),
                    &chunks);
  } catch (const std::exception&) {
    r3d_set_error("r3d_format_ply: out of host memory");
    return R3D_ERR_NOMEM;
  }
  size_t total = header.size() + 5;  // trailer "\n    "
  for (const auto& c : chunks) total += c.size();
  *n_bytes_out = total;
  if (!h_buf) return R3D_OK;
  if (buf_cap < total) {
    r3d_set_error("r3d_format_ply: buffer of %zu bytes is too small for %zu", buf_cap, total);
    return R3D_ERR_NOMEM;
  }
  char* p = h_buf;
  memcpy(p, header.data(), header.size());
  p += header.size();
  for (const auto& c : chunks) {
    memcpy(p, c.data(), c.size());
    p += c.size();
  }
  memcpy(p, "\n    ", 5);
  return R3D_OK;
}

static int write_ply_colour(const char* path, const void* h_xyz, int dtype, const unsigned char* h_rgb, int stride,
                            int64_t n_points);

int r3d_write_ply_rgb(const char* path, const void* h_xyz, int dtype, const unsigned char* h_rgb, int64_t n_points) {
  return write_ply_colour(path, h_xyz, dtype, h_rgb, 3, n_points);
}

int r3d_write_ply_rgba(const char* path, const void* h_xyz, int dtype, const uint32_t* h_rgba, int64_t n_points) {
  return write_ply_colour(path, h_xyz, dtype, reinterpret_cast<const unsigned char*>(h_rgba), 4, n_points);
}

static int write_ply_colour(const char* path, const void* h_xyz, int dtype, const unsigned char* h_rgb, int stride,
                            int64_t n_points) {
  if (!path || n_points < 0 || (n_points > 0 && (!h_xyz || !h_rgb)) || (dtype != R3D_F32 && dtype != R3D_F64)) {
    r3d_set_error("r3d_write_ply_rgb: bad argument");
    return R3D_ERR_INVALID;
  }
  FILE* f = fopen(path, "wb");
  if (!f) {
    r3d_set_error("r3d_write_ply_rgb: cannot open '%s' for writing", path);
    return R3D_ERR_INVALID;
  }
  bool ok = true;
  try {
    char head[400];
    const int hn = ply_header(head, sizeof(head), n_points, true);
    ok = fwrite(head, 1, (size_t)hn, f) == (size_t)hn;
    ok = ok && write_slabs(f, n_points, (int64_t)4 << 20, 65536,
                           by_dtype(dtype, [=](int64_t a, int64_t b_, TextBuf* o) { format_rows_rgb(static_cast<const float*>(h_xyz), h_rgb, stride, a, b_, o); },
                                    [=](int64_t a, int64_t b_, TextBuf* o) { format_rows_rgb(static_cast<const double*>(h_xyz), h_rgb, stride, a, b_, o); }));
    ok = ok && fwrite("\n    ", 1, 5, f) == 5;
  } catch (const std::exception&) {   // bad_alloc, or no thread to be had
    fclose(f);
    r3d_set_error("r3d_write_ply_rgb: out of host memory");
    return R3D_ERR_NOMEM;
  }
  if (fclose(f) != 0) ok = false;
  if (!ok) {
    r3d_set_error("r3d_write_ply_rgb: short write to '%s'", path);
    return R3D_ERR_INVALID;
  }
  return R3D_OK;
}

int r3d_write_ply(const char* path, const void* h_xyz, int dtype, int64_t n_points) {
  if (!path || n_points < 0 || (n_points > 0 && !h_xyz) || (dtype != R3D_F32 && dtype != R3D_F64)) {
    r3d_set_error("r3d_write_ply: bad argument");
    return R3D_ERR_INVALID;
  }
  FILE* f = fopen(path, "wb");
  if (!f) {
    r3d_set_error("r3d_write_ply: cannot open '%s' for writing", path);
    return R3D_ERR_INVALID;
  }
  // formatted and written in slabs of 8 M points so the text never needs more than ~0.5 GB of host memory
  bool ok = true;
  try {
    char head[256];
    const int hn = ply_header(head, sizeof(head), n_points, false);
    ok = fwrite(head, 1, (size_t)hn, f) == (size_t)hn;
    ok = ok && write_slabs(f, n_points, (int64_t)8 << 20, 65536,
                           by_dtype(dtype, [=](int64_t a, int64_t b_, TextBuf* o) { format_rows(static_cast<const float*>(h_xyz), a, b_, o); },
                                    [=](int64_t a, int64_t b_, TextBuf* o) { format_rows(static_cast<const double*>(h_xyz), a, b_, o); }));
    ok = ok && fwrite("\n    ", 1, 5, f) == 5;
  } catch (const std::exception&) {  // bad_alloc, or no thread to be had
    fclose(f);
    r3d_set_error("r3d_write_ply: out of host memory");
    return R3D_ERR_NOMEM;
  }
  if (fclose(f) != 0) ok = false;
  if (!ok) {
    r3d_set_error("r3d_write_ply: short write to '%s'", path);
    return R3D_ERR_INVALID;
  }
  return R3D_OK;
}

// f1's optional binary flag: the same vertices as r3d_write_ply in a STANDARD little-endian binary PLY (the header without the
// reference template's indents, which binary readers do not accept; 12 bytes per vertex instead of ~26 of text).  Not the
// reference's bytes -- an opt-in for users whose next tool reads PLY, not this file's text.
int r3d_write_ply_binary(const char* path, const void* h_xyz, int dtype, int64_t n_points) {
  if (!path || n_points < 0 || (n_points > 0 && !h_xyz) || (dtype != R3D_F32 && dtype != R3D_F64)) {
    r3d_set_error("r3d_write_ply_binary: bad argument");
    return R3D_ERR_INVALID;
  }
  FILE* f = fopen(path, "wb");
  if (!f) {
    r3d_set_error("r3d_write_ply_binary: cannot open '%s' for writing", path);
    return R3D_ERR_INVALID;
  }
  char head[256];
  const int hn = snprintf(head, sizeof(head),
                          "ply\nformat binary_little_endian 1.0\nelement vertex %lld\nproperty float x\nproperty float y\nproperty float z\nend_header\n",
                          (long long)n_points);
  bool ok = fwrite(head, 1, (size_t)hn, f) == (size_t)hn;
  if (dtype == R3D_F32) {
    ok = ok && (n_points == 0 || fwrite(h_xyz, 12, (size_t)n_points, f) == (size_t)n_points);
  } else {
    std::vector<float> block;
    try {
      block.resize((size_t)std::min<int64_t>(n_points, (int64_t)1 << 20) * 3);
    } catch (const std::exception&) {
      fclose(f);
      r3d_set_error("r3d_write_ply_binary: out of host memory");
      return R3D_ERR_NOMEM;
    }
    const double* src = static_cast<const double*>(h_xyz);
    for (int64_t lo = 0; ok && lo < n_points; lo += (int64_t)1 << 20) {
      const int64_t m = std::min<int64_t>((int64_t)1 << 20, n_points - lo);
      for (int64_t k = 0; k < m * 3; ++k) block[(size_t)k] = (float)src[lo * 3 + k];
      ok = fwrite(block.data(), 12, (size_t)m, f) == (size_t)m;
    }
  }
  if (fclose(f) != 0) ok = false;
  if (!ok) {
    r3d_set_error("r3d_write_ply_binary: short write to '%s'", path);
    return R3D_ERR_INVALID;
  }
  return R3D_OK;
}

static auto txt_rows_of(const void* h_xyz, int dtype, const void* h_z_raw, int z_raw_dtype) {
  return by_dtype(dtype, [=](int64_t a, int64_t b, TextBuf* o) { txt_rows(static_cast<const float*>(h_xyz), h_z_raw, z_raw_dtype, a, b, o); },
                  [=](int64_t a, int64_t b, TextBuf* o) { txt_rows(static_cast<const double*>(h_xyz), h_z_raw, z_raw_dtype, a, b, o); });
}

int r3d_format_xyz_txt(const void* h_xyz, int dtype, int64_t n_points, const void* h_z_raw, int z_raw_dtype,
                       char* h_buf, size_t buf_cap, size_t* n_bytes_out) {
  if (n_points < 0 || (n_points > 0 && !h_xyz) || (dtype != R3D_F32 && dtype != R3D_F64) || !n_bytes_out ||
      (h_z_raw && z_raw_dtype != R3D_DEPTH_U8 && z_raw_dtype != R3D_DEPTH_U16)) {
    r3d_set_error("r3d_format_xyz_txt: bad argument");
    return R3D_ERR_INVALID;
  }
  std::vector<TextBuf> chunks;
  try {
    format_parallel(0, n_points, 32768, txt_rows_of(h_xyz, dtype, h_z_raw, z_raw_dtype), &chunks);
  } catch (const std::exception&) {
    r3d_set_error("r3d_format_xyz_txt: out of host memory");
    return R3D_ERR_NOMEM;
  }
  size_t total = 0;
  for (const auto& c : chunks) total += c.size();
  *n_bytes_out = total;
  if (!h_buf) return R3D_OK;
  if (buf_cap < total) {
    r3d_set_error("r3d_format_xyz_txt: buffer of %zu bytes is too small for %zu", buf_cap, total);
    return R3D_ERR_NOMEM;
  }
  char* p = h_buf;
  for (const auto& c : chunks) {
    memcpy(p, c.data(), c.size());
    p += c.size();
  }
  return R3D_OK;
}

int r3d_write_xyz_txt(const char* path, const void* h_xyz, int dtype, int64_t n_points, const void* h_z_raw,
                      int z_raw_dtype, int append) {
  if (!path || n_points < 0 || (n_points > 0 && !h_xyz) || (dtype != R3D_F32 && dtype != R3D_F64) ||
      (h_z_raw && z_raw_dtype != R3D_DEPTH_U8 && z_raw_dtype != R3D_DEPTH_U16)) {
    r3d_set_error("r3d_write_xyz_txt: bad argument");
    return R3D_ERR_INVALID;
  }
  FILE* f = fopen(path, append ? "ab" : "wb");
  if (!f) {
    r3d_set_error("r3d_write_xyz_txt: cannot open '%s' for writing", path);
    return R3D_ERR_INVALID;
  }
  bool ok = true;
  try {
    ok = write_slabs(f, n_points, (int64_t)4 << 20, 32768, txt_rows_of(h_xyz, dtype, h_z_raw, z_raw_dtype));
  } catch (const std::exception&) {
    fclose(f);
    r3d_set_error("r3d_write_xyz_txt: out of host memory");
    return R3D_ERR_NOMEM;
  }
  if (fclose(f) != 0) ok = false;
  if (!ok) {
    r3d_set_error("r3d_write_xyz_txt: short write to '%s'", path);
    return R3D_ERR_INVALID;
  }
  return R3D_OK;
}

// One txt file per frame (the ./point/<stem>.txt of camera_to_world.py:80-81, one per pose line): file k holds points
// [k * points_per_file, (k + 1) * points_per_file).  The files are the unit of parallelism -- cpu_budget() workers, each
// formatting whole files block by block into one buffer of its own and writing them -- instead of one pool of threads per
// file.  Same bytes as n_files calls of r3d_write_xyz_txt.
int r3d_write_xyz_txt_batch(const char* const* paths, int n_files, const void* h_xyz, int dtype, int64_t points_per_file,
                            const void* h_z_raw, int z_raw_dtype) {
  if (n_files < 0 || points_per_file < 0 || (n_files > 0 && !paths) || (n_files > 0 && points_per_file > 0 && !h_xyz) ||
      (dtype != R3D_F32 && dtype != R3D_F64) || (h_z_raw && z_raw_dtype != R3D_DEPTH_U8 && z_raw_dtype != R3D_DEPTH_U16)) {
    r3d_set_error("r3d_write_xyz_txt_batch: bad argument");
    return R3D_ERR_INVALID;
  }
  for (int k = 0; k < n_files; ++k)
    if (!paths[k]) {
      r3d_set_error("r3d_write_xyz_txt_batch: path %d is NULL", k);
      return R3D_ERR_INVALID;
    }
  const unsigned budget = r3d_host::cpu_budget();
  if ((unsigned)n_files * 2 <= budget) {   // few files: the threads go inside each file
    for (int k = 0; k < n_files; ++k) {
      const size_t esz = dtype == R3D_F32 ? 4 : 8, zsz = z_raw_dtype == R3D_DEPTH_U16 ? 2 : 1;
      const int rc = r3d_write_xyz_txt(paths[k], static_cast<const char*>(h_xyz) + (size_t)k * points_per_file * 3 * esz, dtype, points_per_file,
                                       h_z_raw ? static_cast<const char*>(h_z_raw) + (size_t)k * points_per_file * zsz : nullptr, z_raw_dtype, 0);
      if (rc) return rc;
    }
    return R3D_OK;
  }
  try {
    return r3d_host::run_batch(n_files, "txt write failed", [&](int k, std::string* msg) -> int {
      thread_local TextBuf text;   // one per worker thread; gone with the thread
      FILE* f = fopen(paths[k], "wb");
      if (!f) {
        *msg = std::string("r3d_write_xyz_txt_batch: cannot open '") + paths[k] + "' for writing";
        return R3D_ERR_INVALID;
      }
      setvbuf(f, nullptr, _IONBF, 0);   // every fwrite below is a whole block
      bool ok = true;
      const int64_t lo = (int64_t)k * points_per_file, hi = lo + points_per_file, block = 65536;
      try {
        for (int64_t b = lo; ok && b < hi; b += block) {
          const int64_t e = std::min(hi, b + block);
          if (dtype == R3D_F32)
            txt_rows(static_cast<const float*>(h_xyz), h_z_raw, z_raw_dtype, b, e, &text);
          else
            txt_rows(static_cast<const double*>(h_xyz), h_z_raw, z_raw_dtype, b, e, &text);
          ok = fwrite(text.data(), 1, text.size(), f) == text.size();
        }
      } catch (const std::bad_alloc&) {
        fclose(f);
        *msg = "r3d_write_xyz_txt_batch: out of host memory";
        return R3D_ERR_NOMEM;
      }
      if (fclose(f) != 0) ok = false;
      if (!ok) {
        *msg = std::string("r3d_write_xyz_txt_batch: short write to '") + paths[k] + "'";
        return R3D_ERR_INVALID;
      }
      return R3D_OK;
    });
  } catch (const std::exception&) {   // no thread to be had
    r3d_set_error("r3d_write_xyz_txt_batch: out of host memory");
    return R3D_ERR_NOMEM;
  }
}

// ---- the way back: "x,y,z[,...]\n" lines (camera / world txt; read by get_pointdata c2w:92-98 and local_world
// icp:74-80 as the first three comma-separated fields) or "x y z [...]\n" rows (a PLY body) -> [n][3] fp64.
// Threads split the text at line boundaries, count their rows, then parse with std::from_chars (correctly rounded,
// = Python's float()).  Anything from_chars does not take the way float() does (underscores, "Infinity", hex ...)
// reports its line so that the Python host can re-parse the file its own way: semantics stay Python's, speed is native.
namespace {

struct ParseSpan {
  const char* lo;
  const char* hi;
  int64_t rows = 0;        // non-blank lines
  int64_t first_line = 0;  // 1-based number of the span's first line
  int64_t lines = 0;       // newline-terminated (or final) lines, blank ones included
};

inline bool blank_line(const char* a, const char* b) {
  for (; a < b; ++a)
    if (*a != ' ' && *a != '\t' && *a != '\r') return false;
  return true;
}

// Fast path for [-]digits[.digits][e[+-]digits] with at most 19 significant digits and a decimal exponent within +-27:
// the digits are an exact 64-bit integer w, 10^|q| is exact in x87 extended precision (5^27 < 2^63), so w * 10^q or
// w / 10^q is ONE correctly rounded operation to a 64-bit significand.  Rounding that once more to double is wrong only
// if the 64-bit result sits exactly on a midpoint between two doubles (low 11 bits == 0x400) while the true value does
// not; those (and their neighbours, for good measure) go to the exact path.  Result range 1e-27 .. 1.9e46: always normal.
inline bool fast_double(const char* a, const char* b, double* out) {
#if defined(__x86_64__) && __LDBL_MANT_DIG__ == 64
  static const long double p10[28] = {1e0L,  1e1L,  1e2L,  1e3L,  1e4L,  1e5L,  1e6L,  1e7L,  1e8L,  1e9L,
                                      1e10L, 1e11L, 1e12L, 1e13L, 1e14L, 1e15L, 1e16L, 1e17L, 1e18L, 1e19L,
                                      1e20L, 1e21L, 1e22L, 1e23L, 1e24L, 1e25L, 1e26L, 1e27L};
  const char* p = a;
  bool neg = false;
  if (p < b && *p == '-') {
    neg = true;
    ++p;
  }
  uint64_t w = 0;
  int nd = 0;
  int64_t q = 0;
  bool any = false;
  while (p < b && (unsigned)(*p - '0') < 10u) {
    any = true;
    const unsigned d = (unsigned)(*p - '0');
    if (nd || d) {
      if (nd == 19) return false;
      w = w * 10 + d;
      ++nd;
    }
    ++p;
  }
  if (p < b && *p == '.') {
    ++p;
    while (p < b && (unsigned)(*p - '0') < 10u) {
      any = true;
      const unsigned d = (unsigned)(*p - '0');
      if (nd || d) {
        if (nd == 19) return false;
        w = w * 10 + d;
        ++nd;
      }
      --q;
      ++p;
    }
  }
  if (!any) return false;
  if (p < b && (*p == 'e' || *p == 'E')) {
    ++p;
    bool eneg = false;
    if (p < b && (*p == '-' || *p == '+')) {
      eneg = *p == '-';
      ++p;
    }
    if (p >= b) return false;
    int64_t e = 0;
    int digits = 0;
    while (p < b && (unsigned)(*p - '0') < 10u) {
      if (++digits > 5) return false;
      e = e * 10 + (*p - '0');
      ++p;
    }
    q += eneg ? -e : e;
  }
  if (p != b) return false;
  if (w == 0) {
    *out = neg ? -0.0 : 0.0;
    return true;
  }
  if (q < -27 || q > 27) return false;
  long double r = (long double)w;
  r = q >= 0 ? r * p10[q] : r / p10[-q];
  uint64_t mant;
  memcpy(&mant, &r, sizeof(mant));  // x87 extended, little endian: the 64-bit significand comes first
  const unsigned low = (unsigned)(mant & 0x7ffu);
  if (low >= 0x3ffu && low <= 0x401u) return false;
  const double d = (double)r;
  *out = neg ? -d : d;
  return true;
#else
  (void)a;
  (void)b;
  (void)out;
  return false;
#endif
}

// one field [a, b) -> double like float(): surrounding blanks allowed, optional '+', nan / inf in any case
inline bool parse_field(const char* a, const char* b, double* out) {
  while (a < b && (*a == ' ' || *a == '\t' || *a == '\r')) ++a;
  while (b > a && (b[-1] == ' ' || b[-1] == '\t' || b[-1] == '\r')) --b;
  if (a < b && *a == '+') {
    ++a;
    if (a < b && (*a == '-' || *a == '+')) return false;
  }
  if (a >= b) return false;
  if (fast_double(a, b, out)) return true;
  const std::from_chars_result r = std::from_chars(a, b, *out, std::chars_format::general);
  if (r.ec == std::errc::result_out_of_range) {  // float() gives +-inf on overflow and +-0.0 (or a subnormal) on underflow
    char tmp[64];
    const size_t n = (size_t)(b - a);
    if (n >= sizeof(tmp)) return false;
    memcpy(tmp, a, n);
    tmp[n] = 0;
    char* end = nullptr;
    *out = strtod(tmp, &end);
    return end == tmp + n;
  }
  return r.ec == std::errc() && r.ptr == b;
}

// rows of one span into out; returns 0 or the 1-based line number that did not parse
int64_t parse_span(const ParseSpan& sp, char sep, double* out) {
  const char* p = sp.lo;
  int64_t line = sp.first_line;
  while (p < sp.hi) {
    const char* e = static_cast<const char*>(memchr(p, '\n', (size_t)(sp.hi - p)));
    if (!e) e = sp.hi;
    if (!blank_line(p, e)) {
      const char* f = p;
      for (int k = 0; k < 3; ++k) {
        const char* g;
        if (sep == ',') {
          g = static_cast<const char*>(memchr(f, ',', (size_t)(e - f)));
          if (!g) g = e;
          if (k < 2 && g == e) return line;  // fewer than three fields
        } else {                             // any run of blanks separates
          while (f < e && (*f == ' ' || *f == '\t' || *f == '\r')) ++f;
          g = f;
          while (g < e && *g != ' ' && *g != '\t' && *g != '\r') ++g;
          if (f == g) return line;
        }
        if (!parse_field(f, g, out + k)) return line;
        f = g < e ? g + 1 : e;
      }
      out += 3;
    }
    p = e < sp.hi ? e + 1 : sp.hi;
    ++line;
  }
  return 0;
}

}  // namespace

int r3d_parse_xyz_text(const char* h_text, size_t n_bytes, int separator, double* h_xyz_out, int64_t cap_points,
                       int64_t* n_points_out, int64_t* bad_line_out) {
  if ((n_bytes > 0 && !h_text) || !n_points_out || (separator != ',' && separator != ' ') || cap_points < 0) {
    r3d_set_error("r3d_parse_xyz_text: bad argument");
    return R3D_ERR_INVALID;
  }
  *n_points_out = 0;
  if (bad_line_out) *bad_line_out = 0;
  if (n_bytes == 0) return R3D_OK;
  try {
    unsigned hw = r3d_host::cpu_budget();
    unsigned n_threads = std::max(1u, std::min(hw == 0 ? 1u : hw, 16u));
    if (n_bytes < ((size_t)1 << 20)) n_threads = 1;
    // spans end just after a newline
    std::vector<ParseSpan> spans;
    const char* end = h_text + n_bytes;
    const char* lo = h_text;
    for (unsigned t = 0; t < n_threads && lo < end; ++t) {
      const char* hi = t + 1 == n_threads ? end : h_text + (n_bytes / n_threads) * (t + 1);
      if (hi < lo) hi = lo;
      if (hi < end) {
        const char* nl = static_cast<const char*>(memchr(hi, '\n', (size_t)(end - hi)));
        hi = nl ? nl + 1 : end;
      }
      ParseSpan sp;
      sp.lo = lo;
      sp.hi = hi;
      spans.push_back(sp);
      lo = hi;
    }
    const r3d_host::Spread spread;
    auto count = [](ParseSpan* sp) {
      const char* p = sp->lo;
      while (p < sp->hi) {
        const char* e = static_cast<const char*>(memchr(p, '\n', (size_t)(sp->hi - p)));
        if (!e) e = sp->hi;
        if (!blank_line(p, e)) ++sp->rows;
        ++sp->lines;
        p = e < sp->hi ? e + 1 : sp->hi;
      }
    };
    {
      std::vector<std::thread> pool;
      for (size_t t = 1; t < spans.size(); ++t)
        pool.emplace_back([&, t]() {
          spread.place((unsigned)t);
          count(&spans[t]);
        });
      count(&spans[0]);
      for (auto& th : pool) th.join();
    }
    int64_t total = 0, line = 1;
    std::vector<int64_t> row0(spans.size());
    for (size_t t = 0; t < spans.size(); ++t) {
      row0[t] = total;
      total += spans[t].rows;
      spans[t].first_line = line;
      line += spans[t].lines;
    }
    *n_points_out = total;
    if (!h_xyz_out) return R3D_OK;  // count only
    if (cap_points < total) {
      r3d_set_error("r3d_parse_xyz_text: buffer holds %lld points, text has %lld", (long long)cap_points, (long long)total);
      return R3D_ERR_NOMEM;
    }
    std::vector<int64_t> bad(spans.size(), 0);
    {
      std::vector<std::thread> pool;
      for (size_t t = 1; t < spans.size(); ++t)
        pool.emplace_back([&, t]() {
          spread.place((unsigned)t);
          bad[t] = parse_span(spans[t], (char)separator, h_xyz_out + 3 * row0[t]);
        });
      bad[0] = parse_span(spans[0], (char)separator, h_xyz_out);
      for (auto& th : pool) th.join();
    }
    for (size_t t = 0; t < spans.size(); ++t)
      if (bad[t]) {
        if (bad_line_out) *bad_line_out = bad[t];
        r3d_set_error("r3d_parse_xyz_text: line %lld is not 'x%cy%cz[...]'", (long long)bad[t], (char)separator, (char)separator);
        return R3D_ERR_INVALID;
      }
    return R3D_OK;
  } catch (const std::exception&) {
    r3d_set_error("r3d_parse_xyz_text: out of host memory");
    return R3D_ERR_NOMEM;
  }
}

}  // extern "C"
