// f3 ingestion: JPEG -> the 8-bit grey raster `cv.imread(path, IMREAD_GRAYSCALE)` returns, on host threads.
//
// AirSim stores depth as 3-channel JPG (airsim/main.cpp:1369-1392, cv::imwrite); camera_to_world.py:160 reads depth files with
// IMREAD_GRAYSCALE.  For a JPEG OpenCV does not decode colour and convert: its reader (grfmt_jpeg.cpp) asks libjpeg for
// grey output (out_color_space = JCS_GRAYSCALE), and libjpeg then decodes the LUMA component alone -- entropy-decodes every
// block, dequantises and inverse-transforms the Y blocks with its default "islow" integer IDCT (jidctint.c, bit-identical
// in libjpeg-turbo's SIMD), adds 128, clamps.  Chroma never enters, so there is no upsampling and no colour matrix to get
// wrong: the whole path is integer arithmetic with one published algorithm, restated here.
//   * baseline / extended-sequential Huffman JPEG (SOF0 / SOF1), 8-bit, one interleaved scan (what cv::imwrite and PIL write),
//     1 component (grey) or 3 (YCbCr; luma at full resolution: 4:4:4, 4:2:2, 4:2:0 ...), restart intervals;
//   * progressive, arithmetic-coded, 12-bit, CMYK / Adobe-RGB files, multi-scan sequential files: R3D_ERR_UNSUPPORTED (the
//     Python host names the PIL fallback).
// PINNED against libjpeg-turbo itself through PIL's draft('L') decode (the same library request OpenCV makes):
// tests/test_host_logic.py compares rasters byte for byte over sizes, qualities, subsamplings and restart intervals.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "r3d.h"
#include "r3d_hostpool.h"

namespace {

const unsigned char kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  // canonical code, JPEG Annex F.2.2.3: per length the smallest code, the largest, and where its symbols start
  int32_t mincode[17], maxcode[18], valptr[17];
  unsigned char vals[256];
  // 9-bit lookahead: (length << 8) | symbol, 0 = longer than 9 bits
  uint16_t look[512];
  // AC tables: a code AND the magnitude bits behind it in one step when both fit in kFastBits --
  // (coefficient << 16) | (zero run << 8) | bits consumed; 0 = take the long way (EOB, ZRL, long codes, large magnitudes).
  // Most coefficients of a photograph are small and their codes short: one table read per coefficient instead of a code
  // lookup, a bit-field read and the sign extension (entropy decoding is ~90 % of a decode; the IDCT ~7 %).
  int32_t fast[1 << 10];
};
constexpr int kFastBits = 10;

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
};

struct Jpeg {
  int width = 0, height = 0, n_comp = 0;
  Component comp[4];
  uint16_t quant[4][64];   // in zigzag order, as stored
  bool have_quant[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  int restart_interval = 0;
  bool progressive = false, arithmetic = false, adobe = false;
  int adobe_transform = -1, precision = 8;
  size_t scan_begin = 0;   // first byte of the entropy-coded segment
  bool have_sof = false, have_sos = false, scan_is_whole_image = false;
};

int fail(int code, const char* path, const std::string& why, std::string* msg) {
  *msg = std::string("'") + path + "': " + why;
  return code;
}

// false: the code lengths do not describe a prefix code (more codes of a length than that length has left -- libjpeg's
// JERR_BAD_HUFF_TABLE)
bool build_huff(Huff* h, const unsigned char counts[16], const unsigned char* symbols, int n_symbols) {
  memcpy(h->vals, symbols, (size_t)n_symbols);
  int code = 0, k = 0;
  for (int len = 1; len <= 16; ++len) {
    h->valptr[len] = k;
    h->mincode[len] = code;
    code += counts[len - 1];
    if (code > (1 << len)) return false;
    k += counts[len - 1];
    h->maxcode[len] = counts[len - 1] ? code - 1 : -1;
    code <<= 1;
  }
  h->present = true;
  h->maxcode[17] = 0x7fffffff;
  memset(h->look, 0, sizeof(h->look));
  code = 0;
  k = 0;
  for (int len = 1; len <= 9; ++len) {
    for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
      const int first = code << (9 - len), n = 1 << (9 - len);
      for (int j = 0; j < n; ++j) h->look[first + j] = (uint16_t)((len << 8) | symbols[k]);
    }
    code <<= 1;
  }
  for (int i = 0; i < (1 << kFastBits); ++i) {
    h->fast[i] = 0;
    const uint16_t e = h->look[i >> (kFastBits - 9)];
    const int len = e >> 8, run = (e >> 4) & 15, size = e & 15;
    if (!e || size == 0 || len + size > kFastBits) continue;
    const int mag = (i >> (kFastBits - len - size)) & ((1 << size) - 1);
    const int value = mag < (1 << (size - 1)) ? mag - (1 << size) + 1 : mag;   // extend()
    h->fast[i] = (int32_t)(((uint32_t)value << 16) | ((uint32_t)run << 8) | (uint32_t)(len + size));
  }
  return true;
}

// the marker segments up to (and including) the first SOS header
int parse_headers(const char* path, const std::vector<unsigned char>& f, Jpeg* j, std::string* msg) {
  if (f.size() < 4 || f[0] != 0xff || f[1] != 0xd8) return fail(R3D_ERR_INVALID, path, "not a JPEG file", msg);
  size_t pos = 2;
  while (pos + 4 <= f.size()) {
    if (f[pos] != 0xff) return fail(R3D_ERR_INVALID, path, "marker expected", msg);
    while (pos < f.size() && f[pos] == 0xff) ++pos;   // fill bytes
    if (pos >= f.size()) break;
    const int m = f[pos++];
    if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;   // stand-alone markers
    if (m == 0xd9) break;
    if (pos + 2 > f.size()) break;
    const size_t len = ((size_t)f[pos] << 8) | f[pos + 1];
    if (len < 2 || pos + len > f.size()) return fail(R3D_ERR_INVALID, path, "truncated marker segment", msg);
    const unsigned char* d = &f[pos + 2];
    const size_t n = len - 2;
    if (m == 0xdb) {   // DQT
      size_t at = 0;
      while (at < n) {
        const int pq = d[at] >> 4, tq = d[at] & 15;
        ++at;
        if (tq > 3 || at + (pq ? 128u : 64u) > n) return fail(R3D_ERR_INVALID, path, "bad quantisation table", msg);
        for (int k = 0; k < 64; ++k) {
          j->quant[tq][k] = pq ? (uint16_t)((d[at] << 8) | d[at + 1]) : d[at];
          at += pq ? 2 : 1;
        }
        j->have_quant[tq] = true;
      }
    } else if (m == 0xc4) {   // DHT
      size_t at = 0;
      while (at + 17 <= n) {
        const int tc = d[at] >> 4, th = d[at] & 15;
        int total = 0;
        for (int k = 0; k < 16; ++k) total += d[at + 1 + k];
        if (tc > 1 || th > 3 || total > 256 || at + 17 + (size_t)total > n) return fail(R3D_ERR_INVALID, path, "bad Huffman table", msg);
        if (!build_huff(tc ? &j->ac[th] : &j->dc[th], &d[at + 1], &d[at + 17], total)) return fail(R3D_ERR_INVALID, path, "bad Huffman table", msg);
        at += 17 + (size_t)total;
      }
    } else if (m == 0xc0 || m == 0xc1 || m == 0xc2 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {   // SOFn
      if (m == 0xc2 || m == 0xc6 || m == 0xca || m == 0xce) j->progressive = true;
      if (m >= 0xc9) j->arithmetic = true;
      if (m != 0xc0 && m != 0xc1 && m != 0xc2) return fail(R3D_ERR_UNSUPPORTED, path, "lossless / hierarchical / arithmetic-coded JPEG", msg);
      if (n < 6) return fail(R3D_ERR_INVALID, path, "bad frame header", msg);
      j->precision = d[0];
      j->height = (d[1] << 8) | d[2];
      j->width = (d[3] << 8) | d[4];
      j->n_comp = d[5];
      if (j->n_comp < 1 || j->n_comp > 4 || n < 6 + 3 * (size_t)j->n_comp) return fail(R3D_ERR_INVALID, path, "bad frame header", msg);
      for (int c = 0; c < j->n_comp; ++c) {
        j->comp[c].id = d[6 + 3 * c];
        j->comp[c].h = d[7 + 3 * c] >> 4;
        j->comp[c].v = d[7 + 3 * c] & 15;
        j->comp[c].tq = d[8 + 3 * c];
        if (j->comp[c].h < 1 || j->comp[c].h > 4 || j->comp[c].v < 1 || j->comp[c].v > 4 || j->comp[c].tq > 3)
          return fail(R3D_ERR_INVALID, path, "bad sampling factors", msg);
      }
      j->have_sof = true;
    } else if (m == 0xdd) {   // DRI
      if (n >= 2) j->restart_interval = (d[0] << 8) | d[1];
    } else if (m == 0xee) {   // APP14 "Adobe"
      if (n >= 12 && !memcmp(d, "Adobe", 5)) {
        j->adobe = true;
        j->adobe_transform = d[11];
      }
    } else if (m == 0xda) {   // SOS
      if (!j->have_sof) return fail(R3D_ERR_INVALID, path, "scan before frame header", msg);
      const int ns = n ? d[0] : 0;
      if (ns < 1 || ns > 4 || n < 4 + 2 * (size_t)ns) return fail(R3D_ERR_INVALID, path, "bad scan header", msg);
      j->scan_is_whole_image = ns == j->n_comp;
      for (int s = 0; s < ns; ++s) {
        const int cid = d[1 + 2 * s];
        bool found = false;
        for (int c = 0; c < j->n_comp; ++c)
          if (j->comp[c].id == cid) {
            // the components of the one scan must come in frame order (libjpeg requires it too)
            if (j->scan_is_whole_image && c != s) return fail(R3D_ERR_UNSUPPORTED, path, "scan components out of frame order", msg);
            j->comp[c].td = d[2 + 2 * s] >> 4;
            j->comp[c].ta = d[2 + 2 * s] & 15;
            if (j->comp[c].td > 3 || j->comp[c].ta > 3) return fail(R3D_ERR_INVALID, path, "scan names a Huffman table beyond 3", msg);
            found = true;
          }
        if (!found) return fail(R3D_ERR_INVALID, path, "scan names an unknown component", msg);
      }
      j->scan_begin = pos + len;
      j->have_sos = true;
      return R3D_OK;
    }
    pos += len;
  }
  return fail(R3D_ERR_INVALID, path, "no image data", msg);
}

struct Bits {
  const unsigned char* p;
  const unsigned char* end;
  uint64_t buf = 0;   // the coming bits, first bit in bit 63.  Below the n valid ones there may be a PREVIEW of the next byte's
                      // leading bits (left by the 8-byte refill); the same bits are or-ed over it by the next refill, and the
                      // byte-wise path clears it first
  int n = 0;          // valid bits
  int marker = 0;     // a marker met in the data (its second byte); nothing is read past it
  void fill() {
    if (!marker && end - p >= 8) {   // eight bytes without an 0xff among them: no stuffing, no marker -- take what fits at once
      uint64_t w;
      memcpy(&w, p, 8);
      const uint64_t x = ~w;
      if (((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) == 0) {
        buf |= __builtin_bswap64(w) >> n;   // n < 64 here: fill() is called with fewer than 16 bits left
        p += (63 - n) >> 3;                 // the whole bytes that fitted
        n |= 56;
        return;
      }
    }
    buf = n ? (buf & (~0ull << (64 - n))) : 0;
    while (n <= 56) {
      int byte = 0;
      if (!marker && p < end) {
        byte = *p;
        if (byte == 0xff) {
          if (p + 1 < end && p[1] == 0x00) {
            p += 2;   // stuffed zero
          } else {
            // skip fill bytes, remember the marker, feed zeros from here on (libjpeg does the same)
            const unsigned char* q = p + 1;
            while (q < end && *q == 0xff) ++q;
            marker = q < end ? *q : 0xd9;
            p = q < end ? q + 1 : end;
            byte = 0;
          }
        } else {
          ++p;
        }
      }
      buf |= (uint64_t)byte << (56 - n);
      n += 8;
    }
  }
  inline int peek(int k) { return (int)(buf >> (64 - k)); }   // 1 <= k <= n
  inline void drop(int k) {
    buf <<= k;
    n -= k;
  }
  inline int get(int k) {
    if (k == 0) return 0;
    if (n < k) fill();
    const int v = peek(k);
    drop(k);
    return v;
  }
};

inline int decode_symbol(Bits& b, const Huff& h) {
  if (b.n < 16) b.fill();
  const uint16_t e = h.look[b.peek(9)];
  if (e) {
    b.drop(e >> 8);
    return e & 0xff;
  }
  int code = b.peek(9), len = 9;
  b.drop(9);
  while (len <= 16 && code > h.maxcode[len]) {   // maxcode is -1 where a length has no codes: the loop moves on
    if (b.n < 1) b.fill();
    code = (code << 1) | b.get(1);
    ++len;
  }
  if (len > 16) return -1;
  return h.vals[h.valptr[len] + code - h.mincode[len]];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// jidctint.c (IJG "islow", also libjpeg-turbo's reference for its SIMD): 13-bit constants, two passes, the first kept at 2
// extra bits.  DESCALE(x, n) = (x + 2^(n-1)) >> n, arithmetic.
constexpr int kConstBits = 13, kPass1Bits = 2;
typedef int64_t wide;   // libjpeg's JLONG is `long`: 64 bits here, so hostile coefficients wrap nowhere (and UBSan stays quiet)
constexpr wide F0_298631336 = 2446, F0_390180644 = 3196, F0_541196100 = 4433, F0_765366865 = 6270, F0_899976223 = 7373,
               F1_175875602 = 9633, F1_501321110 = 12299, F1_847759065 = 15137, F1_961570560 = 16069, F2_053119869 = 16819,
               F2_562915447 = 20995, F3_072711026 = 25172;
inline wide descale(wide x, int n) { return (x + ((wide)1 << (n - 1))) >> n; }

unsigned char g_range_limit[1024];   // index = value & 1023 (the value before the +128): libjpeg's table, wrap included
struct RangeInit {
  RangeInit() {
    for (int i = 0; i < 1024; ++i) g_range_limit[i] = (unsigned char)(i < 128 ? 128 + i : i < 512 ? 255 : i < 896 ? 0 : i - 896);
  }
} g_range_init;

// one 1-D pass over eight values; the two passes differ in where the values come from and how the results are scaled
struct Idct8 {
  wide o[8];
  inline void run(wide c0, wide c1, wide c2, wide c3, wide c4, wide c5, wide c6, wide c7) {
    wide z2 = c2, z3 = c6;
    wide z1 = (z2 + z3) * F0_541196100;
    wide tmp2 = z1 + z3 * (-F1_847759065);
    wide tmp3 = z1 + z2 * F0_765366865;
    wide tmp0 = (c0 + c4) * ((wide)1 << kConstBits);
    wide tmp1 = (c0 - c4) * ((wide)1 << kConstBits);
    const wide tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = c7;
    tmp1 = c5;
    tmp2 = c3;
    tmp3 = c1;
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    wide z4 = tmp1 + tmp3;
    const wide z5 = (z3 + z4) * F1_175875602;
    tmp0 *= F0_298631336;
    tmp1 *= F2_053119869;
    tmp2 *= F3_072711026;
    tmp3 *= F1_501321110;
    z1 *= -F0_899976223;
    z2 *= -F2_562915447;
    z3 *= -F1_961570560;
    z4 *= -F0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3;
    o[7] = tmp10 - tmp3;
    o[1] = tmp11 + tmp2;
    o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1;
    o[5] = tmp12 - tmp1;
    o[3] = tmp13 + tmp0;
    o[4] = tmp13 - tmp0;
  }
};

void idct_islow(const int16_t* coef, const uint16_t* quant_natural, unsigned char* out, size_t stride) {
  wide ws[64];
  Idct8 t;
  for (int c = 0; c < 8; ++c) {   // pass 1: columns of the dequantised block -> workspace, 2 extra bits kept
    const int16_t* in = coef + c;
    const uint16_t* q = quant_natural + c;
    t.run((wide)in[0] * q[0], (wide)in[8] * q[8], (wide)in[16] * q[16], (wide)in[24] * q[24], (wide)in[32] * q[32], (wide)in[40] * q[40],
          (wide)in[48] * q[48], (wide)in[56] * q[56]);
    for (int k = 0; k < 8; ++k) ws[8 * k + c] = descale(t.o[k], kConstBits - kPass1Bits);
  }
  for (int r = 0; r < 8; ++r) {   // pass 2: rows of the workspace -> samples (+128, clamped through libjpeg's table)
    const wide* w = ws + 8 * r;
    t.run(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
    unsigned char* o = out + stride * r;
    for (int k = 0; k < 8; ++k) o[k] = g_range_limit[(int)(descale(t.o[k], kConstBits + kPass1Bits + 3) & 1023)];
  }
}

int read_file(const char* path, std::vector<unsigned char>* file, std::string* msg) {
  FILE* f = fopen(path, "rb");
  if (!f) return fail(R3D_ERR_INVALID, path, "cannot open", msg);
  unsigned char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file->insert(file->end(), buf, buf + n);
  fclose(f);
  return R3D_OK;
}

struct Planes {
  int width = 0, height = 0, n_comp = 0, hmax = 1, vmax = 1;
  int h[3] = {1, 1, 1}, v[3] = {1, 1, 1};
  size_t stride[3] = {0, 0, 0};                 // samples per row of each decoded plane (whole blocks)
  unsigned char* data[3] = {nullptr, nullptr, nullptr};   // in the calling thread's scratch (buffers 1..3)
};

// Checks the file, and (decode) entropy-decodes it and inverse-transforms the first `keep` components into planes of whole
// blocks.  keep = 1: luma only.  header query: decode = false.
int decode_components(const char* path, bool decode, int keep, Planes* P, std::string* msg) {
  std::vector<unsigned char>& file = r3d_host::scratch(0);
  int rc = read_file(path, &file, msg);
  if (rc) return rc;
  Jpeg j;
  if ((rc = parse_headers(path, file, &j, msg))) return rc;
  P->width = j.width;
  P->height = j.height;
  P->n_comp = j.n_comp;
  if (j.progressive) return fail(R3D_ERR_UNSUPPORTED, path, "progressive JPEG (only sequential Huffman files are decoded natively)", msg);
  if (j.precision != 8) return fail(R3D_ERR_UNSUPPORTED, path, "12-bit JPEG", msg);
  if (j.width < 1 || j.height < 1) return fail(R3D_ERR_UNSUPPORTED, path, "JPEG without its height in the frame header (DNL)", msg);
  if (j.n_comp != 1 && j.n_comp != 3) return fail(R3D_ERR_UNSUPPORTED, path, "a JPEG of 2 or 4 components (CMYK / YCCK)", msg);
  if (j.n_comp == 3 && j.adobe && j.adobe_transform == 0)
    return fail(R3D_ERR_UNSUPPORTED, path, "an Adobe RGB JPEG (no luma component to take)", msg);
  if (j.n_comp == 3 && !j.adobe && j.comp[0].id == 'R' && j.comp[1].id == 'G' && j.comp[2].id == 'B')
    return fail(R3D_ERR_UNSUPPORTED, path, "an RGB JPEG (no luma component to take)", msg);
  if (!j.scan_is_whole_image) return fail(R3D_ERR_UNSUPPORTED, path, "a multi-scan sequential JPEG", msg);
  int hmax = 1, vmax = 1;
  for (int c = 0; c < j.n_comp; ++c) {
    hmax = std::max(hmax, j.comp[c].h);
    vmax = std::max(vmax, j.comp[c].v);
  }
  if (j.n_comp == 1) {   // a single-component scan is not interleaved: one block per "MCU", whatever the sampling factors say
    j.comp[0].h = j.comp[0].v = 1;
    hmax = vmax = 1;
  }
  if (j.comp[0].h != hmax || j.comp[0].v != vmax)
    return fail(R3D_ERR_UNSUPPORTED, path, "luma stored below full resolution", msg);
  for (int c = 0; c < j.n_comp; ++c)
    if (!j.have_quant[j.comp[c].tq] || !j.dc[j.comp[c].td].present || !j.ac[j.comp[c].ta].present)
      return fail(R3D_ERR_INVALID, path, "a table the scan needs is missing", msg);
  P->hmax = hmax;
  P->vmax = vmax;
  keep = std::min(keep, j.n_comp);
  for (int c = 0; c < j.n_comp && c < 3; ++c) {
    P->h[c] = j.comp[c].h;
    P->v[c] = j.comp[c].v;
  }
  if (!decode) return R3D_OK;
  const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
  const int mcus_x = (j.width + mcu_w - 1) / mcu_w, mcus_y = (j.height + mcu_h - 1) / mcu_h;
  uint16_t q_nat[3][64];
  for (int c = 0; c < keep; ++c) {
    P->stride[c] = (size_t)mcus_x * 8 * j.comp[c].h;
    std::vector<unsigned char>& plane = r3d_host::scratch(1 + c);
    plane.resize(P->stride[c] * (size_t)mcus_y * 8 * j.comp[c].v);
    P->data[c] = plane.data();
    for (int k = 0; k < 64; ++k) q_nat[c][kNatural[k]] = j.quant[j.comp[c].tq][k];
  }
  Bits b{file.data() + j.scan_begin, file.data() + file.size()};
  int pred[4] = {0, 0, 0, 0};
  int until_restart = j.restart_interval, next_rst = 0;
  int16_t coef[64];
  for (int my = 0; my < mcus_y; ++my) {
    for (int mx = 0; mx < mcus_x; ++mx) {
      if (j.restart_interval) {
        if (until_restart == 0) {
          // byte-align, expect RSTn
          b.n = 0;
          b.buf = 0;
          if (!b.marker) {   // the marker has not been met yet: it must be right here
            const unsigned char* q = b.p;
            while (q < b.end && *q == 0xff) ++q;
            if (q == b.p || q >= b.end) return fail(R3D_ERR_INVALID, path, "restart marker missing", msg);
            b.marker = *q;
            b.p = q + 1;
          }
          if (b.marker != 0xd0 + next_rst) return fail(R3D_ERR_INVALID, path, "restart marker out of sequence", msg);
          b.marker = 0;
          next_rst = (next_rst + 1) & 7;
          pred[0] = pred[1] = pred[2] = pred[3] = 0;
          until_restart = j.restart_interval;
        }
        --until_restart;
      }
      for (int c = 0; c < j.n_comp; ++c) {
        const Component& cp = j.comp[c];
        const Huff &hd = j.dc[cp.td], &ha = j.ac[cp.ta];
        const bool kept = c < keep;
        for (int by = 0; by < cp.v; ++by)
          for (int bx = 0; bx < cp.h; ++bx) {
            if (kept) memset(coef, 0, sizeof(coef));
            int s = decode_symbol(b, hd);
            if (s < 0 || s > 15) return fail(R3D_ERR_INVALID, path, "corrupt entropy-coded data", msg);
            if (s) pred[c] = (int)((unsigned)pred[c] + (unsigned)extend(b.get(s), s));   // (hostile data may wrap, never overflow)
            if (kept) coef[0] = (int16_t)pred[c];
            for (int k = 1; k < 64; ++k) {
              if (b.n < 16) b.fill();
              const int32_t f = ha.fast[b.peek(kFastBits)];
              if (f) {   // code and magnitude in one step
                k += (f >> 8) & 15;
                if (k > 63) return fail(R3D_ERR_INVALID, path, "corrupt entropy-coded data", msg);
                b.drop(f & 255);
                if (kept) coef[kNatural[k]] = (int16_t)(f >> 16);
                continue;
              }
              const int rs = decode_symbol(b, ha);
              if (rs < 0) return fail(R3D_ERR_INVALID, path, "corrupt entropy-coded data", msg);
              const int r = rs >> 4;
              s = rs & 15;
              if (s == 0) {
                if (r != 15) break;   // end of block
                k += 15;
                continue;
              }
              k += r;
              if (k > 63) return fail(R3D_ERR_INVALID, path, "corrupt entropy-coded data", msg);
              const int v = extend(b.get(s), s);
              if (kept) coef[kNatural[k]] = (int16_t)v;
            }
            if (kept)
              idct_islow(coef, q_nat[c],
                         P->data[c] + ((size_t)my * cp.v + by) * 8 * P->stride[c] + ((size_t)mx * cp.h + bx) * 8, P->stride[c]);
          }
      }
    }
  }
  return R3D_OK;
}

// out == NULL: header query
int decode_gray_impl(const char* path, unsigned char* out, size_t cap_bytes, int* h_out, int* w_out, std::string* msg) {
  r3d_host::ScratchScope scope;
  Planes P;
  const int rc = decode_components(path, out != nullptr, 1, &P, msg);
  if (h_out) *h_out = P.height;
  if (w_out) *w_out = P.width;
  if (rc || !out) return rc;
  if (cap_bytes < (size_t)P.width * P.height) return fail(R3D_ERR_NOMEM, path, "output buffer too small", msg);
  for (int y = 0; y < P.height; ++y) memcpy(out + (size_t)y * P.width, P.data[0] + (size_t)y * P.stride[0], (size_t)P.width);
  return R3D_OK;
}

// ---- colour: what PIL's Image.open(path) (the reference's genply_noRGB, pixel_to_camera.py:58-60) gets from libjpeg --------
// libjpeg's default colour decode: every component through the islow IDCT, chroma brought to full resolution by "fancy"
// (triangle-filter) upsampling (jdsample.c: h2v1 / h2v2; plain replication when a chroma row has fewer than three samples),
// then YCbCr -> RGB with 16-bit fixed-point tables (jdcolor.c).
// one chroma row pair -> one full-resolution row.  near / far: the chroma rows this output row lies between (the nearer one
// weighs 3, the other 1; the same row twice when the component is not subsampled vertically); dw real samples per row.
// (h1v2 -- 4:4:0 -- is refused by the caller.)  jdsample.c walks a row with running sums and special first / last columns;
// the same numbers come out of ONE formula per output parity over a row padded by a copy of its end samples --
//   h2v2: s[x] = 3 near[x] + far[x];  out[2x] = (3 s[x] + s[x-1] + 8) >> 4,  out[2x+1] = (3 s[x] + s[x+1] + 7) >> 4
//         (first column: (4 s[0] + 8) >> 4 = the formula with s[-1] = s[0]; last: (4 s + 7) >> 4 likewise)
//   h2v1: out[2x] = (3 in[x] + in[x-1] + 1) >> 2,  out[2x+1] = (3 in[x] + in[x+1] + 2) >> 2   (ends: (4 in + 1 or 2) >> 2 = in)
// -- loops without a carried dependency, which the compiler turns into 16-bit vector arithmetic.
// sums: dw + 2 shorts of scratch; out: room for 2 dw bytes (one more than out_w when the width is odd).
void upsample_row(const unsigned char* __restrict near_, const unsigned char* __restrict far_, int dw, bool h2, bool v2, bool fancy,
                  unsigned char* __restrict out, int out_w, int16_t* __restrict sums) {
  if (!h2 && !v2) {
    memcpy(out, near_, (size_t)out_w);
    return;
  }
  if (!fancy) {   // replication
    for (int x = 0; x < out_w; ++x) out[x] = near_[h2 ? x >> 1 : x];
    return;
  }
  int16_t* s = sums + 1;
  if (v2) {
    for (int x = 0; x < dw; ++x) s[x] = (int16_t)(3 * near_[x] + far_[x]);
  } else {
    for (int x = 0; x < dw; ++x) s[x] = near_[x];
  }
  s[-1] = s[0];
  s[dw] = s[dw - 1];
  if (v2) {
    for (int x = 0; x < dw; ++x) {
      out[2 * x] = (unsigned char)((3 * s[x] + s[x - 1] + 8) >> 4);
      out[2 * x + 1] = (unsigned char)((3 * s[x] + s[x + 1] + 7) >> 4);
    }
  } else {
    for (int x = 0; x < dw; ++x) {
      out[2 * x] = (unsigned char)((3 * s[x] + s[x - 1] + 1) >> 2);
      out[2 * x + 1] = (unsigned char)((3 * s[x] + s[x + 1] + 2) >> 2);
    }
  }
}

// out == NULL: header query.  R,G,B bytes, [H][W][3].
int decode_rgb_impl(const char* path, unsigned char* out, size_t cap_bytes, int* h_out, int* w_out, int* comps_out, std::string* msg) {
  r3d_host::ScratchScope scope;
  Planes P;
  const int rc = decode_components(path, out != nullptr, 3, &P, msg);
  if (h_out) *h_out = P.height;
  if (w_out) *w_out = P.width;
  if (comps_out) *comps_out = P.n_comp;
  if (rc) return rc;
  if (P.n_comp == 3) {
    if (P.h[1] != P.h[2] || P.v[1] != P.v[2] || !((P.hmax == P.h[1] || P.hmax == 2 * P.h[1]) && (P.vmax == P.v[1] || P.vmax == 2 * P.v[1])) ||
        (P.hmax == P.h[1] && P.vmax == 2 * P.v[1]))
      return fail(R3D_ERR_UNSUPPORTED, path, "a chroma layout other than 4:4:4, 4:2:2 or 4:2:0", msg);
  }
  if (!out) return R3D_OK;
  const size_t W = (size_t)P.width, H = (size_t)P.height;
  if (cap_bytes < W * H * 3) return fail(R3D_ERR_NOMEM, path, "output buffer too small", msg);
  if (P.n_comp == 1) {
    for (size_t y = 0; y < H; ++y) {
      const unsigned char* src = P.data[0] + y * P.stride[0];
      unsigned char* o = out + y * W * 3;
      for (size_t x = 0; x < W; ++x) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = src[x];
    }
    return R3D_OK;
  }
  const bool h2 = P.hmax == 2 * P.h[1], v2 = P.vmax == 2 * P.v[1];
  const int dw = h2 ? (P.width + 1) / 2 : P.width, dh = v2 ? (P.height + 1) / 2 : P.height;   // real chroma samples
  const bool fancy = dw > 2;   // jinit_upsampler: fancy upsampling needs more than two samples in a row
  std::vector<unsigned char> cb(W + 2), cr(W + 2), red(W), green(W), blue(W);
  std::vector<int16_t> sums((size_t)dw + 2);
  for (size_t y = 0; y < H; ++y) {
    // chroma rows this output row lies between: row y/2 and its neighbour above (even y) or below (odd y), edges replicated
    const int cy = v2 ? (int)(y >> 1) : (int)y;
    int other = cy;
    if (v2) {
      other = (y & 1) ? cy + 1 : cy - 1;
      if (other < 0) other = 0;
      if (other > dh - 1) other = dh - 1;
    }
    for (int c = 1; c <= 2; ++c) {
      const unsigned char* near_ = P.data[c] + (size_t)cy * P.stride[c];
      const unsigned char* far_ = P.data[c] + (size_t)other * P.stride[c];
      upsample_row(near_, far_, dw, h2, v2, fancy, c == 1 ? cb.data() : cr.data(), (int)W, sums.data());
    }
    const unsigned char* yrow = P.data[0] + y * P.stride[0];
    unsigned char* o = out + y * W * 3;
    // jdcolor.c's tables written out -- R = Y + ((91881 Cr' + 32768) >> 16), B = Y + ((116130 Cb' + 32768) >> 16),
    // G = Y + ((-22554 Cb' - 46802 Cr' + 32768) >> 16), Cb' = Cb - 128 -- as arithmetic into three planes (vector code), then
    // the planes interleaved
    {
      const unsigned char* __restrict yp = yrow;
      const unsigned char* __restrict bp = cb.data();
      const unsigned char* __restrict rp = cr.data();
      unsigned char* __restrict R_ = red.data();
      unsigned char* __restrict G_ = green.data();
      unsigned char* __restrict B_ = blue.data();
      for (size_t x = 0; x < W; ++x) {
        const int yy = yp[x], cbv = bp[x] - 128, crv = rp[x] - 128;
        const int r = yy + ((91881 * crv + 32768) >> 16);
        const int g = yy + ((-22554 * cbv - 46802 * crv + 32768) >> 16);
        const int bl = yy + ((116130 * cbv + 32768) >> 16);
        R_[x] = (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
        G_[x] = (unsigned char)(g < 0 ? 0 : g > 255 ? 255 : g);
        B_[x] = (unsigned char)(bl < 0 ? 0 : bl > 255 ? 255 : bl);
      }
      for (size_t x = 0; x < W; ++x) {
        o[3 * x] = R_[x];
        o[3 * x + 1] = G_[x];
        o[3 * x + 2] = B_[x];
      }
    }
  }
  return R3D_OK;
}

template <typename F>
int guarded(const char* path, std::string* msg, F&& f) {
  try {
    return f();
  } catch (const std::exception& e) {
    try {
      *msg = std::string("'") + path + "': " + e.what();
    } catch (...) {
    }
    return R3D_ERR_NOMEM;
  }
}

int decode_gray(const char* path, unsigned char* out, size_t cap_bytes, int* h_out, int* w_out, std::string* msg) {
  return guarded(path, msg, [&] { return decode_gray_impl(path, out, cap_bytes, h_out, w_out, msg); });
}
int decode_rgb(const char* path, unsigned char* out, size_t cap_bytes, int* h_out, int* w_out, int* comps_out, std::string* msg) {
  return guarded(path, msg, [&] { return decode_rgb_impl(path, out, cap_bytes, h_out, w_out, comps_out, msg); });
}

}  // namespace

extern "C" {

int r3d_jpeg_gray_info(const char* path, int* height, int* width) {
  if (!path) {
    r3d_set_error("r3d_jpeg_gray_info: path is NULL");
    return R3D_ERR_INVALID;
  }
  std::string msg;
  const int rc = decode_gray(path, nullptr, 0, height, width, &msg);
  if (rc) r3d_set_error("%s", msg.c_str());
  return rc;
}

int r3d_jpeg_gray_decode_batch(const char* const* paths, int n_files, unsigned char* h_out, int height, int width) {
  if (n_files < 0 || (n_files > 0 && (!paths || !h_out)) || height <= 0 || width <= 0) {
    r3d_set_error("r3d_jpeg_gray_decode_batch: bad argument");
    return R3D_ERR_INVALID;
  }
  const size_t frame_bytes = (size_t)height * width;
  return r3d_host::run_batch(n_files, "JPEG decode failed", [&](int k, std::string* msg) -> int {
    int h = 0, w = 0;
    int rc = paths[k] ? decode_gray(paths[k], h_out + frame_bytes * k, frame_bytes, &h, &w, msg) : R3D_ERR_INVALID;
    if (rc == R3D_OK && (h != height || w != width)) {
      rc = R3D_ERR_INVALID;
      *msg = std::string("'") + paths[k] + "' is " + std::to_string(w) + "x" + std::to_string(h) + ", the batch expects " +
             std::to_string(width) + "x" + std::to_string(height);
    }
    return rc;
  });
}

int r3d_jpeg_rgb_info(const char* path, int* height, int* width, int* components) {
  if (!path) {
    r3d_set_error("r3d_jpeg_rgb_info: path is NULL");
    return R3D_ERR_INVALID;
  }
  std::string msg;
  const int rc = decode_rgb(path, nullptr, 0, height, width, components, &msg);
  if (rc) r3d_set_error("%s", msg.c_str());
  return rc;
}

int r3d_jpeg_rgb_decode_batch(const char* const* paths, int n_files, unsigned char* h_out, int height, int width) {
  if (n_files < 0 || (n_files > 0 && (!paths || !h_out)) || height <= 0 || width <= 0) {
    r3d_set_error("r3d_jpeg_rgb_decode_batch: bad argument");
    return R3D_ERR_INVALID;
  }
  const size_t frame_bytes = (size_t)height * width * 3;
  return r3d_host::run_batch(n_files, "JPEG decode failed", [&](int k, std::string* msg) -> int {
    int h = 0, w = 0, nc = 0;
    int rc = paths[k] ? decode_rgb(paths[k], h_out + frame_bytes * k, frame_bytes, &h, &w, &nc, msg) : R3D_ERR_INVALID;
    if (rc == R3D_OK && (h != height || w != width)) {
      rc = R3D_ERR_INVALID;
      *msg = std::string("'") + paths[k] + "' is " + std::to_string(w) + "x" + std::to_string(h) + ", the batch expects " +
             std::to_string(width) + "x" + std::to_string(height);
    }
    return rc;
  });
}

}  // extern "C"
