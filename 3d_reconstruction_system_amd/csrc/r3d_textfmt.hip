// f1 on the device: the reference's text -- camera / world txt lines and PLY vertex rows -- formatted by the GPU.
//
// What the reference spends its time on is turning points into text (camera_to_world.py:80-81 and 103-104: `str(x) + ','
// + str(y) + ',' + str(z) + '\n'`; genply, :112-134: "%.4f %.4f %.4f \n" per vertex).  Both are INTEGER algorithms once the
// double's bits are taken apart: repr() is the shortest digit string that reads back as the same double (Schubfach: three
// 64 x 128-bit products against a table of powers of ten), "%.4f" is round-half-even of the exact value x * 10^4 (one
// 53 x 14-bit product and a shift).  So the cloud never has to leave the GPU as fp64: the text does (41 B/point of camera
// txt instead of 24 + 24 B/point of camera and world clouds into pageable memory), and the host only copies it into files.
//
//   pass A  text_rows_kernel<false>   one point per lane, 256 per workgroup: the row's length -> bytes per tile
//   scan    text_scan_kernel          exclusive prefix sum over the tiles (one workgroup, 8192 tiles per round)
//   pass B  text_rows_kernel<true>    the same arithmetic again, a block scan of the lengths, every lane writes its row's
//                                     characters into LDS at its place in the tile's text, the tile goes out in aligned
//                                     16-byte stores (LDS is laid out with the tile's misalignment, so both sides agree)
// Tiles never straddle a segment (= a file of the per-frame camera txt), so a segment's text starts where its first tile
// does.  The CPU formatter (r3d_format.cpp) stays: it is the checker of this one (tests/test_gpu_textfmt.py: goldens, every
// binade, ties, 3 M random bit patterns) and takes the requests this one refuses -- "%.4f" of |x| >= 2^40, whose rows do
// not fit the 80 bytes a lane has (R3D_ERR_UNSUPPORTED).
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "r3d_hostpool.h"
#include "r3d_internal.h"
#include "r3d_pow10_table.h"

namespace {

constexpr int kTile = 256;      // points per workgroup, one per lane
constexpr int kRowMax = 80;     // bytes a row may take: 3 x 24 (repr) + 3, or 3 x 19 (%.4f below 2^40) + 3 + "255 255 255 0\n"
constexpr int kScanPer = 8;     // tiles per lane and round of the scan kernel

__device__ uint64_t g_pow10[(r3d_pow10::kMax - r3d_pow10::kMin + 1) * 2];   // {hi, lo} of r3d_pow10::kG, uploaded once per device
__constant__ uint64_t kPow10d[18] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull,
                                     1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull,
                                     100000000000000ull, 1000000000000000ull, 10000000000000000ull, 100000000000000000ull};

// ---- the arithmetic (mirrors r3d_format.cpp line by line; 128-bit values as two words) -----------------------------------

// bits 128..191 of the 192-bit product g * cp, bit 0 set when anything non-zero lies below ("round to odd")
__device__ __forceinline__ uint64_t round_to_odd(uint64_t g_hi, uint64_t g_lo, uint64_t cp) {
  const uint64_t x_hi = __umul64hi(cp, g_lo);
  const uint64_t p_lo = cp * g_hi, p_hi = __umul64hi(cp, g_hi);
  const uint64_t y0 = p_lo + x_hi;
  const uint64_t y1 = p_hi + (y0 < p_lo ? 1u : 0u);
  return y1 | (uint64_t)(y0 > 1);
}

__device__ __forceinline__ int floor_log10_pow2(int e) { return (e * 1262611) >> 22; }
__device__ __forceinline__ int floor_log10_three_quarters_pow2(int e) { return (e * 1262611 - 524031) >> 22; }
__device__ __forceinline__ int floor_log2_pow10(int e) { return (e * 1741647) >> 19; }

// decimal digits of v, 1 <= v < 10^17
__device__ __forceinline__ int count_digits(uint64_t v) {
  const int t = ((64 - __clzll((long long)(v | 1))) * 1233) >> 12;
  return t + (v >= kPow10d[t] ? 1 : 0);
}

// shortest decimal that reads back as the double (Schubfach; r3d_format.cpp shortest_decimal): value = digits * 10^exp10
__device__ __forceinline__ void shortest_decimal(uint64_t significand, int biased_exponent, uint64_t* digits, int* exp10) {
  uint64_t c;
  int q;
  if (biased_exponent != 0) {
    c = (1ull << 52) | significand;
    q = biased_exponent - 1075;
    if (0 <= -q && -q < 53 && (c & ((1ull << -q) - 1)) == 0) {   // an integer below 2^53: its own digits
      *digits = c >> -q;
      *exp10 = 0;
      return;
    }
  } else {
    c = significand;
    q = -1074;
  }
  const bool even = (c & 1) == 0;
  const bool lower_is_closer = significand == 0 && biased_exponent > 1;
  const uint64_t cbl = 4 * c - 2 + (lower_is_closer ? 1 : 0), cb = 4 * c, cbr = 4 * c + 2;
  const int k = lower_is_closer ? floor_log10_three_quarters_pow2(q) : floor_log10_pow2(q);
  const int h = q + floor_log2_pow10(-k) + 1;
  const uint64_t g_hi = g_pow10[2 * (-k - r3d_pow10::kMin)], g_lo = g_pow10[2 * (-k - r3d_pow10::kMin) + 1];
  const uint64_t vbl = round_to_odd(g_hi, g_lo, cbl << h), vb = round_to_odd(g_hi, g_lo, cb << h), vbr = round_to_odd(g_hi, g_lo, cbr << h);
  const uint64_t lower = vbl + (even ? 0 : 1), upper = vbr - (even ? 0 : 1);
  const uint64_t s = vb / 4;
  if (s >= 10) {   // a multiple of 10^(k+1) inside the interval is one digit shorter
    const uint64_t sp = s / 10;
    const bool up_inside = lower <= 40 * sp, wp_inside = 40 * sp + 40 <= upper;
    if (up_inside != wp_inside) {
      *digits = sp + (wp_inside ? 1 : 0);
      *exp10 = k + 1;
      return;
    }
  }
  const bool u_inside = lower <= 4 * s, w_inside = 4 * s + 4 <= upper;
  if (u_inside != w_inside) {
    *digits = s + (w_inside ? 1 : 0);
    *exp10 = k;
    return;
  }
  const uint64_t mid = 4 * s + 2;   // both inside: the closer one, the even one on a tie
  const bool round_up = vb > mid || (vb == mid && (s & 1) != 0);
  *digits = s + (round_up ? 1 : 0);
  *exp10 = k;
}

// One number, analysed: everything its characters follow from.  `form` says which layout (below), `len` how long it is.
struct Num {
  uint32_t hi, lo;   // the digits as hi * 10^8 + lo
  int nd;            // how many of them
  int decpt;         // repr: position of the decimal point relative to the first digit; fixed4: unused
  int len;
  uint8_t form, neg;
};
enum : uint8_t { kNan = 0, kInf, kZeroRepr, kReprExp, kReprSmall, kReprInt, kReprMid, kFixed4, kUint };

__device__ __forceinline__ Num plan_repr(double x) {
  Num n{};
  if (x != x) {
    n.form = kNan;
    n.len = 3;
    return n;
  }
  const uint64_t bits = (uint64_t)__double_as_longlong(x);
  n.neg = (uint8_t)(bits >> 63);
  const uint64_t mag = bits & 0x7fffffffffffffffull;
  if (mag == 0x7ff0000000000000ull) {
    n.form = kInf;
    n.len = 3 + n.neg;
    return n;
  }
  if (mag == 0) {
    n.form = kZeroRepr;
    n.len = 3 + n.neg;
    return n;
  }
  uint64_t digits;
  int e10;
  shortest_decimal(mag & 0xfffffffffffffull, (int)(mag >> 52), &digits, &e10);
  while (digits % 10 == 0) {   // zeros at the end belong to the exponent
    digits /= 10;
    ++e10;
  }
  n.nd = count_digits(digits);
  n.hi = (uint32_t)(digits / 100000000u);
  n.lo = (uint32_t)(digits % 100000000u);
  n.decpt = e10 + n.nd;
  if (n.decpt <= -4 || n.decpt > 16) {
    int e = n.decpt - 1;
    if (e < 0) e = -e;
    n.form = kReprExp;
    n.len = n.neg + (n.nd > 1 ? n.nd + 1 : 1) + 2 + (e >= 100 ? 3 : 2);
  } else if (n.decpt <= 0) {
    n.form = kReprSmall;
    n.len = n.neg + 2 - n.decpt + n.nd;
  } else if (n.decpt >= n.nd) {
    n.form = kReprInt;
    n.len = n.neg + n.decpt + 2;
  } else {
    n.form = kReprMid;
    n.len = n.neg + n.nd + 1;
  }
  return n;
}

// "%.4f": round-half-even of the EXACT value |x| * 10^4 (what printf prints), for |x| < 2^40.  *giant: a finite |x| >= 2^40.
__device__ __forceinline__ Num plan_fixed4(double x, bool* giant) {
  Num n{};
  if (x != x) {
    n.form = kNan;
    n.len = 3;
    return n;
  }
  const uint64_t bits = (uint64_t)__double_as_longlong(x);
  n.neg = (uint8_t)(bits >> 63);
  const uint64_t mag = bits & 0x7fffffffffffffffull;
  if (mag == 0x7ff0000000000000ull) {
    n.form = kInf;
    n.len = 3 + n.neg;
    return n;
  }
  const int biased = (int)(mag >> 52);
  uint64_t scaled = 0;
  if (biased >= 1023 + 40) {
    *giant = true;
  } else if (mag != 0) {
    const uint64_t m = biased ? ((mag & 0xfffffffffffffull) | (1ull << 52)) : (mag & 0xfffffffffffffull);   // |x| = m * 2^e
    const int e = (biased ? biased : 1) - 1075;
    const uint64_t p_lo = m * 10000u, p_hi = __umul64hi(m, 10000u);   // m * 10^4 < 2^67
    if (e >= 0) {
      scaled = p_lo << e;   // |x| < 2^40 keeps the product below 2^54: p_hi is 0 here
    } else {
      const int sh = -e;    // 1 .. 1074
      if (sh < 120) {
        // q = prod >> sh; round up when the first dropped bit is set and (a lower one is, or q is odd)
        uint64_t q, round_bit;
        bool sticky;
        const int k = sh - 1;   // index of the first dropped bit
        if (k < 64) {
          round_bit = (p_lo >> k) & 1;
          sticky = k > 0 && (p_lo & ((1ull << k) - 1)) != 0;
        } else {
          round_bit = (p_hi >> (k - 64)) & 1;
          sticky = p_lo != 0 || (k > 64 && (p_hi & ((1ull << (k - 64)) - 1)) != 0);
        }
        if (sh < 64)
          q = (p_lo >> sh) | (p_hi << (64 - sh));
        else if (sh == 64)
          q = p_hi;
        else
          q = p_hi >> (sh - 64);
        scaled = q + ((round_bit && (sticky || (q & 1))) ? 1 : 0);
      }
    }
  }
  const uint64_t ip = scaled / 10000u;
  n.form = kFixed4;
  n.decpt = (int)(scaled % 10000u);          // the four decimals
  n.nd = ip ? count_digits(ip) : 1;
  n.hi = (uint32_t)(ip / 100000000u);
  n.lo = (uint32_t)(ip % 100000000u);
  n.len = n.neg + n.nd + 5;
  return n;
}

__device__ __forceinline__ Num plan_uint(uint32_t v) {
  Num n{};
  n.form = kUint;
  n.lo = v;
  n.nd = v ? count_digits(v) : 1;
  n.len = n.nd;
  return n;
}

// digit i (0 = most significant of nd) goes to p[i + (i >= split ? gap : 0)]
__device__ __forceinline__ void put_digits(char* p, uint32_t hi, uint32_t lo, int nd, int split, int gap) {
  uint32_t v = lo;
  const int from_lo = nd < 8 ? nd : 8;
  for (int j = 0; j < from_lo; ++j) {
    const int i = nd - 1 - j;
    p[i + (i >= split ? gap : 0)] = (char)('0' + v % 10);
    v /= 10;
  }
  v = hi;
  for (int j = 8; j < nd; ++j) {
    const int i = nd - 1 - j;
    p[i + (i >= split ? gap : 0)] = (char)('0' + v % 10);
    v /= 10;
  }
}

// the characters of one analysed number at p (LDS); returns p + n.len
__device__ __forceinline__ char* emit(char* p, const Num& n) {
  char* const end = p + n.len;
  if (n.form == kNan) {
    p[0] = 'n'; p[1] = 'a'; p[2] = 'n';
    return end;
  }
  if (n.neg) *p++ = '-';
  switch (n.form) {
    case kInf:
      p[0] = 'i'; p[1] = 'n'; p[2] = 'f';
      break;
    case kZeroRepr:
      p[0] = '0'; p[1] = '.'; p[2] = '0';
      break;
    case kReprExp: {
      put_digits(p, n.hi, n.lo, n.nd, 1, 1);
      if (n.nd > 1) p[1] = '.';
      p += n.nd > 1 ? n.nd + 1 : 1;
      int e = n.decpt - 1;
      p[0] = 'e';
      p[1] = e < 0 ? '-' : '+';
      if (e < 0) e = -e;
      if (e >= 100) {
        p[2] = (char)('0' + e / 100);
        p[3] = (char)('0' + (e / 10) % 10);
        p[4] = (char)('0' + e % 10);
      } else {
        p[2] = (char)('0' + e / 10);
        p[3] = (char)('0' + e % 10);
      }
      break;
    }
    case kReprSmall:   // 0.000ddd
      p[0] = '0';
      p[1] = '.';
      for (int z = 0; z < -n.decpt; ++z) p[2 + z] = '0';
      put_digits(p + 2 - n.decpt, n.hi, n.lo, n.nd, n.nd, 0);
      break;
    case kReprInt:     // ddd000.0
      put_digits(p, n.hi, n.lo, n.nd, n.nd, 0);
      for (int z = n.nd; z < n.decpt; ++z) p[z] = '0';
      p[n.decpt] = '.';
      p[n.decpt + 1] = '0';
      break;
    case kReprMid:     // dd.ddd
      put_digits(p, n.hi, n.lo, n.nd, n.decpt, 1);
      p[n.decpt] = '.';
      break;
    case kFixed4: {
      put_digits(p, n.hi, n.lo, n.nd, n.nd, 0);
      p += n.nd;
      const unsigned frac = (unsigned)n.decpt;
      p[0] = '.';
      p[1] = (char)('0' + frac / 1000);
      p[2] = (char)('0' + (frac / 100) % 10);
      p[3] = (char)('0' + (frac / 10) % 10);
      p[4] = (char)('0' + frac % 10);
      break;
    }
    default:           // kUint
      put_digits(p, 0, n.lo, n.nd, n.nd, 0);
      break;
  }
  return end;
}

// ---- the kernels ---------------------------------------------------------------------------------------------------------

struct TextJob {
  const void* xyz;      // [n][3] f32 / f64
  const void* aux;      // kind 0: integer third column (u8 / u16) or NULL; kind 2: colours, aux_stride bytes apart
  int64_t n_points;
  int64_t seg_points;   // points per segment (> 0)
  int tiles_per_seg;
  int kind, f64, aux_stride;
};

template <bool F64>
__device__ __forceinline__ double coord(const void* xyz, int64_t i, int a) {
  return F64 ? static_cast<const double*>(xyz)[i * 3 + a] : (double)static_cast<const float*>(xyz)[i * 3 + a];
}

// exclusive scan of one value per lane over the 256 lanes of the workgroup; *total = the sum.  `sh` holds 4 + 1 words.
__device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t* sh, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(inc, d, 64);
    if (lane >= d) inc += up;
  }
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += sh[w];
  *total = sh[0] + sh[1] + sh[2] + sh[3];
  return base + inc - v;
}

template <bool EMIT, bool F64>
__global__ __launch_bounds__(kTile) void text_rows_kernel(TextJob job, uint32_t* __restrict__ tile_len,
                                                         const uint64_t* __restrict__ tile_off, char* __restrict__ text,
                                                         unsigned int* __restrict__ flags) {
  __shared__ uint32_t sh_scan[4];
  __shared__ __attribute__((aligned(16))) char sh_text[EMIT ? kTile * kRowMax + 16 : 16];
  const int64_t t = blockIdx.x;
  const int64_t seg = t / job.tiles_per_seg;
  const int64_t in_seg = (t - seg * job.tiles_per_seg) * (int64_t)kTile + threadIdx.x;
  const int64_t i = seg * job.seg_points + in_seg;
  const bool live = in_seg < job.seg_points && i < job.n_points;
  Num a{}, b{}, c{}, r{}, g{}, bl{};
  uint32_t len = 0;
  if (live) {
    const double x = coord<F64>(job.xyz, i, 0), y = coord<F64>(job.xyz, i, 1), z = coord<F64>(job.xyz, i, 2);
    if (job.kind == R3D_TEXT_XYZ_TXT) {
      a = plan_repr(x);
      b = plan_repr(y);
      if (job.aux)
        c = plan_uint(job.aux_stride == 1 ? static_cast<const uint8_t*>(job.aux)[i] : static_cast<const uint16_t*>(job.aux)[i]);
      else
        c = plan_repr(z);
      len = a.len + b.len + c.len + 3;                       // two commas, newline
    } else {
      bool giant = false;
      a = plan_fixed4(x, &giant);
      b = plan_fixed4(y, &giant);
      c = plan_fixed4(z, &giant);
      if (giant) atomicOr(flags, 1u);
      len = a.len + b.len + c.len + 4;                       // "x y z \n"
      if (job.kind == R3D_TEXT_PLY_ROWS_RGB) {
        const uint8_t* col = static_cast<const uint8_t*>(job.aux) + i * job.aux_stride;
        r = plan_uint(col[0]);
        g = plan_uint(col[1]);
        bl = plan_uint(col[2]);
        len = a.len + b.len + c.len + r.len + g.len + bl.len + 8;   // "x y z r g b 0\n"
      }
    }
  }
  uint32_t total;
  const uint32_t at = block_scan_256(len, sh_scan, &total);
  if (!EMIT) {
    if (threadIdx.x == 0) tile_len[t] = total;
    return;
  }
  const uint64_t off = tile_off[t];
  const uint32_t shift = (uint32_t)((reinterpret_cast<uintptr_t>(text) + off) & 15);   // LDS byte j <-> address text + off - shift + j: both 16-byte aligned
  if (live) {
    char* p = sh_text + shift + at;
    p = emit(p, a);
    *p++ = job.kind == R3D_TEXT_XYZ_TXT ? ',' : ' ';
    p = emit(p, b);
    *p++ = job.kind == R3D_TEXT_XYZ_TXT ? ',' : ' ';
    p = emit(p, c);
    if (job.kind == R3D_TEXT_PLY_ROWS_RGB) {
      *p++ = ' ';
      p = emit(p, r);
      *p++ = ' ';
      p = emit(p, g);
      *p++ = ' ';
      p = emit(p, bl);
      *p++ = ' ';
      *p++ = '0';
    } else if (job.kind == R3D_TEXT_PLY_ROWS) {
      *p++ = ' ';
    }
    *p++ = '\n';
  }
  __syncthreads();
  char* const dst = text + (off - shift);
  const uint32_t end = shift + total;
  for (uint32_t lo = threadIdx.x * 16u; lo < end; lo += kTile * 16u) {
    if (lo >= shift && lo + 16 <= end) {
      *reinterpret_cast<uint4*>(dst + lo) = *reinterpret_cast<const uint4*>(sh_text + lo);
    } else {   // the tile's first and last 16 bytes are shared with its neighbours: byte stores
      const uint32_t a0 = lo < shift ? shift : lo, a1 = lo + 16 < end ? lo + 16 : end;
      for (uint32_t j = a0; j < a1; ++j) dst[j] = sh_text[j];
    }
  }
}

// tile_off[t] = sum of tile_len[0..t); tile_off[n] = the text's size.  One workgroup of 1024 lanes, 8 tiles per lane and round.
__global__ __launch_bounds__(1024) void text_scan_kernel(const uint32_t* __restrict__ tile_len, uint64_t* __restrict__ tile_off, int64_t n) {
  __shared__ uint64_t sh[16];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t base = 0; base < n; base += 1024 * kScanPer) {
    uint32_t v[kScanPer];
    uint64_t mine = 0;
    const int64_t first = base + (int64_t)threadIdx.x * kScanPer;
    for (int k = 0; k < kScanPer; ++k) {
      v[k] = first + k < n ? tile_len[first + k] : 0;
      mine += v[k];
    }
    uint64_t inc = mine;
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t up = __shfl_up(inc, d, 64);
      if (lane >= d) inc += up;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint64_t before = carry;
    for (int w = 0; w < wave; ++w) before += sh[w];
    uint64_t run = before + inc - mine;
    for (int k = 0; k < kScanPer; ++k) {
      if (first + k < n) tile_off[first + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = run;
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_off[n] = carry;
}

std::mutex g_table_mutex;
bool g_table_on_device[64] = {};

int upload_table(r3d_ctx* ctx) {
  std::lock_guard<std::mutex> lock(g_table_mutex);
  if (ctx->device >= 0 && ctx->device < 64 && g_table_on_device[ctx->device]) return R3D_OK;
  constexpr int n = r3d_pow10::kMax - r3d_pow10::kMin + 1;
  std::vector<uint64_t> flat((size_t)n * 2);
  for (int k = 0; k < n; ++k) {
    flat[2 * k] = r3d_pow10::kG[k].hi;
    flat[2 * k + 1] = r3d_pow10::kG[k].lo;
  }
  R3D_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pow10), flat.data(), flat.size() * 8, 0, hipMemcpyHostToDevice));
  if (ctx->device >= 0 && ctx->device < 64) g_table_on_device[ctx->device] = true;
  return R3D_OK;
}

}  // namespace

extern "C" {

int r3d_format_text_device(r3d_ctx* ctx, int kind, const void* d_xyz, int dtype, int64_t n_points, const void* d_aux, int aux_dtype,
                           int64_t segment_points, char* d_text, size_t text_cap, int64_t* h_segment_offsets_out,
                           int64_t* n_bytes_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(kind == R3D_TEXT_XYZ_TXT || kind == R3D_TEXT_PLY_ROWS || kind == R3D_TEXT_PLY_ROWS_RGB, "unknown text kind %d", kind);
  R3D_REQUIRE(dtype == R3D_F32 || dtype == R3D_F64, "unknown point dtype %d", dtype);
  R3D_REQUIRE(n_points >= 0 && segment_points >= 0, "negative size");
  R3D_REQUIRE(n_bytes_out != nullptr, "n_bytes_out is NULL");
  *n_bytes_out = 0;
  int aux_stride = 0;
  if (kind == R3D_TEXT_XYZ_TXT && d_aux) {
    R3D_REQUIRE(aux_dtype == R3D_DEPTH_U8 || aux_dtype == R3D_DEPTH_U16, "the integer third column is u8 or u16");
    aux_stride = aux_dtype == R3D_DEPTH_U8 ? 1 : 2;
  } else if (kind == R3D_TEXT_PLY_ROWS_RGB) {
    R3D_REQUIRE(n_points == 0 || d_aux != nullptr, "colours are NULL");
    R3D_REQUIRE(aux_dtype == 3 || aux_dtype == 4, "colours are 3 (r,g,b) or 4 (rgba words) bytes apart");
    aux_stride = aux_dtype;
  } else {
    R3D_REQUIRE(d_aux == nullptr, "this kind takes no second array");
  }
  const int64_t seg = segment_points > 0 ? segment_points : std::max<int64_t>(n_points, 1);
  const int64_t n_seg = n_points ? (n_points + seg - 1) / seg : 0;
  if (n_points == 0) {
    if (h_segment_offsets_out) h_segment_offsets_out[0] = 0;
    return R3D_OK;
  }
  R3D_REQUIRE(d_xyz != nullptr, "NULL device pointer");
  const int64_t tps = (seg + kTile - 1) / kTile;
  R3D_REQUIRE(tps <= 0x7fffffff && n_seg * tps <= 0x7fffffff, "too many tiles for one launch (%lld points)", (long long)n_points);
  const int64_t n_tiles = n_seg * tps;
  if ((rc = upload_table(ctx))) return rc;
  // workspace: flags (16 B) | tile_off [n_tiles + 1] u64 | tile_len [n_tiles] u32
  void* ws = nullptr;
  const size_t off_bytes = (size_t)(n_tiles + 1) * 8;
  if ((rc = r3d_scratch(ctx, 5, 16 + off_bytes + (size_t)n_tiles * 4, &ws))) return rc;
  unsigned int* d_flags = static_cast<unsigned int*>(ws);
  uint64_t* d_off = reinterpret_cast<uint64_t*>(static_cast<char*>(ws) + 16);
  uint32_t* d_len = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + 16 + off_bytes);
  R3D_HIP(hipMemsetAsync(d_flags, 0, 16, ctx->stream));
  TextJob job{d_xyz, d_aux, n_points, seg, (int)tps, kind, dtype == R3D_F64, aux_stride};
  if (dtype == R3D_F64)
    hipLaunchKernelGGL((text_rows_kernel<false, true>), dim3((unsigned)n_tiles), dim3(kTile), 0, ctx->stream, job, d_len, nullptr, nullptr, d_flags);
  else
    hipLaunchKernelGGL((text_rows_kernel<false, false>), dim3((unsigned)n_tiles), dim3(kTile), 0, ctx->stream, job, d_len, nullptr, nullptr, d_flags);
  hipLaunchKernelGGL(text_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_len, d_off, n_tiles);
  R3D_HIP(hipGetLastError());
  // the offsets come to the host: the size, the segment starts (and the flag word in front of them)
  std::vector<uint64_t> h((size_t)n_tiles + 3);
  R3D_HIP(hipMemcpyAsync(h.data(), ws, 16 + off_bytes, hipMemcpyDeviceToHost, ctx->stream));
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  const unsigned int flags = (unsigned int)(h[0] & 0xffffffffu);
  const uint64_t* off = h.data() + 2;
  if (flags & 1u) {
    r3d_set_error("r3d_format_text_device: a coordinate of magnitude >= 2^40 in a %%.4f row (more digits than a device row holds); "
                  "format this cloud on the host (r3d_format_ply / r3d_write_ply)");
    return R3D_ERR_UNSUPPORTED;
  }
  *n_bytes_out = (int64_t)off[n_tiles];
  if (h_segment_offsets_out) {
    for (int64_t s = 0; s < n_seg; ++s) h_segment_offsets_out[s] = (int64_t)off[s * tps];
    h_segment_offsets_out[n_seg] = (int64_t)off[n_tiles];
  }
  if (!d_text) return R3D_OK;   // size query
  if (text_cap < off[n_tiles]) {
    r3d_set_error("r3d_format_text_device: text buffer of %zu bytes is too small for %llu", text_cap, (unsigned long long)off[n_tiles]);
    return R3D_ERR_NOMEM;
  }
  r3d_wrote(ctx, d_text, (size_t)off[n_tiles]);
  if (dtype == R3D_F64)
    hipLaunchKernelGGL((text_rows_kernel<true, true>), dim3((unsigned)n_tiles), dim3(kTile), 0, ctx->stream, job, nullptr, d_off, d_text, d_flags);
  else
    hipLaunchKernelGGL((text_rows_kernel<true, false>), dim3((unsigned)n_tiles), dim3(kTile), 0, ctx->stream, job, nullptr, d_off, d_text, d_flags);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

// Device text -> files.  Every file is one job: its head (host bytes), a range of the device text, its tail.  A pool of host
// threads takes the jobs largest first; a worker brings its range over in 1 MiB pieces through two pinned buffers and a
// stream of its own -- piece k+1 crosses PCIe while piece k goes into the file with write().  One file is ONE sequential
// write stream (tmpfs takes 6.2 GB/s from one stream and less from several on the same inode), different files run side by
// side: the fused PLY beside the per-frame camera txts.
int r3d_write_device_text_files(r3d_ctx* ctx, const r3d_text_file* files, int n_files) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_files >= 0 && (n_files == 0 || files != nullptr), "bad file list");
  if (n_files == 0) return R3D_OK;
  for (int k = 0; k < n_files; ++k) {
    R3D_REQUIRE(files[k].path != nullptr, "path %d is NULL", k);
    R3D_REQUIRE(files[k].n_bytes == 0 || files[k].d_bytes != nullptr, "device bytes of file %d are NULL", k);
    R3D_REQUIRE((files[k].head_bytes == 0 || files[k].head) && (files[k].tail_bytes == 0 || files[k].tail), "head / tail of file %d is NULL", k);
  }
  R3D_HIP(hipStreamSynchronize(ctx->stream));   // the text is complete
  std::vector<int> order((size_t)n_files);
  for (int k = 0; k < n_files; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return files[a].n_bytes > files[b].n_bytes; });
  // How many writers: the largest file is one stream (~6 GB/s into tmpfs, measured on the MI355X box's host; several streams on
  // ONE inode are slower, an mmap half as fast: tools/tmpfs_write_probe.cpp) and sets the finishing time; the other files need
  // about (rest / largest) streams to be done by then, plus one to spare.  More than that only competes with the critical
  // stream for the process's CPU quota: C2's 3.29 GB took 298 / 269 / 242 / 227 ms with 32 / 16 / 8 / 4 writers.
  size_t all_bytes = 0;
  for (int k = 0; k < n_files; ++k) all_bytes += files[k].n_bytes;
  const size_t big_bytes = std::max<size_t>(files[order[0]].n_bytes, 1);
  const unsigned want_workers = (unsigned)std::min<size_t>(2 + (all_bytes - files[order[0]].n_bytes + big_bytes - 1) / big_bytes,
                                                            std::max(1u, r3d_host::cpu_budget() / 2));
  const unsigned n_workers = std::max(1u, std::min<unsigned>(want_workers, (unsigned)n_files));
  constexpr size_t kPiece = (size_t)1 << 20;
  char* pinned = nullptr;
  R3D_HIP(hipHostMalloc(reinterpret_cast<void**>(&pinned), (size_t)n_workers * 2 * kPiece, hipHostMallocDefault));
  std::atomic<int> next{0}, first_rc{R3D_OK};
  std::mutex msg_mutex;
  std::string first_msg;
  auto fail = [&](int code, const std::string& msg) {
    int expected = R3D_OK;
    if (first_rc.compare_exchange_strong(expected, code)) {
      std::lock_guard<std::mutex> lock(msg_mutex);
      first_msg = msg;
    }
  };
  const int device = ctx->device;
  auto worker = [&](unsigned w) {
    if (hipSetDevice(device) != hipSuccess) return fail(R3D_ERR_HIP, "hipSetDevice failed in a writer thread");
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) {
      fail(R3D_ERR_HIP, "no stream / event for a writer thread");
    } else {
      char* buf[2] = {pinned + (size_t)w * 2 * kPiece, pinned + (size_t)w * 2 * kPiece + kPiece};
      for (;;) {
        const int j = next.fetch_add(1);
        if (j >= n_files || first_rc.load() != R3D_OK) break;
        const r3d_text_file& f = files[order[(size_t)j]];
        const int fd = open(f.path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) {
          fail(R3D_ERR_INVALID, std::string("r3d_write_device_text_files: cannot open '") + f.path + "' for writing");
          break;
        }
        auto put = [&](const char* p, size_t n) {
          while (n) {
            const ssize_t k = write(fd, p, n);
            if (k <= 0) return false;
            p += k;
            n -= (size_t)k;
          }
          return true;
        };
        bool ok = put(f.head, f.head_bytes);
        const size_t n_pieces = (f.n_bytes + kPiece - 1) / kPiece;
        const char* d_src = static_cast<const char*>(f.d_bytes);
        auto fetch = [&](size_t c) {
          const size_t lo = c * kPiece, n = std::min(kPiece, f.n_bytes - lo);
          return hipMemcpyAsync(buf[c & 1], d_src + lo, n, hipMemcpyDeviceToHost, st) == hipSuccess &&
                 hipEventRecord(ev[c & 1], st) == hipSuccess;
        };
        bool hip_ok = n_pieces == 0 || fetch(0);
        for (size_t c = 0; ok && hip_ok && c < n_pieces; ++c) {
          if (c + 1 < n_pieces) hip_ok = fetch(c + 1);
          hip_ok = hip_ok && hipEventSynchronize(ev[c & 1]) == hipSuccess;
          if (hip_ok) ok = put(buf[c & 1], std::min(kPiece, f.n_bytes - c * kPiece));
        }
        (void)hipStreamSynchronize(st);   // nothing of this file is in flight into the buffers any more
        ok = ok && hip_ok && put(f.tail, f.tail_bytes);
        if (close(fd) != 0) ok = false;
        if (!hip_ok)
          fail(R3D_ERR_HIP, std::string("r3d_write_device_text_files: a device-to-host copy failed for '") + f.path + "'");
        else if (!ok)
          fail(R3D_ERR_INVALID, std::string("r3d_write_device_text_files: short write to '") + f.path + "'");
      }
    }
    if (ev[0]) (void)hipEventDestroy(ev[0]);
    if (ev[1]) (void)hipEventDestroy(ev[1]);
    if (st) (void)hipStreamDestroy(st);
  };
  {
    std::vector<std::thread> pool;
    pool.reserve(n_workers);
    for (unsigned w = 1; w < n_workers; ++w) {
      try {
        pool.emplace_back(worker, w);
      } catch (const std::system_error&) {   // fewer workers, same files
        break;
      }
    }
    worker(0);
    for (auto& t : pool) t.join();
  }
  (void)hipHostFree(pinned);
  if (first_rc.load() != R3D_OK) {
    r3d_set_error("%s", first_msg.c_str());
    return first_rc.load();
  }
  return R3D_OK;
}

}  // extern "C"
