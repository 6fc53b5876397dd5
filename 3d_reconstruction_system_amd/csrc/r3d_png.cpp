// f3 ingestion: PNG -> raster, on host threads (zlib inflate + PNG unfiltering): grey depth rasters and, for the RGBD
// path (config 5; colour attach of pixel_to_camera.py:55-91), 8-bit colour images as R,G,B bytes.
//
// Replaces `cv.imread('./depth/'+name, cv.IMREAD_GRAYSCALE)` of camera_to_world.py:160 for the files that path
// actually reads: non-interlaced greyscale PNGs (colour type 0) of 8 bits (-> uint8, identical to OpenCV) or
// 16 bits (-> uint16 big-endian samples converted to host order; OpenCV would reduce these to 8 bits, the caller
// decides), and 8-bit RGB(A) / grey+alpha files whose colour channels agree in every pixel (a depth map saved "in
// colour": its grey value is the same under every conversion rule).  Anything else (real colour, palette, interlace)
// returns R3D_ERR_UNSUPPORTED so the Python host can fall back to cv2/PIL, whose colour->grey conversions are theirs
// to define.
// A batch of files is decoded straight into one [n][H][W] buffer (e.g. pinned memory) by a thread pool.
#include <zlib.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "r3d.h"
#include "r3d_hostpool.h"

namespace {

struct PngInfo {
  uint32_t width = 0, height = 0;
  int bit_depth = 0, colour_type = 0, interlace = 0;
  bool gamma_tagged = false;   // a gAMA / sRGB / iCCP chunk in front of the image data: libpng then converts colour to grey in linear light
};

uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// reads the file, returns concatenated IDAT payload; rc: 0 ok, else R3D_ERR_*; msg set
int read_png(const char* path, PngInfo* info, std::vector<unsigned char>* idat, std::string* msg) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    *msg = std::string("cannot open '") + path + "'";
    return R3D_ERR_INVALID;
  }
  std::vector<unsigned char>& file = r3d_host::scratch(0);
  {
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
  }
  fclose(f);
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) {
    *msg = std::string("'") + path + "' is not a PNG file";
    return R3D_ERR_INVALID;
  }
  size_t pos = 8;
  bool have_ihdr = false;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const unsigned char* type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) break;
    const unsigned char* data = &file[pos + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      info->width = be32(data);
      info->height = be32(data + 4);
      info->bit_depth = data[8];
      info->colour_type = data[9];
      info->interlace = data[12];
      have_ihdr = true;
    } else if (!memcmp(type, "gAMA", 4) || !memcmp(type, "sRGB", 4) || !memcmp(type, "iCCP", 4)) {
      if (idat->empty()) info->gamma_tagged = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      idat->insert(idat->end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || idat->empty()) {
    *msg = std::string("'") + path + "': truncated PNG";
    return R3D_ERR_INVALID;
  }
  return R3D_OK;
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// One scanline's filter undone (PNG spec 9.2): row = filtered + predictor(left, up, upper left), bytes `bpp` apart; the row above
// is all zeros for the first line.  One loop per filter type (the type is a property of the line, not of the byte): Up is a
// plain vector add, Sub / Average / Paeth carry a dependency from `bpp` bytes back.
inline bool unfilter_row(int filter, const unsigned char* __restrict src, unsigned char* __restrict row,
                         const unsigned char* __restrict up, size_t stride, size_t bpp) {
  switch (filter) {
    case 0:
      memcpy(row, src, stride);
      return true;
    case 1:
      for (size_t x = 0; x < bpp && x < stride; ++x) row[x] = src[x];
      for (size_t x = bpp; x < stride; ++x) row[x] = (unsigned char)(src[x] + row[x - bpp]);
      return true;
    case 2:
      if (!up) {
        memcpy(row, src, stride);
      } else {
        for (size_t x = 0; x < stride; ++x) row[x] = (unsigned char)(src[x] + up[x]);
      }
      return true;
    case 3:
      for (size_t x = 0; x < stride; ++x) {
        const int a = x >= bpp ? row[x - bpp] : 0, b = up ? up[x] : 0;
        row[x] = (unsigned char)(src[x] + ((a + b) >> 1));
      }
      return true;
    case 4:
      if (!up) {   // predictor = left
        for (size_t x = 0; x < bpp && x < stride; ++x) row[x] = src[x];
        for (size_t x = bpp; x < stride; ++x) row[x] = (unsigned char)(src[x] + row[x - bpp]);
      } else {
        for (size_t x = 0; x < bpp && x < stride; ++x) row[x] = (unsigned char)(src[x] + up[x]);   // paeth(0, b, 0) = b
        for (size_t x = bpp; x < stride; ++x) row[x] = (unsigned char)(src[x] + paeth(row[x - bpp], up[x], up[x - bpp]));
      }
      return true;
    default:
      return false;
  }
}

// Inflate + unfilter one non-interlaced PNG: `pixels` receives height rows of `stride` bytes (no filter bytes).
// channels_out = samples per pixel in the file (1 grey, 2 grey+alpha, 3 RGB, 4 RGBA).  Palette / interlace / sub-byte
// depths -> R3D_ERR_UNSUPPORTED.  header_only: stop after IHDR checks.
int decode_png(const char* path, bool header_only, PngInfo* info, int* channels_out, std::vector<unsigned char>* pixels,
               std::string* msg) {
  r3d_host::ScratchScope scope;
  std::vector<unsigned char>& idat = r3d_host::scratch(1);
  int rc = read_png(path, info, &idat, msg);
  if (rc) return rc;
  if (info->width == 0 || info->height == 0 || info->width > (1u << 20) || info->height > (1u << 20)) {
    *msg = std::string("'") + path + "': implausible PNG dimensions";
    return R3D_ERR_INVALID;
  }
  int channels = 0;
  switch (info->colour_type) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: channels = 0; break;  // 3 = palette
  }
  if (channels == 0 || info->interlace != 0 || (info->bit_depth != 8 && info->bit_depth != 16)) {
    *msg = std::string("'") + path + "': only non-interlaced 8/16-bit grey / RGB(A) PNGs are decoded natively";
    return R3D_ERR_UNSUPPORTED;
  }
  *channels_out = channels;
  if (header_only) return R3D_OK;
  const size_t bpp = (size_t)channels * info->bit_depth / 8, stride = (size_t)info->width * bpp;
  std::vector<unsigned char>& raw = r3d_host::scratch(2);
  raw.resize((stride + 1) * info->height);
  uLongf raw_len = (uLongf)raw.size();
  const int z = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
  if (z != Z_OK || raw_len != raw.size()) {
    *msg = std::string("'") + path + "': zlib inflate failed";
    return R3D_ERR_INVALID;
  }
  pixels->resize(stride * info->height);
  unsigned char* dst = pixels->data();
  for (uint32_t y = 0; y < info->height; ++y) {
    const unsigned char* src = &raw[(stride + 1) * y];
    unsigned char* row = dst + stride * y;
    if (!unfilter_row(src[0], src + 1, row, y ? row - stride : nullptr, stride, bpp)) {
      *msg = std::string("'") + path + "': bad PNG filter type";
      return R3D_ERR_INVALID;
    }
  }
  return R3D_OK;
}

// one grey PNG into out (row-major, H*W samples of bits/8 bytes, host byte order); out == NULL: header query
int decode_gray_impl(const char* path, void* out, size_t cap_bytes, int* h_out, int* w_out, int* bits_out, std::string* msg) {
  PngInfo info;
  int channels = 0;
  r3d_host::ScratchScope scope;
  std::vector<unsigned char>& px = r3d_host::scratch(3);
  // one pass over the file: the pixels are decoded only when there is somewhere to put them and the file is grey
  int rc = decode_png(path, out == nullptr, &info, &channels, &px, msg);
  if (h_out) *h_out = (int)info.height;
  if (w_out) *w_out = (int)info.width;
  if (bits_out) *bits_out = info.bit_depth;
  if (rc) return rc;
  if (channels != 1 && info.bit_depth != 8) {
    *msg = std::string("'") + path + "': not a greyscale PNG (the colour -> grey conversion is the caller's to define)";
    return R3D_ERR_UNSUPPORTED;
  }
  if (!out) return R3D_OK;
  const size_t need = (size_t)info.width * info.height * (info.bit_depth / 8);
  if (cap_bytes < need) {
    *msg = std::string("'") + path + "': output buffer too small";
    return R3D_ERR_NOMEM;
  }
  unsigned char* dst = static_cast<unsigned char*>(out);
  if (channels != 1) {
    // 8-bit RGB / RGBA / grey+alpha: a depth map saved "in colour" has R = G = B in every pixel, and such a pixel IS its
    // grey value under every colour -> grey rule there is (OpenCV's, libpng's, PIL's); alpha is dropped by all of them.
    // One pixel with differing channels and the conversion is the caller's to define.
    const unsigned char* s = px.data();
    for (size_t k = 0; k < need; ++k, s += channels) {
      if (channels >= 3 && (s[0] != s[1] || s[1] != s[2])) {
        *msg = std::string("'") + path + "': a colour PNG whose channels differ (the colour -> grey conversion is the caller's to define)";
        return R3D_ERR_UNSUPPORTED;
      }
      dst[k] = s[0];
    }
  } else if (info.bit_depth == 8) {
    memcpy(dst, px.data(), need);
  } else {  // PNG samples are big-endian
    for (size_t k = 0; k < need; k += 2) {
      dst[k] = px[k + 1];
      dst[k + 1] = px[k];
    }
  }
  return R3D_OK;
}

int decode_gray(const char* path, void* out, size_t cap_bytes, int* h_out, int* w_out, int* bits_out, std::string* msg) {
  try {
    return decode_gray_impl(path, out, cap_bytes, h_out, w_out, bits_out, msg);
  } catch (const std::exception& e) {
    try {
      *msg = std::string("'") + path + "': " + e.what();
    } catch (...) {
    }
    return R3D_ERR_NOMEM;
  }
}

// ---- colour -> grey, the two integer rules OpenCV has ---------------------------------------------------------------
// R3D_GRAY_OPENCV_PNG: what cv.imread(<png>, IMREAD_GRAYSCALE) executes (camera_to_world.py:160).  OpenCV's PNG reader
//   (modules/imgcodecs/src/grfmt_png.cpp, the pinned 4.2.0 included) does not call cvtColor: it asks libpng for grey,
//   png_set_rgb_to_gray(png_ptr, 1, 0.299, 0.587).  libpng turns the weights into 15-bit integers WITHOUT rounding
//   (29900 * 32768 / 100000 = 9797, 58700 * 32768 / 100000 = 19234, blue = 32768 - 9797 - 19234 = 3737) and, for 8-bit
//   samples of a file without gamma information, truncates: (9797 R + 19234 G + 3737 B) >> 15; a pixel whose channels
//   agree keeps its value; 16-bit samples get (... + 16384) >> 15 and are then reduced to their high byte (strip_16 runs
//   after rgb_to_gray in libpng's transformation order).
// R3D_GRAY_CVTCOLOR: cv.cvtColor(..., COLOR_BGR2GRAY) on 8-bit data, which imread applies to the decoders that deliver
//   colour (BMP, TIFF, WebP ...): (4899 R + 9617 G + 1868 B + 8192) >> 14.
// Neither equals PIL's 'L' ((19595 R + 38470 G + 7471 B + 32768) >> 16).  cv2 is not in this image, so the rules are
// restated from the two libraries' sources and pinned by known-answer vectors worked out by hand (tests/test_host_logic.py).
inline unsigned gray8(unsigned r, unsigned g, unsigned b, int rule) {
  if (rule == R3D_GRAY_CVTCOLOR) return (4899u * r + 9617u * g + 1868u * b + 8192u) >> 14;
  return (r == g && r == b) ? r : (9797u * r + 19234u * g + 3737u * b) >> 15;
}
inline unsigned gray16(unsigned r, unsigned g, unsigned b) {   // libpng, 16-bit samples
  return (r == g && r == b) ? r : (9797u * r + 19234u * g + 3737u * b + 16384u) >> 15;
}

// A colour pixel in a file that carries gamma information: libpng (hence OpenCV's PNG reader) builds gamma tables and converts
// in LINEAR light -- gamma_from_1[(9797 to_1[R] + 19234 to_1[G] + 3737 to_1[B] + 16384) >> 15] with tables from its own
// fixed-point pow -- which is not restated here.  Refused, not approximated; pixels with R = G = B pass through unchanged
// there too (no overall gamma correction is requested), so depth maps saved "in colour" by such tools still decode.
int gamma_refusal(const char* path, std::string* msg) {
  *msg = std::string("'") + path + "': a colour PNG with a gAMA / sRGB / iCCP chunk -- OpenCV (libpng) converts such files to grey in linear "
         "light, which this build does not restate; strip the chunk (e.g. re-save with PIL) or ask for rule R3D_GRAY_CVTCOLOR explicitly";
  return R3D_ERR_UNSUPPORTED;
}

// one PNG as the uint8 raster cv.imread(path, IMREAD_GRAYSCALE) returns; out == NULL: header query
int decode_gray8_impl(const char* path, unsigned char* out, size_t cap_bytes, int rule, int* h_out, int* w_out, std::string* msg) {
  PngInfo info;
  int channels = 0;
  r3d_host::ScratchScope scope;
  std::vector<unsigned char>& px = r3d_host::scratch(3);
  int rc = decode_png(path, out == nullptr, &info, &channels, &px, msg);
  if (h_out) *h_out = (int)info.height;
  if (w_out) *w_out = (int)info.width;
  if (rc || !out) return rc;
  const size_t n = (size_t)info.width * info.height;
  if (cap_bytes < n) {
    *msg = std::string("'") + path + "': output buffer too small";
    return R3D_ERR_NOMEM;
  }
  const unsigned char* s = px.data();
  if (info.bit_depth == 8) {
    if (channels == 1) {
      memcpy(out, s, n);
    } else if (channels == 2) {
      for (size_t k = 0; k < n; ++k) out[k] = s[2 * k];                       // alpha is stripped
    } else {
      for (size_t k = 0; k < n; ++k, s += channels) {
        if (info.gamma_tagged && rule == R3D_GRAY_OPENCV_PNG && (s[0] != s[1] || s[1] != s[2])) return gamma_refusal(path, msg);
        out[k] = (unsigned char)gray8(s[0], s[1], s[2], rule);
      }
    }
  } else {  // 16-bit big-endian samples
    const size_t px_bytes = (size_t)channels * 2;
    if (channels <= 2) {
      for (size_t k = 0; k < n; ++k) out[k] = s[px_bytes * k];                 // the high byte (png_set_strip_16)
    } else {
      for (size_t k = 0; k < n; ++k, s += px_bytes) {
        const unsigned r = (s[0] << 8) | s[1], g = (s[2] << 8) | s[3], b = (s[4] << 8) | s[5];
        if (info.gamma_tagged && rule == R3D_GRAY_OPENCV_PNG && (r != g || g != b)) return gamma_refusal(path, msg);
        out[k] = rule == R3D_GRAY_CVTCOLOR ? (unsigned char)gray8(r >> 8, g >> 8, b >> 8, rule) : (unsigned char)(gray16(r, g, b) >> 8);
      }
    }
  }
  return R3D_OK;
}

int decode_gray8(const char* path, unsigned char* out, size_t cap_bytes, int rule, int* h_out, int* w_out, std::string* msg) {
  try {
    return decode_gray8_impl(path, out, cap_bytes, rule, h_out, w_out, msg);
  } catch (const std::exception& e) {
    try {
      *msg = std::string("'") + path + "': " + e.what();
    } catch (...) {
    }
    return R3D_ERR_NOMEM;
  }
}

// one 8-bit PNG as R,G,B bytes ([H][W][3]); grey is replicated, alpha is dropped; out == NULL: header query
int decode_rgb(const char* path, unsigned char* out, size_t cap_bytes, int* h_out, int* w_out, int* channels_out, std::string* msg) {
  try {
    PngInfo info;
    int channels = 0;
    r3d_host::ScratchScope scope;
  std::vector<unsigned char>& px = r3d_host::scratch(3);
    int rc = decode_png(path, out == nullptr, &info, &channels, &px, msg);
    if (h_out) *h_out = (int)info.height;
    if (w_out) *w_out = (int)info.width;
    if (channels_out) *channels_out = channels;
    if (rc) return rc;
    if (info.bit_depth != 8) {
      *msg = std::string("'") + path + "': 16-bit samples are not a colour image this path reads";
      return R3D_ERR_UNSUPPORTED;
    }
    if (!out) return R3D_OK;
    const size_t n = (size_t)info.width * info.height;
    if (cap_bytes < n * 3) {
      *msg = std::string("'") + path + "': output buffer too small";
      return R3D_ERR_NOMEM;
    }
    const unsigned char* s = px.data();
    if (channels == 3) {
      memcpy(out, s, n * 3);
    } else if (channels == 4) {
      for (size_t k = 0; k < n; ++k) {
        out[3 * k] = s[4 * k];
        out[3 * k + 1] = s[4 * k + 1];
        out[3 * k + 2] = s[4 * k + 2];
      }
    } else {  // grey (+ alpha): replicate
      for (size_t k = 0; k < n; ++k) out[3 * k] = out[3 * k + 1] = out[3 * k + 2] = s[(size_t)channels * k];
    }
    return R3D_OK;
  } catch (const std::exception& e) {
    try {
      *msg = std::string("'") + path + "': " + e.what();
    } catch (...) {
    }
    return R3D_ERR_NOMEM;
  }
}

}  // namespace

extern "C" {

int r3d_png_gray_info(const char* path, int* height, int* width, int* bit_depth) {
  if (!path) {
    r3d_set_error("r3d_png_gray_info: path is NULL");
    return R3D_ERR_INVALID;
  }
  std::string msg;
  const int rc = decode_gray(path, nullptr, 0, height, width, bit_depth, &msg);
  if (rc) r3d_set_error("%s", msg.c_str());
  return rc;
}

int r3d_png_gray_decode_batch(const char* const* paths, int n_files, void* h_out, int height, int width, int bit_depth) {
  if (n_files < 0 || (n_files > 0 && (!paths || !h_out)) || height <= 0 || width <= 0 || (bit_depth != 8 && bit_depth != 16)) {
    r3d_set_error("r3d_png_gray_decode_batch: bad argument");
    return R3D_ERR_INVALID;
  }
  const size_t frame_bytes = (size_t)height * width * (bit_depth / 8);
  return r3d_host::run_batch(n_files, "PNG decode failed", [&](int k, std::string* msg) -> int {
    int h = 0, w = 0, bits = 0;
    int rc = paths[k] ? decode_gray(paths[k], static_cast<char*>(h_out) + frame_bytes * k, frame_bytes, &h, &w, &bits, msg)
                      : R3D_ERR_INVALID;
    if (rc == R3D_OK && (h != height || w != width || bits != bit_depth)) {
      rc = R3D_ERR_INVALID;
      *msg = std::string("'") + paths[k] + "' is " + std::to_string(w) + "x" + std::to_string(h) + "x" + std::to_string(bits) +
             " bits, the batch expects " + std::to_string(width) + "x" + std::to_string(height) + "x" + std::to_string(bit_depth);
    }
    return rc;
  });
}

int r3d_rgb_to_gray_u8(const unsigned char* pixels, int64_t n_pixels, int channels, int rule, unsigned char* gray_out) {
  if (n_pixels < 0 || (channels != 3 && channels != 4) || (rule != R3D_GRAY_OPENCV_PNG && rule != R3D_GRAY_CVTCOLOR) ||
      (n_pixels > 0 && (!pixels || !gray_out))) {
    r3d_set_error("r3d_rgb_to_gray_u8: bad argument (3 or 4 channels, rule R3D_GRAY_OPENCV_PNG or R3D_GRAY_CVTCOLOR)");
    return R3D_ERR_INVALID;
  }
  for (int64_t k = 0; k < n_pixels; ++k, pixels += channels) gray_out[k] = (unsigned char)gray8(pixels[0], pixels[1], pixels[2], rule);
  return R3D_OK;
}

int r3d_png_gray8_info(const char* path, int* height, int* width) {
  if (!path) {
    r3d_set_error("r3d_png_gray8_info: path is NULL");
    return R3D_ERR_INVALID;
  }
  std::string msg;
  const int rc = decode_gray8(path, nullptr, 0, R3D_GRAY_OPENCV_PNG, height, width, &msg);
  if (rc) r3d_set_error("%s", msg.c_str());
  return rc;
}

int r3d_png_gray8_decode_batch(const char* const* paths, int n_files, unsigned char* h_out, int height, int width, int rule) {
  if (n_files < 0 || (n_files > 0 && (!paths || !h_out)) || height <= 0 || width <= 0 ||
      (rule != R3D_GRAY_OPENCV_PNG && rule != R3D_GRAY_CVTCOLOR)) {
    r3d_set_error("r3d_png_gray8_decode_batch: bad argument");
    return R3D_ERR_INVALID;
  }
  const size_t frame_bytes = (size_t)height * width;
  return r3d_host::run_batch(n_files, "PNG decode failed", [&](int k, std::string* msg) -> int {
    int h = 0, w = 0;
    int rc = paths[k] ? decode_gray8(paths[k], h_out + frame_bytes * k, frame_bytes, rule, &h, &w, msg) : R3D_ERR_INVALID;
    if (rc == R3D_OK && (h != height || w != width)) {
      rc = R3D_ERR_INVALID;
      *msg = std::string("'") + paths[k] + "' is " + std::to_string(w) + "x" + std::to_string(h) + ", the batch expects " +
             std::to_string(width) + "x" + std::to_string(height);
    }
    return rc;
  });
}

int r3d_png_rgb_info(const char* path, int* height, int* width, int* channels) {
  if (!path) {
    r3d_set_error("r3d_png_rgb_info: path is NULL");
    return R3D_ERR_INVALID;
  }
  std::string msg;
  const int rc = decode_rgb(path, nullptr, 0, height, width, channels, &msg);
  if (rc) r3d_set_error("%s", msg.c_str());
  return rc;
}

int r3d_png_rgb_decode_batch(const char* const* paths, int n_files, unsigned char* h_out, int height, int width) {
  if (n_files < 0 || (n_files > 0 && (!paths || !h_out)) || height <= 0 || width <= 0) {
    r3d_set_error("r3d_png_rgb_decode_batch: bad argument");
    return R3D_ERR_INVALID;
  }
  const size_t frame_bytes = (size_t)height * width * 3;
  return r3d_host::run_batch(n_files, "PNG decode failed", [&](int k, std::string* msg) -> int {
    int h = 0, w = 0, ch = 0;
    int rc = paths[k] ? decode_rgb(paths[k], h_out + frame_bytes * k, frame_bytes, &h, &w, &ch, msg) : R3D_ERR_INVALID;
    if (rc == R3D_OK && (h != height || w != width)) {
      rc = R3D_ERR_INVALID;
      *msg = std::string("'") + paths[k] + "' is " + std::to_string(w) + "x" + std::to_string(h) + ", the batch expects " +
             std::to_string(width) + "x" + std::to_string(height);
    }
    return rc;
  });
}

}  // extern "C"
