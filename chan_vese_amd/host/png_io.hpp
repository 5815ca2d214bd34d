// png_io.hpp — minimal PNG reader/writer over zlib for bin/chan_vese (no libpng/OpenCV in this image).
//
// Stands in for cv::imread / cv::imwrite on .png files (src/main.cpp:877-887, :946, :1005).
// Reader: non-interlaced PNG, colour types 0/2/3/4/6, bit depths 1..16, delivered as 8-bit gray
// or 8-bit RGB the way OpenCV 2.4's PngDecoder sets libpng up for an 8-bit imread: 16-bit samples keep
// their high byte (png_set_strip_16), 1/2/4-bit gray and palettes are expanded, alpha is dropped
// (png_set_strip_alpha).  Interlaced files are refused.
// Writer: 8-bit gray or RGB, filter 0, one IDAT.
// Parity unpinned: neither libpng nor OpenCV exists here; tests cross-check against Python's zlib.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace pngio {

struct Decoded {
  int h = 0, w = 0, channels = 0;  // channels: 1 (gray) or 3 (RGB interleaved)
  std::vector<uint8_t> px;
};

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline bool is_png(const std::string &path)
{
  std::ifstream in(path, std::ios::binary);
  uint8_t sig[8] = {0};
  in.read(reinterpret_cast<char *>(sig), 8);
  static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  return in.gcount() == 8 && !std::memcmp(sig, kSig, 8);
}

inline int paeth(int a, int b, int c)
{
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// returns an empty string on success, else what went wrong
inline std::string read(const std::string &path, Decoded &out)
{
  std::ifstream in(path, std::ios::binary);
  if (!in) return "cannot open";
  std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (file.size() < 8 + 25 || !is_png(path)) return "not a PNG file";
  size_t pos = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool have_ihdr = false, have_iend = false;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    if (len > file.size() - pos - 12) return "truncated chunk";
    const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
    const uint32_t crc = be32(&file[pos + 8 + len]);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != crc) return "chunk CRC mismatch";
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return "bad IHDR";
      w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0) return "unknown compression/filter method";
      have_ihdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      have_iend = true;
      break;
    }
    pos += 12 + len;
  }
  if (!have_ihdr || !have_iend || idat.empty()) return "missing IHDR/IDAT/IEND";
  if (w == 0 || h == 0 || w > (1u << 30) / 8 || h > (1u << 30) / 8) return "bad dimensions";
  if (interlace != 0) return "interlaced PNG is not supported";
  int samples;
  switch (ctype) {
    case 0: samples = 1; break;
    case 2: samples = 3; break;
    case 3: samples = 1; break;
    case 4: samples = 2; break;
    case 6: samples = 4; break;
    default: return "bad colour type";
  }
  const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                        (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                        ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
  if (!depth_ok) return "bad bit depth";
  if (ctype == 3 && (plte.empty() || plte.size() % 3)) return "missing palette";
  const size_t bits_pp = (size_t)samples * depth, stride = ((size_t)w * bits_pp + 7) / 8, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
  std::vector<uint8_t> raw((stride + 1) * (size_t)h);
  {
    uLongf dlen = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &dlen, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || dlen != raw.size()) return "inflate failed";
  }
  // unfilter in place (PNG specification, section 9)
  std::vector<uint8_t> zero(stride, 0);
  for (size_t y = 0; y < h; ++y) {
    uint8_t *cur = &raw[y * (stride + 1) + 1];
    const uint8_t *up = y ? &raw[(y - 1) * (stride + 1) + 1] : zero.data();
    const int ft = raw[y * (stride + 1)];
    if (ft < 0 || ft > 4) return "bad filter type";
    for (size_t x = 0; x < stride; ++x) {
      const int a = x >= bpp ? cur[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
      int pred = 0;
      if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) >> 1; else if (ft == 4) pred = paeth(a, b, c);
      cur[x] = (uint8_t)(cur[x] + pred);
    }
  }
  out.h = (int)h; out.w = (int)w; out.channels = (ctype == 0 || ctype == 4) ? 1 : 3;
  out.px.resize((size_t)h * w * out.channels);
  for (size_t y = 0; y < h; ++y) {
    const uint8_t *row = &raw[y * (stride + 1) + 1];
    uint8_t *dst = &out.px[y * (size_t)w * out.channels];
    for (size_t x = 0; x < w; ++x) {
      auto sample = [&](int k) -> int {  // k-th sample of pixel x as 8 bits
        if (depth == 8) return row[x * samples + k];
        if (depth == 16) return row[(x * samples + k) * 2];  // high byte
        const size_t bit = x * depth;                          // depth < 8: one sample per pixel
        const int v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
      };
      if (ctype == 0 || ctype == 4) dst[x] = (uint8_t)sample(0);
      else if (ctype == 3) {
        const size_t idx = (size_t)sample(0);
        if (idx * 3 + 2 >= plte.size()) return "palette index out of range";
        dst[3 * x] = plte[3 * idx]; dst[3 * x + 1] = plte[3 * idx + 1]; dst[3 * x + 2] = plte[3 * idx + 2];
      } else { dst[3 * x] = (uint8_t)sample(0); dst[3 * x + 1] = (uint8_t)sample(1); dst[3 * x + 2] = (uint8_t)sample(2); }
    }
  }
  return "";
}

inline void put_chunk(std::vector<uint8_t> &f, const char *type, const uint8_t *data, size_t len)
{
  const uint32_t l = (uint32_t)len;
  const uint8_t hdr[8] = {(uint8_t)(l >> 24), (uint8_t)(l >> 16), (uint8_t)(l >> 8), (uint8_t)l,
                          (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
  f.insert(f.end(), hdr, hdr + 8);
  if (len) f.insert(f.end(), data, data + len);
  uLong crc = crc32(0L, Z_NULL, 0);
  crc = crc32(crc, hdr + 4, 4);
  if (len) crc = crc32(crc, data, (uInt)len);
  const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
  f.insert(f.end(), c, c + 4);
}

// px: h*w*channels, channels 1 (gray) or 3 (RGB interleaved)
inline bool write(const std::string &path, int h, int w, int channels, const uint8_t *px)
{
  if (h <= 0 || w <= 0 || (channels != 1 && channels != 3)) return false;
  const size_t stride = (size_t)w * channels;
  std::vector<uint8_t> raw((stride + 1) * (size_t)h);
  for (int y = 0; y < h; ++y) {
    raw[(size_t)y * (stride + 1)] = 0;
    std::memcpy(&raw[(size_t)y * (stride + 1) + 1], px + (size_t)y * stride, stride);
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 3) != Z_OK) return false;
  std::vector<uint8_t> f = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                            (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                            8, (uint8_t)(channels == 1 ? 0 : 2), 0, 0, 0};
  put_chunk(f, "IHDR", ihdr, 13);
  put_chunk(f, "IDAT", comp.data(), clen);
  put_chunk(f, "IEND", nullptr, 0);
  std::ofstream out(path, std::ios::binary);
  if (!out) return false;
  out.write(reinterpret_cast<const char *>(f.data()), (std::streamsize)f.size());
  return (bool)out;
}

}  // namespace pngio
