// overlay_text.hpp — the -O overlay of the frame dump: VideoWriterManager::write_frame / overlay_color,
// src/VideoWriterManager.cpp:40-57, :76-114.  The reference draws "t = <iteration>" with cv::putText (Hershey plain,
// scale 0.8, thickness 1, anti-aliased) at one of four corners (padding 5 px) in black or white, chosen from the mean
// intensity of the INPUT image under the text box.  OpenCV's Hershey glyph tables are not available here, so the glyphs
// are an own 5x7 bitmap font (text box = 6 len - 1 by 7 pixels): the position rule (:88-107), the box the mean is
// taken over (:109), the luma weights (:110) and the black/white threshold (:80,:111) follow the reference; the glyph
// shapes (and therefore the exact box size) cannot be pinned offline.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace overlay {

enum class Pos { TL, BL, TR, BR };

// 5x7 glyphs, one byte per row, bit 4 = leftmost column
inline const uint8_t *glyph(char ch)
{
  static const uint8_t digits[10][7] = {
    {0x0E, 0x11, 0x13, 0x15, 0x19, 0x11, 0x0E}, {0x04, 0x0C, 0x04, 0x04, 0x04, 0x04, 0x0E}, {0x0E, 0x11, 0x01, 0x02, 0x04, 0x08, 0x1F},
    {0x1F, 0x02, 0x04, 0x02, 0x01, 0x11, 0x0E}, {0x02, 0x06, 0x0A, 0x12, 0x1F, 0x02, 0x02}, {0x1F, 0x10, 0x1E, 0x01, 0x01, 0x11, 0x0E},
    {0x06, 0x08, 0x10, 0x1E, 0x11, 0x11, 0x0E}, {0x1F, 0x01, 0x02, 0x04, 0x08, 0x08, 0x08}, {0x0E, 0x11, 0x11, 0x0E, 0x11, 0x11, 0x0E},
    {0x0E, 0x11, 0x11, 0x0F, 0x01, 0x02, 0x0C}};
  static const uint8_t t_[7] = {0x08, 0x08, 0x1C, 0x08, 0x08, 0x09, 0x06};
  static const uint8_t eq[7] = {0x00, 0x00, 0x1F, 0x00, 0x1F, 0x00, 0x00};
  static const uint8_t minus[7] = {0x00, 0x00, 0x00, 0x1F, 0x00, 0x00, 0x00};
  static const uint8_t blank[7] = {0, 0, 0, 0, 0, 0, 0};
  if (ch >= '0' && ch <= '9') return digits[ch - '0'];
  if (ch == 't') return t_;
  if (ch == '=') return eq;
  if (ch == '-') return minus;
  return blank;
}

inline void text_size(const std::string &txt, int *width, int *height)
{
  *width = txt.empty() ? 0 : 6 * (int)txt.size() - 1;
  *height = 7;
}

// Where the text goes (p = bottom-left corner of the text, as cv::putText takes it; q = top-left corner of the box
// whose mean decides the colour) and whether it is black: src/VideoWriterManager.cpp:76-114.
// img_bgr: the reference's `img` (h x w x 3, interleaved BGR).
inline void place(const uint8_t *img_bgr, int h, int w, const std::string &txt, Pos pos, int *px, int *py, bool *black)
{
  const int threshold = 105, padding = 5;   // :80, :83 ("bias towards black font")
  int tw, th;
  text_size(txt, &tw, &th);
  int qx, qy;
  if (pos == Pos::TL) { *px = padding; *py = padding + th; qx = padding; qy = padding; }
  else if (pos == Pos::TR) { *px = w - padding - tw; *py = padding + th; qx = w - padding - tw; qy = padding; }
  else if (pos == Pos::BL) { *px = padding; *py = h - padding; qx = padding; qy = h - padding - th; }
  else { *px = w - padding - tw; *py = h - padding; qx = w - padding - tw; qy = h - padding - th; }
  // cv::mean(img(cv::Rect(q, txt_sz))) per channel (OpenCV throws when the box leaves the image; here it is clipped)
  double sum[3] = {0, 0, 0};
  long cnt = 0;
  for (int y = qy; y < qy + th; ++y)
    for (int x = qx; x < qx + tw; ++x) {
      if (y < 0 || y >= h || x < 0 || x >= w) continue;
      for (int c = 0; c < 3; ++c) sum[c] += img_bgr[((size_t)y * w + x) * 3 + c];
      ++cnt;
    }
  const double inv = cnt ? 1.0 / cnt : 0.0;
  const double intensity_avg = 0.114 * sum[0] * inv + 0.587 * sum[1] * inv + 0.299 * sum[2] * inv;   // :110
  *black = 255 - intensity_avg < threshold;                                                          // :111
}

// Draws txt into an interleaved RGB frame (h x w x 3); (px, py) = bottom-left corner of the text.
inline void draw(uint8_t *frame_rgb, int h, int w, const std::string &txt, int px, int py, bool black)
{
  const uint8_t v = black ? 0 : 255;
  for (size_t k = 0; k < txt.size(); ++k) {
    const uint8_t *g = glyph(txt[k]);
    for (int r = 0; r < 7; ++r)
      for (int c = 0; c < 5; ++c) {
        if (!((g[r] >> (4 - c)) & 1)) continue;
        const int x = px + 6 * (int)k + c, y = py - 7 + r;
        if (x < 0 || x >= w || y < 0 || y >= h) continue;
        uint8_t *p = frame_rgb + ((size_t)y * w + x) * 3;
        p[0] = v; p[1] = v; p[2] = v;
      }
  }
}

}  // namespace overlay
