// main.cpp — bin/chan_vese: the reference's command line over the MI355X HIP library.
//
// Keeps the CLI surface of the reference's main() (src/main.cpp:752-874: option names, short
// forms, defaults, multitoken --lambda1/--lambda2, validation order and messages, msg_exit
// behaviour, output naming through add_suffix) and its orchestration (:877-1007), but every
// array operation of the hot path goes through include/chanvese_hip.h into HIP kernels.
// No OpenCV/Boost: images are binary PGM/PPM or PNG (png_io.hpp over zlib), formats cv::imread
// also accepts; outputs keep the input's format like cv::imwrite by extension.
// Reference GUI code is out of scope: -R/--rectangle and -C/--circle (mouse selection) are parsed
// and validated as in the reference, then rejected with a message; their non-interactive forms are
// --rect x,y,w,h (1 inside / 0 outside, src/InteractiveDataRect.cpp:20-27) and --circ cx,cy,r
// (1-pixel outline of ones on zeros, src/InteractiveDataCirc.cpp:18-25).
// -V/--video: the XVID writer (src/VideoWriterManager.cpp) is replaced by an image sequence
// <stem>_frames/frame_NNNNNN<ext> holding the same frames (the input with the contour drawn in
// --line-color, one frame for t = 0 and one after every iteration, :926-931,:997); the overlay
// text of -O is not rendered (no font rasteriser here), --fps has nothing to act on.
// Additions that do not collide with reference options: --dump-u, --dump-mask, --device, --math,
// --state, --rect, --circ, --verbose.
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "chanvese_hip.h"
#include "png_io.hpp"
#include "overlay_text.hpp"

namespace {

// src/main.cpp:173-178
[[noreturn]] void msg_exit(const std::string &msg)
{
  std::cerr << "\n" << msg << "\n\n";
  std::exit(EXIT_FAILURE);
}

// src/main.cpp:158-167 (boost::filesystem parent_path / stem / extension semantics)
std::string add_suffix(const std::string &path, const std::string &suffix, const std::string &delim = "_")
{
  const size_t slash = path.find_last_of('/');
  const std::string parent = slash == std::string::npos ? "" : path.substr(0, slash);
  const std::string file = slash == std::string::npos ? path : path.substr(slash + 1);
  const size_t dot = file.find_last_of('.');
  const bool has_ext = dot != std::string::npos && dot != 0 && file != "..";
  const std::string stem = has_ext ? file.substr(0, dot) : file;
  const std::string ext = has_ext ? file.substr(dot) : "";
  const std::string name = stem + delim + suffix + ext;
  if (slash == std::string::npos) return name;
  return (parent.empty() ? std::string("/") : parent + "/") + name;
}

bool iequals(const std::string &a, const std::string &b)
{
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); ++i)
    if (std::tolower((unsigned char)a[i]) != std::tolower((unsigned char)b[i])) return false;
  return true;
}

// ---- PNM I/O ---------------------------------------------------------------------------
struct Image {
  int h = 0, w = 0, channels = 0;   // channels: 1 (P5) or 3 (P6, stored RGB interleaved)
  std::vector<uint8_t> px;
};

bool read_token(std::istream &in, std::string &tok)
{
  tok.clear();
  int c;
  while ((c = in.get()) != EOF) {
    if (c == '#') { while ((c = in.get()) != EOF && c != '\n') {} continue; }
    if (!std::isspace(c)) { tok.push_back((char)c); break; }
  }
  while ((c = in.peek()) != EOF && !std::isspace(c) && c != '#') tok.push_back((char)in.get());
  return !tok.empty();
}

bool read_pnm(const std::string &path, Image &img)
{
  std::ifstream in(path, std::ios::binary);
  if (!in) return false;
  std::string magic, sw, sh, smax;
  if (!read_token(in, magic) || (magic != "P5" && magic != "P6")) return false;
  if (!read_token(in, sw) || !read_token(in, sh) || !read_token(in, smax)) return false;
  const long w = std::strtol(sw.c_str(), nullptr, 10), h = std::strtol(sh.c_str(), nullptr, 10);
  const long maxv = std::strtol(smax.c_str(), nullptr, 10);
  if (w <= 0 || h <= 0 || w > INT_MAX / 4 || h > INT_MAX / 4 || maxv != 255) return false;
  in.get();  // the single whitespace after maxval
  img.h = (int)h; img.w = (int)w; img.channels = magic == "P5" ? 1 : 3;
  img.px.resize((size_t)h * w * img.channels);
  in.read(reinterpret_cast<char *>(img.px.data()), (std::streamsize)img.px.size());
  return (size_t)in.gcount() == img.px.size();
}

bool write_pnm(const std::string &path, int h, int w, int channels, const uint8_t *px)
{
  std::ofstream out(path, std::ios::binary);
  if (!out) return false;
  out << (channels == 1 ? "P5" : "P6") << "\n" << w << " " << h << "\n255\n";
  out.write(reinterpret_cast<const char *>(px), (std::streamsize)((size_t)h * w * channels));
  return (bool)out;
}

// cv::imread / cv::imwrite stand-ins: format by content on input, by extension on output
bool has_ext(const std::string &path, const char *ext)
{
  const size_t n = std::strlen(ext);
  return path.size() >= n && iequals(path.substr(path.size() - n), ext);
}

bool read_image(const std::string &path, Image &img, bool &is_png)
{
  is_png = pngio::is_png(path);
  if (!is_png) return read_pnm(path, img);
  pngio::Decoded d;
  const std::string err = pngio::read(path, d);
  if (!err.empty()) { std::cerr << "PNG: " << err << "\n"; return false; }
  img.h = d.h; img.w = d.w; img.channels = d.channels; img.px.swap(d.px);
  return true;
}

bool write_image(const std::string &path, int h, int w, int channels, const uint8_t *px)
{
  if (has_ext(path, ".png")) return pngio::write(path, h, w, channels, px);
  return write_pnm(path, h, w, channels, px);
}

// cv::circle(u, centre, radius, 1) with the default thickness 1 and 8-connected line type
// (src/InteractiveDataCirc.cpp:24): OpenCV's integer midpoint circle, restated from memory of
// drawing.cpp (Circle): parity unpinned.
void draw_circle_outline(std::vector<double> &u, int h, int w, int cx, int cy, int radius)
{
  auto put = [&](int x, int y) { if (x >= 0 && x < w && y >= 0 && y < h) u[(size_t)y * w + x] = 1; };
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    const int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
    const int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
    put(x11, y11); put(x12, y11); put(x11, y12); put(x12, y12);
    put(x21, y21); put(x22, y21); put(x21, y22); put(x22, y22);
    dy++;
    err += plus; plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

// ---- option parsing (the subset of boost::program_options behaviour the reference uses) --
struct Spec { const char *lname; char sname; int kind; };  // kind: 0 switch, 1 one value, 2 multitoken
const Spec kSpecs[] = {
    {"help", 'h', 0}, {"input", 'i', 1}, {"mu", 0, 1}, {"nu", 0, 1}, {"dt", 0, 1},
    {"lambda1", 0, 2}, {"lambda2", 0, 2}, {"epsilon", 'e', 1}, {"tolerance", 't', 1},
    {"max-steps", 'N', 1}, {"fps", 'f', 1}, {"overlay-pos", 'P', 1}, {"line-color", 'l', 1},
    {"edge-coef", 'K', 1}, {"laplacian-coef", 'L', 1}, {"segment-time", 'T', 1},
    {"segment", 'S', 0}, {"grayscale", 'g', 0}, {"video", 'V', 0}, {"overlay-text", 'O', 0},
    {"invert-selection", 'I', 0}, {"select", 's', 0}, {"rectangle", 'R', 0}, {"circle", 'C', 0},
    // additions of this build
    {"dump-u", 0, 1}, {"dump-mask", 0, 1}, {"device", 0, 1}, {"math", 0, 1}, {"state", 0, 1}, {"rect", 0, 1}, {"circ", 0, 1}, {"verbose", 0, 0}};

struct Parsed {
  std::vector<std::pair<std::string, std::vector<std::string>>> opts;
  int count(const std::string &n) const { int c = 0; for (auto &o : opts) c += o.first == n; return c; }
  const std::vector<std::string> *get(const std::string &n) const
  {
    const std::vector<std::string> *r = nullptr;
    for (auto &o : opts) if (o.first == n) r = &o.second;
    return r;
  }
};

const Spec *find_long(const std::string &n)
{
  const Spec *hit = nullptr;
  int hits = 0;
  for (const Spec &s : kSpecs) {
    if (n == s.lname) return &s;
    if (std::string(s.lname).compare(0, n.size(), n) == 0) { hit = &s; ++hits; }  // unambiguous prefix
  }
  return hits == 1 ? hit : nullptr;
}
const Spec *find_short(char c)
{
  for (const Spec &s : kSpecs) if (s.sname && s.sname == c) return &s;
  return nullptr;
}
bool looks_like_option(const std::string &t) { return t.size() >= 2 && t[0] == '-' && !std::isdigit((unsigned char)t[1]) && t[1] != '.'; }

Parsed parse_args(int argc, char **argv)
{
  Parsed p;
  for (int i = 1; i < argc; ++i) {
    std::string tok = argv[i];
    const Spec *sp = nullptr;
    std::string attached;
    bool has_attached = false;
    if (tok.rfind("--", 0) == 0) {
      std::string name = tok.substr(2);
      const size_t eq = name.find('=');
      if (eq != std::string::npos) { attached = name.substr(eq + 1); name = name.substr(0, eq); has_attached = true; }
      sp = find_long(name);
      if (!sp) msg_exit("error: unrecognised option '" + tok.substr(0, eq == std::string::npos ? std::string::npos : eq + 2) + "'");
    } else if (tok.size() >= 2 && tok[0] == '-') {
      sp = find_short(tok[1]);
      if (!sp) msg_exit("error: unrecognised option '" + tok.substr(0, 2) + "'");
      if (tok.size() > 2) {
        if (sp->kind == 0) {  // grouped switches: -gs
          for (size_t k = 1; k < tok.size(); ++k) {
            const Spec *s2 = find_short(tok[k]);
            if (!s2 || s2->kind != 0) msg_exit(std::string("error: unrecognised option '-") + tok[k] + "'");
            p.opts.push_back({s2->lname, {}});
          }
          continue;
        }
        attached = tok.substr(2); has_attached = true;
      }
    } else {
      msg_exit("error: too many positional options have been specified on the command line");
    }
    std::vector<std::string> vals;
    if (sp->kind == 0) {
      if (has_attached) msg_exit(std::string("error: option '--") + sp->lname + "' does not take any arguments");
    } else if (sp->kind == 1) {
      if (has_attached) vals.push_back(attached);
      else if (i + 1 < argc && !looks_like_option(argv[i + 1])) vals.push_back(argv[++i]);
      else msg_exit(std::string("error: the required argument for option '--") + sp->lname + "' is missing");
    } else {
      if (has_attached) vals.push_back(attached);
      while (i + 1 < argc && !looks_like_option(argv[i + 1])) vals.push_back(argv[++i]);
      if (vals.empty()) msg_exit(std::string("error: the required argument for option '--") + sp->lname + "' is missing");
    }
    if (sp->kind != 2 && p.count(sp->lname) > 0 && sp->kind == 1)
      msg_exit(std::string("error: option '--") + sp->lname + "' cannot be specified more than once");
    p.opts.push_back({sp->lname, vals});
  }
  return p;
}

double to_double(const std::string &opt, const std::string &v)
{
  char *end = nullptr;
  const double d = std::strtod(v.c_str(), &end);
  if (v.empty() || *end != '\0') msg_exit("error: the argument ('" + v + "') for option '--" + opt + "' is invalid");
  return d;
}
int to_int(const std::string &opt, const std::string &v)
{
  char *end = nullptr;
  const long d = std::strtol(v.c_str(), &end, 10);
  if (v.empty() || *end != '\0' || d > INT_MAX || d < INT_MIN) msg_exit("error: the argument ('" + v + "') for option '--" + opt + "' is invalid");
  return (int)d;
}

void print_help()
{
  std::cout <<
      "Allowed options:\n"
      "  -h [ --help ]                      this message\n"
      "  -i [ --input ] arg                 input image (binary PGM/PPM)\n"
      "  --mu arg (=0.5)                    length penalty parameter (must be positive or zero)\n"
      "  --nu arg (=0)                      area penalty parameter\n"
      "  --dt arg (=1)                      timestep\n"
      "  --lambda1 arg                      penalty of variance inside the contour (default: 1's)\n"
      "  --lambda2 arg                      penalty of variance outside the contour (default: 1's)\n"
      "  -e [ --epsilon ] arg (=1)          smoothing parameter in Heaviside/delta\n"
      "  -t [ --tolerance ] arg (=0.001)    tolerance in stopping condition\n"
      "  -N [ --max-steps ] arg (=-1)       maximum nof iterations (negative means unlimited)\n"
      "  -f [ --fps ] arg (=10)             video fps\n"
      "  -P [ --overlay-pos ] arg (=TL)     overlay tex position; allowed only: TL, BL, TR, BR\n"
      "  -l [ --line-color ] arg (=blue)    contour color (allowed only: black, white, R, G, B, Y, M, C\n"
      "  -K [ --edge-coef ] arg (=10)       coefficient for enhancing edge detection in Perona-Malik\n"
      "  -L [ --laplacian-coef ] arg (=0.25) coefficient in the gradient FD scheme of Perona-Malik (must be [0, 1/4])\n"
      "  -T [ --segment-time ] arg (=20)    number of smoothing steps in Perona-Malik\n"
      "  -S [ --segment ]                   segment the image with Perona-Malik beforehand\n"
      "  -g [ --grayscale ]                 read in as grayscale\n"
      "  -V [ --video ]                     enable video output (here: image sequence <stem>_frames/frame_NNNNNN<ext>)\n"
      "  -O [ --overlay-text ]              add overlay text\n"
      "  -I [ --invert-selection ]          invert selected region (see: select)\n"
      "  -s [ --select ]                    separate the region encolosed by the contour (adds suffix '_selection')\n"
      "  -R [ --rectangle ]                 select rectangular contour interactively (not supported in this build)\n"
      "  -C [ --circle ]                    select circular contour interactively (not supported in this build)\n"
      "MI355X build additions:\n"
      "  --rect x,y,w,h                     rectangular initial contour (1 inside, 0 outside)\n"
      "  --circ cx,cy,r                     circular initial contour (1-pixel outline of ones on zeros)\n"
      "  --dump-u arg                       write the final level set as raw little-endian float64 (h*w)\n"
      "  --dump-mask arg                    write the final mask ((float)u > 0) as PGM or PNG by extension (0/255)\n"
      "  --device arg (=0)                  HIP device\n"
      "  --math arg (=fast)                 strict | fast (see include/chanvese_hip.h)\n"
      "  --state arg (=64)                  64 | 32: level set kept as double (the reference's CV_64FC1) or, DECLARED, as float in GPU memory\n"
      "  --verbose                          print the iteration count and last norm to stderr\n"
      "\n";
}

void cvh_check(cvh_context *ctx, int rc, const char *what)
{
  if (rc != CVH_OK) msg_exit(std::string("Error: ") + what + ": " + cvh_last_error(ctx));
}

}  // namespace

int main(int argc, char **argv)
{
  // defaults: src/main.cpp:759-772
  double mu = 0.5, nu = 0, eps = 1, tol = 0.001, dt = 1, fps = 10, K = 10, L = 0.25, T = 20;
  int max_steps = -1, device = 0;
  std::vector<double> lambda1, lambda2;
  std::string input_filename, text_position = "TL", line_color_str = "blue", dump_u, dump_mask, math = "fast", rect, circ;
  bool grayscale = false, write_video = false, overlay_text = false, object_selection = false, invert = false,
       segment = false, rectangle_contour = false, circle_contour = false;

  const Parsed vm = parse_args(argc, argv);
  auto one = [&](const char *n) -> const std::string * { auto v = vm.get(n); return v && !v->empty() ? &(*v)[0] : nullptr; };
  if (auto v = one("input")) input_filename = *v;
  if (auto v = one("mu")) mu = to_double("mu", *v);
  if (auto v = one("nu")) nu = to_double("nu", *v);
  if (auto v = one("dt")) dt = to_double("dt", *v);
  if (auto v = vm.get("lambda1")) for (auto &s : *v) lambda1.push_back(to_double("lambda1", s));
  if (auto v = vm.get("lambda2")) for (auto &s : *v) lambda2.push_back(to_double("lambda2", s));
  if (auto v = one("epsilon")) eps = to_double("epsilon", *v);
  if (auto v = one("tolerance")) tol = to_double("tolerance", *v);
  if (auto v = one("max-steps")) max_steps = to_int("max-steps", *v);
  if (auto v = one("fps")) fps = to_double("fps", *v);
  if (auto v = one("overlay-pos")) text_position = *v;
  if (auto v = one("line-color")) line_color_str = *v;
  if (auto v = one("edge-coef")) K = to_double("edge-coef", *v);
  if (auto v = one("laplacian-coef")) L = to_double("laplacian-coef", *v);
  if (auto v = one("segment-time")) T = to_double("segment-time", *v);
  if (auto v = one("dump-u")) dump_u = *v;
  if (auto v = one("dump-mask")) dump_mask = *v;
  if (auto v = one("device")) device = to_int("device", *v);
  if (auto v = one("math")) math = *v;
  int state_bits = 64;
  if (auto v = one("state")) state_bits = to_int("state", *v);
  if (auto v = one("rect")) rect = *v;
  if (auto v = one("circ")) circ = *v;
  segment = vm.count("segment"); grayscale = vm.count("grayscale"); write_video = vm.count("video");
  overlay_text = vm.count("overlay-text"); invert = vm.count("invert-selection");
  object_selection = vm.count("select"); rectangle_contour = vm.count("rectangle"); circle_contour = vm.count("circle");
  (void)fps;

  // ---- validation, in the reference's order with its messages: src/main.cpp:786-869
  if (vm.count("help")) { print_help(); return EXIT_SUCCESS; }
  if (!vm.count("input")) msg_exit("Error: you have to specify input file name!");
  else if (!std::ifstream(input_filename).good()) msg_exit("Error: file \"" + input_filename + "\" does not exists!");
  if (dt <= 0) msg_exit("Cannot have negative or zero timestep: " + std::to_string(dt) + ".");
  if (mu < 0) msg_exit("Length penalty parameter cannot be negative: " + std::to_string(mu) + ".");
  if (vm.count("lambda1")) {
    if (grayscale && lambda1.size() != 1) msg_exit("Too many lambda1 values for a grayscale image.");
    else if (!grayscale && lambda1.size() != 3) msg_exit("Number of lambda1 values must be 3 for a colored input image.");
    else if (grayscale && lambda1[0] < 0) msg_exit("The value of lambda1 cannot be negative.");
    else if (!grayscale && (lambda1[0] < 0 || lambda1[1] < 0 || lambda1[2] < 0)) msg_exit("Any value of lambda1 cannot be negative.");
  } else {
    lambda1 = grayscale ? std::vector<double>{1} : std::vector<double>{1, 1, 1};
  }
  if (vm.count("lambda2")) {
    if (grayscale && lambda2.size() != 1) msg_exit("Too many lambda2 values for a grayscale image.");
    else if (!grayscale && lambda2.size() != 3) msg_exit("Number of lambda2 values must be 3 for a colored input image.");
    else if (grayscale && lambda2[0] < 0) msg_exit("The value of lambda2 cannot be negative.");
    else if (!grayscale && (lambda2[0] < 0 || lambda2[1] < 0 || lambda2[2] < 0)) msg_exit("Any value of lambda2 cannot be negative.");
  } else {
    lambda2 = grayscale ? std::vector<double>{1} : std::vector<double>{1, 1, 1};
  }
  // (the reference's eps/tol checks look up non-existent keys and never fire: src/main.cpp:831-834)
  if (!(iequals(text_position, "TL") || iequals(text_position, "BL") || iequals(text_position, "TR") || iequals(text_position, "BR")))
    msg_exit("Invalid text position requested.\nCorrect values are: TL -- top left\n"
             "                    BL -- bottom left\n                    TR -- top right\n"
             "                    BR -- bottom right");
  {
    const char *colors[] = {"red", "green", "blue", "black", "white", "magenta", "yellow", "cyan"};
    bool ok = false;
    for (const char *c : colors) ok = ok || iequals(line_color_str, c);
    if (!ok) msg_exit("Invalid contour color requested.\nCorrect values are: red, green, blue, black, white, magenta, yellow, cyan.");
  }
  if (L > 0.25 || L < 0) msg_exit("The Laplacian coefficient in Perona-Malik segmentation must be between 0 and 0.25.");
  if (vm.count("segment-time") && T < L)
    msg_exit("The segmentation duration must exceed the value of Laplacian coefficient, " + std::to_string(L) + ".");
  if (rectangle_contour && circle_contour) msg_exit("Cannot initialize with both rectangular and circular contour");
  if (rectangle_contour || circle_contour)
    msg_exit("Interactive contour selection (-R/-C) needs a display and is not supported in this build; use --rect x,y,w,h or --circ cx,cy,r.");
  if (!rect.empty() && !circ.empty()) msg_exit("Cannot initialize with both rectangular and circular contour");
  if (math != "strict" && math != "fast") msg_exit("error: the argument ('" + math + "') for option '--math' is invalid");
  if (state_bits != 64 && state_bits != 32) msg_exit("error: the argument ('" + std::to_string(state_bits) + "') for option '--state' is invalid");

  // ---- read the image: src/main.cpp:877-887 (8-bit gray or BGR)
  Image file;
  bool input_is_png = false;
  if (!read_image(input_filename, file, input_is_png)) msg_exit("Error on opening \"" + input_filename + "\" (probably not an image)!");
  if (!input_is_png && has_ext(input_filename, ".png")) msg_exit("Error on opening \"" + input_filename + "\" (probably not an image)!");
  const int h = file.h, w = file.w;
  const size_t n = (size_t)h * w;
  const int nof_channels = grayscale ? 1 : 3;
  // img: what the reference calls `img` (3 channels, BGR); planes: what cv::split leaves (:934-937)
  std::vector<uint8_t> img_bgr(n * 3);
  std::vector<std::vector<uint8_t>> planes(nof_channels, std::vector<uint8_t>(n));
  for (size_t q = 0; q < n; ++q) {
    uint8_t b, g, r;
    if (file.channels == 1) { b = g = r = file.px[q]; }
    else { r = file.px[3 * q]; g = file.px[3 * q + 1]; b = file.px[3 * q + 2]; }
    if (grayscale) {
      // cv::imread(..., GRAYSCALE) of a colour file.  PxM decoder: icvCvt_BGR2Gray_8u_C3C1R, 14-bit fixed
      // point BT.601 (R 4899, G 9617, B 1868).  PNG decoder: libpng's png_set_rgb_to_gray(0.299, 0.587),
      // 15-bit fixed point (R 9798, G 19235, B 3735, rounded; equal channels pass through).
      uint8_t y;
      if (file.channels == 1) y = b;
      else if (!input_is_png) y = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
      else y = (r == g && g == b) ? r : (uint8_t)((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15);
      img_bgr[3 * q] = img_bgr[3 * q + 1] = img_bgr[3 * q + 2] = y;  // cvtColor GRAY2RGB (:885)
      planes[0][q] = y;
    } else {
      img_bgr[3 * q] = b; img_bgr[3 * q + 1] = g; img_bgr[3 * q + 2] = r;
      planes[0][q] = b; planes[1][q] = g; planes[2][q] = r;
    }
  }

  // ---- constants: src/main.cpp:890-895
  max_steps = max_steps < 0 ? std::numeric_limits<int>::max() : max_steps;
  cvh_params prm;
  cvh_default_params(&prm);
  prm.mu = mu; prm.nu = nu; prm.dt = dt; prm.eps = eps; prm.tol = tol;
  for (int k = 0; k < nof_channels; ++k) { prm.lambda1[k] = lambda1[k]; prm.lambda2[k] = lambda2[k]; }

  cvh_context *ctx = nullptr;
  if (cvh_create(&ctx, h, w, nof_channels, &prm, device) != CVH_OK)
    msg_exit(std::string("Error: cannot initialise the HIP backend: ") + cvh_last_error(nullptr));
  cvh_check(ctx, cvh_set_option(ctx, "math_mode", math == "strict" ? CVH_MATH_STRICT : CVH_MATH_FAST), "math_mode");
  if (state_bits == 32) cvh_check(ctx, cvh_set_option(ctx, "state", 32), "state");   // declared FP32-state mode: never the default
  {
    std::vector<const uint8_t *> pp;
    for (auto &p : planes) pp.push_back(p.data());
    cvh_check(ctx, cvh_set_image(ctx, pp.data()), "cvh_set_image");
  }

  // ---- level set: src/main.cpp:898-923
  if (!rect.empty()) {
    int rx, ry, rw, rh;
    if (std::sscanf(rect.c_str(), "%d,%d,%d,%d", &rx, &ry, &rw, &rh) != 4 || rw <= 0 || rh <= 0)
      msg_exit("You must specify the contour with non-zero dimensions");
    std::vector<double> u(n, 0.0);  // src/InteractiveDataRect.cpp:24-25
    for (int i = std::max(ry, 0); i < std::min(ry + rh, h); ++i)
      for (int j = std::max(rx, 0); j < std::min(rx + rw, w); ++j) u[(size_t)i * w + j] = 1;
    cvh_check(ctx, cvh_set_levelset(ctx, u.data()), "cvh_set_levelset");
  } else if (!circ.empty()) {
    int cx, cy, cr;
    if (std::sscanf(circ.c_str(), "%d,%d,%d", &cx, &cy, &cr) != 3 || cr <= 0)  // is_ok(): radius > 0
      msg_exit("You must specify the contour with non-zero dimensions");
    std::vector<double> u(n, 0.0);  // src/InteractiveDataCirc.cpp:23-24
    draw_circle_outline(u, h, w, cx, cy, cr);
    cvh_check(ctx, cvh_set_levelset(ctx, u.data()), "cvh_set_levelset");
  } else {
    cvh_check(ctx, cvh_init_checkerboard(ctx), "cvh_init_checkerboard");
  }

  // ---- video frames: src/main.cpp:926-931 (first frame before the channels are split / smoothed)
  const size_t dot = input_filename.find_last_of('.');
  const size_t slash = input_filename.find_last_of('/');
  const bool with_ext = dot != std::string::npos && (slash == std::string::npos || dot > slash + 1);
  const std::string frames_dir = (with_ext ? input_filename.substr(0, dot) : input_filename) + "_frames";
  const std::string frame_ext = with_ext ? input_filename.substr(dot) : std::string(".ppm");
  uint8_t contour_bgr[3] = {255, 0, 0};  // ChanVese::Colors::blue = CV_RGB(0,0,255), src/main.cpp:111-118,747
  {
    struct { const char *name; uint8_t r, g, b; } table[] = {{"red", 255, 0, 0}, {"green", 0, 255, 0}, {"blue", 0, 0, 255},
      {"black", 0, 0, 0}, {"white", 255, 255, 255}, {"magenta", 255, 0, 255}, {"yellow", 255, 255, 0}, {"cyan", 0, 255, 255}};
    for (auto &e : table) if (iequals(line_color_str, e.name)) { contour_bgr[0] = e.b; contour_bgr[1] = e.g; contour_bgr[2] = e.r; }
  }
  int frame_no = 0;
  std::vector<uint8_t> contour, frame;
  overlay::Pos opos = overlay::Pos::TL;   // src/main.cpp:837-849
  if (iequals(text_position, "BL")) opos = overlay::Pos::BL;
  else if (iequals(text_position, "TR")) opos = overlay::Pos::TR;
  else if (iequals(text_position, "BR")) opos = overlay::Pos::BR;
  auto write_frame = [&](const std::string &txt) {  // VideoWriterManager::write_frame, src/VideoWriterManager.cpp:40-57
    contour.resize(n); frame.resize(n * 3);
    cvh_check(ctx, cvh_get_contour(ctx, contour.data()), "cvh_get_contour");
    for (size_t q = 0; q < n; ++q) {   // frames are RGB files; img_bgr is the reference's `img`
      const bool c = contour[q] != 0;
      frame[3 * q] = c ? contour_bgr[2] : img_bgr[3 * q + 2];
      frame[3 * q + 1] = c ? contour_bgr[1] : img_bgr[3 * q + 1];
      frame[3 * q + 2] = c ? contour_bgr[0] : img_bgr[3 * q];
    }
    if (overlay_text && !txt.empty()) {   // :47-53: colour and corner from overlay_color (:76-114), own 5x7 glyphs
      int px, py; bool black;
      overlay::place(img_bgr.data(), h, w, txt, opos, &px, &py, &black);
      overlay::draw(frame.data(), h, w, txt, px, py, black);
    }
    char name[64];
    std::snprintf(name, sizeof(name), "/frame_%06d", frame_no++);
    const std::string path = frames_dir + name + (has_ext(frame_ext, ".png") ? ".png" : ".ppm");
    if (!write_image(path, h, w, 3, frame.data())) msg_exit("Error: cannot write \"" + path + "\"");
  };
  if (write_video) {
    if (mkdir(frames_dir.c_str(), 0777) != 0 && errno != EEXIST) msg_exit("Error: cannot create \"" + frames_dir + "\"");
    write_frame("t = 0");   // :930
  }

  // ---- Perona-Malik: src/main.cpp:940-947
  if (segment) {
    cvh_check(ctx, cvh_perona_malik(ctx, K, L, T), "cvh_perona_malik");
    std::vector<uint8_t *> pp;
    for (auto &p : planes) pp.push_back(p.data());
    cvh_check(ctx, cvh_get_image(ctx, pp.data()), "cvh_get_image");
    std::vector<uint8_t> out(n * nof_channels);
    for (size_t q = 0; q < n; ++q) {
      if (nof_channels == 1) out[q] = planes[0][q];
      else { out[3 * q] = planes[2][q]; out[3 * q + 1] = planes[1][q]; out[3 * q + 2] = planes[0][q]; }  // BGR -> RGB file order
    }
    if (!write_image(add_suffix(input_filename, "pm"), h, w, nof_channels, out.data()))
      msg_exit("Error: cannot write \"" + add_suffix(input_filename, "pm") + "\"");
  }

  // ---- timestep loop: src/main.cpp:950-1001 (stop condition and every iteration on the GPU)
  int steps_done = 0;
  double last_norm = 0;
  if (!write_video) {
    cvh_check(ctx, cvh_run(ctx, max_steps, &steps_done, &last_norm), "cvh_run");
  } else {
    // one iteration per frame; the frame is saved before the stop test (:997-1000)
    for (int t = 1; t <= max_steps; ++t) {
      int stopped = 0;
      cvh_check(ctx, cvh_enqueue_steps(ctx, 1), "cvh_enqueue_steps");
      cvh_check(ctx, cvh_sync(ctx, &steps_done, &last_norm, &stopped), "cvh_sync");
      write_frame("t = " + std::to_string(t));   // :997
      if (stopped) break;
    }
  }

  if (!dump_u.empty()) {
    std::vector<double> u(n);
    cvh_check(ctx, cvh_get_levelset(ctx, u.data()), "cvh_get_levelset");
    std::ofstream out(dump_u, std::ios::binary);
    out.write(reinterpret_cast<const char *>(u.data()), (std::streamsize)(n * sizeof(double)));
    if (!out) msg_exit("Error: cannot write \"" + dump_u + "\"");
  }
  if (!dump_mask.empty()) {
    std::vector<uint8_t> m(n);
    cvh_check(ctx, cvh_get_mask(ctx, m.data(), invert ? 1 : 0), "cvh_get_mask");
    for (auto &v : m) v = v ? 255 : 0;
    if (!write_image(dump_mask, h, w, 1, m.data())) msg_exit("Error: cannot write \"" + dump_mask + "\"");
  }

  // ---- selection: src/main.cpp:1004-1005
  if (object_selection) {
    std::vector<uint8_t> sel(n * 3), rgb(n * 3);
    cvh_check(ctx, cvh_separate(ctx, img_bgr.data(), invert ? 1 : 0, sel.data()), "cvh_separate");
    for (size_t q = 0; q < n; ++q) { rgb[3 * q] = sel[3 * q + 2]; rgb[3 * q + 1] = sel[3 * q + 1]; rgb[3 * q + 2] = sel[3 * q]; }
    if (!write_image(add_suffix(input_filename, "selection"), h, w, 3, rgb.data()))
      msg_exit("Error: cannot write \"" + add_suffix(input_filename, "selection") + "\"");
  }
  if (vm.count("verbose"))  // the reference prints nothing
    std::fprintf(stderr, "chan_vese: %d iterations, last ||u_diff|| = %.17g\n", steps_done, last_norm);
  cvh_destroy(ctx);
  return EXIT_SUCCESS;
}
