// minicv.hpp — the few OpenCV types the ParallelPixelFunction operator surface names
// (cv::Mat, cv::Range, cv::ParallelLoopBody, cv::parallel_for_), for builds without OpenCV.
// The reference uses OpenCV 2.4.8 (README.md:12); neither this image nor the GPU box has
// it.  Define CHANVESE_WITH_OPENCV to bind the operator to the real types instead.
#ifndef CHANVESE_MINICV_HPP
#define CHANVESE_MINICV_HPP

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>

#define CV_8U 0
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

typedef unsigned char uchar;

namespace cv {

struct Range {
  int start, end;
  Range() : start(0), end(0) {}
  Range(int s, int e) : start(s), end(e) {}
  int size() const { return end - start; }
};

// Row-major, continuous, reference-counted like cv::Mat (copy = shared header, clone = deep).
class Mat {
public:
  int rows, cols;
  uchar *data;
  Mat() : rows(0), cols(0), data(nullptr), type_(0) {}
  Mat(int r, int c, int type) { create(r, c, type); }
  static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data, 0, m.total() * m.elemSize()); return m; }
  void create(int r, int c, int type)
  {
    rows = r; cols = c; type_ = type;
    buf_.reset(new uchar[(size_t)r * c * elemSize()], std::default_delete<uchar[]>());
    data = buf_.get();
  }
  int type() const { return type_; }
  int channels() const { return (type_ >> 3) + 1; }
  size_t elemSize() const { return (size_t)channels() * ((type_ & 7) == CV_64F ? 8 : 1); }
  size_t total() const { return (size_t)rows * cols; }
  bool empty() const { return data == nullptr; }
  bool isContinuous() const { return true; }
  Mat clone() const { Mat m(rows, cols, type_); std::memcpy(m.data, data, total() * elemSize()); return m; }
  template <typename T> T &at(int i, int j) { return reinterpret_cast<T *>(data)[(size_t)i * cols + j]; }
  template <typename T> const T &at(int i, int j) const { return reinterpret_cast<const T *>(data)[(size_t)i * cols + j]; }
  template <typename T> T *ptr(int i = 0) { return reinterpret_cast<T *>(data) + (size_t)i * cols * channels(); }
  template <typename T> const T *ptr(int i = 0) const { return reinterpret_cast<const T *>(data) + (size_t)i * cols * channels(); }

private:
  int type_;
  std::shared_ptr<uchar> buf_;
};

class ParallelLoopBody {
public:
  virtual ~ParallelLoopBody() {}
  virtual void operator()(const Range &range) const = 0;
};

// OpenCV hands the body disjoint sub-ranges from worker threads; one call over the whole
// range is a valid schedule (and the only sensible one when the body launches a GPU kernel).
inline void parallel_for_(const Range &range, const ParallelLoopBody &body, double /*nstripes*/ = -1.) { body(range); }

}  // namespace cv
#endif
