// ParallelPixelFunction.cpp — see include/ParallelPixelFunction.hpp.
#include "ParallelPixelFunction.hpp"

#include <cmath>
#include <stdexcept>
#include <string>

#include "chanvese_hip.h"

namespace {

const double kPi = 3.14159265358979323846;

double ref_delta(double x, double eps) { return eps / (kPi * (eps * eps + x * x)); }       // src/main.cpp:209
double ref_heaviside(double x, double eps) { return (1 + 2 / kPi * std::atan(x / eps)) / 2; } // src/main.cpp:193

bool close(double a, double b) { return std::fabs(a - b) <= 8e-16 * std::fmax(std::fabs(a), std::fabs(b)) + 1e-300; }

// Probe the callable: which of delta_eps / H_eps / 1 - H_eps is it, and for which eps?
ChanVese::PixelOp identify(const std::function<double(double)> &f, double *eps_out)
{
  if (!f) return ChanVese::PixelOp::Unknown;
  static const double xs[] = {-7.25, -1.0, -0.3, 0.0, 0.5, 2.0, 31.5, 400.0};
  const double f0 = f(0.0), f1 = f(1.0);
  // delta_eps(0) = 1/(pi eps)
  if (f0 > 0 && std::isfinite(f0)) {
    const double e = 1.0 / (kPi * f0);
    double cand[3] = {e, std::nearbyint(e * 1048576.0) / 1048576.0, 0};
    for (int c = 0; c < 2; ++c) {
      bool ok = true;
      for (double x : xs) ok = ok && close(f(x), ref_delta(x, cand[c]));
      if (ok) { *eps_out = cand[c]; return ChanVese::PixelOp::Delta; }
    }
  }
  // H_eps(0) = 1/2 (and 1 - H too); H_eps(1) = 1/2 + atan(1/eps)/pi
  if (close(f0, 0.5) && f1 != 0.5) {
    const bool rising = f1 > 0.5;
    const double t = std::tan(kPi * std::fabs(f1 - 0.5));
    if (t > 0) {
      const double e = 1.0 / t;
      double cand[2] = {e, std::nearbyint(e * 1048576.0) / 1048576.0};
      for (int c = 0; c < 2; ++c) {
        bool ok = true;
        for (double x : xs) {
          const double h = ref_heaviside(x, cand[c]);
          ok = ok && std::fabs(f(x) - (rising ? h : 1 - h)) <= 4e-16;
        }
        if (ok) { *eps_out = cand[c]; return rising ? ChanVese::PixelOp::Heaviside : ChanVese::PixelOp::OneMinusHeaviside; }
      }
    }
  }
  return ChanVese::PixelOp::Unknown;
}

}  // namespace

ParallelPixelFunction::ParallelPixelFunction(cv::Mat &_data, int _w, std::function<double(double)> _func)
  : data(_data), w(_w), func(_func), op_(ChanVese::PixelOp::Unknown), eps_(1.0), device_(0)
{
  op_ = identify(func, &eps_);
}

ParallelPixelFunction::ParallelPixelFunction(cv::Mat &_data, int _w, ChanVese::PixelOp _op, double _eps, int _device)
  : data(_data), w(_w), func(), op_(_op), eps_(_eps), device_(_device)
{}

void ParallelPixelFunction::operator()(const cv::Range &r) const
{
  if (op_ == ChanVese::PixelOp::Unknown) {
    // Any other callable keeps the reference's semantics exactly (src/ParallelPixelFunction.cpp:12-17):
    // the caller's own function, applied on the host over [start, end).  (An opaque std::function
    // cannot run on a GPU; this loop is the caller's code, not a CPU copy of the library's kernels.)
    for (int i = r.start; i < r.end; ++i)
      data.at<double>(i / w, i % w) = func(data.at<double>(i / w, i % w));
    return;
  }
  if (data.type() != CV_64FC1 || !data.isContinuous())
    throw std::invalid_argument("ParallelPixelFunction: data must be a continuous CV_64FC1 matrix");
  // data.at<double>(i / w, i % w) for i in [start, end)  (src/ParallelPixelFunction.cpp:15-16)
  const int rc = cvh_ppf_apply(reinterpret_cast<double *>(data.data), w, r.start, r.end, static_cast<int>(op_), eps_, device_);
  if (rc != CVH_OK) throw std::runtime_error(std::string("ParallelPixelFunction: ") + cvh_last_error(nullptr));
}
