// ParallelPixelFunction.hpp — the reference's operator surface for the per-pixel map,
// backed by the MI355X HIP library (include/chanvese_hip.h).
//
// Mirrors include/ParallelPixelFunction.hpp:16-39 of the reference: same class name, same
// constructor (cv::Mat& data, int w, std::function<double(double)> func), same
// `virtual void operator()(const cv::Range&) const`, usable with cv::parallel_for_ exactly
// as at src/main.cpp:988-989.  An opaque std::function cannot execute on a GPU, so:
//   * the tagged constructor (data, w, PixelOp, eps) names the function directly;
//   * the std::function constructor IDENTIFIES the callable by probing it (about 20 calls at fixed
//     arguments -- observable for a stateful functor) against the three functions the reference ever
//     maps over pixels -- regularized_delta (src/main.cpp:204-210), regularized_heaviside (:188-194)
//     and the Outside lambda 1 - heaviside (:267) -- and recovers their epsilon; a positive match runs
//     on the GPU.  Any OTHER callable keeps the reference's behaviour (src/ParallelPixelFunction.cpp:12-17):
//     operator() applies the caller's function on the host, data(i/w, i%w) = func(data(i/w, i%w)).
// operator()(Range(a,b)) replaces elements [a,b) of the flat index range of the w-wide
// CV_64FC1 matrix by f(x), in place (src/ParallelPixelFunction.cpp:15-16), on the GPU.
#ifndef PARALLELPIXELFUNCTION_HPP
#define PARALLELPIXELFUNCTION_HPP

#include <functional>

#ifdef CHANVESE_WITH_OPENCV
#include <opencv2/imgproc/imgproc.hpp>
#else
#include "minicv.hpp"
#endif

#include "ChanVeseCommon.hpp"

class ParallelPixelFunction : public cv::ParallelLoopBody
{
public:
  /// Reference constructor (include/ParallelPixelFunction.hpp:26-28).
  ParallelPixelFunction(cv::Mat &_data, int _w, std::function<double(double)> _func);
  /// Tagged constructor: the function is named, nothing is probed.
  ParallelPixelFunction(cv::Mat &_data, int _w, ChanVese::PixelOp _op, double _eps, int _device = 0);
  /// Needed by cv::parallel_for_ (include/ParallelPixelFunction.hpp:33).
  virtual void operator()(const cv::Range &r) const;

  /// What the std::function was recognised as (PixelOp::Unknown if nothing matched).
  ChanVese::PixelOp op() const { return op_; }
  double eps() const { return eps_; }

private:
  cv::Mat &data;
  const int w;
  const std::function<double(double)> func;
  ChanVese::PixelOp op_;
  double eps_;
  int device_;
};

#endif  // PARALLELPIXELFUNCTION_HPP
