// ChanVeseCommon.hpp — hot-path constants of the reference's include/ChanVeseCommon.hpp
// (Region :19, finite-difference stencils :46-54 / src/main.cpp:120-125) plus the tag that
// names a pixel function for the device.  Colours, text positions and the window title of
// the reference header belong to its GUI/video code, which is out of scope here.
#ifndef CHANVESECOMMON_HPP
#define CHANVESECOMMON_HPP

typedef unsigned char uchar;
typedef unsigned long ulong;

namespace ChanVese
{
  /// include/ChanVeseCommon.hpp:19 — which side of the contour region_variance averages
  enum Region { Inside, Outside };

  /// Functions the ParallelPixelFunction operator can run on the GPU (values = cvh_pixel_op)
  enum class PixelOp { Delta = 0, Heaviside = 1, OneMinusHeaviside = 2, Unknown = -1 };

  /// Finite-difference stencils (src/main.cpp:120-125); correlation, anchor at the centre
  struct Kernel
  {
    static constexpr double fwd[3] = {0, -1, 1};      ///< fwd_x / fwd_y
    static constexpr double bwd[3] = {-1, 1, 0};      ///< bwd_x / bwd_y
    static constexpr double ctr[3] = {-0.5, 0, 0.5};  ///< ctr_x / ctr_y
  };

  /// eta of curvature(), src/main.cpp:347
  constexpr double eta = 1E-8;
}

#endif  // CHANVESECOMMON_HPP
