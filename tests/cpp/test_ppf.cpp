// test_ppf.cpp — the reference's call site for the operator (src/main.cpp:894-895,988-989),
// compiled against include/ParallelPixelFunction.hpp.  Reads doubles from stdin
// (h w eps op then h*w values), prints the mapped values with 17 digits.
#include <cmath>
#include <cstdio>
#include <functional>
#include <iostream>

#include "ParallelPixelFunction.hpp"

static double regularized_heaviside(double x, double eps = 1)
{
  const double pi = 3.14159265358979323846;
  return (1 + 2 / pi * std::atan(x / eps)) / 2;
}
static double regularized_delta(double x, double eps = 1)
{
  const double pi = 3.14159265358979323846;
  return eps / (pi * (std::pow(eps, 2) + std::pow(x, 2)));
}

int main()
{
  int h, w, op;
  double eps;
  if (!(std::cin >> h >> w >> eps >> op)) return 2;
  cv::Mat u(h, w, CV_64FC1);
  for (int i = 0; i < h * w; ++i) std::cin >> u.ptr<double>()[i];
  const auto heaviside = std::bind(regularized_heaviside, std::placeholders::_1, eps);
  const auto delta = std::bind(regularized_delta, std::placeholders::_1, eps);
  cv::Mat u_cp = u.clone();  // src/main.cpp:988
  try {
    if (op == 0) cv::parallel_for_(cv::Range(0, h * w), ParallelPixelFunction(u_cp, w, delta));  // :989
    else if (op == 1) cv::parallel_for_(cv::Range(0, h * w), ParallelPixelFunction(u_cp, w, heaviside));
    else if (op == 2) cv::parallel_for_(cv::Range(0, h * w), ParallelPixelFunction(u_cp, w, [&heaviside](double x) -> double { return 1 - heaviside(x); }));
    else if (op == 3) cv::parallel_for_(cv::Range(3, h * w - 2), ParallelPixelFunction(u_cp, w, ChanVese::PixelOp::Delta, eps));
    else if (op == 9) cv::parallel_for_(cv::Range(0, h * w), ParallelPixelFunction(u_cp, w, [](double x) { return std::sin(x); }));
    else cv::parallel_for_(cv::Range(2, h * w - 5), ParallelPixelFunction(u_cp, w, [](double x) { return x * x; }));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  for (int i = 0; i < h * w; ++i) std::printf("%.17g\n", u_cp.ptr<double>()[i]);
  return 0;
}
