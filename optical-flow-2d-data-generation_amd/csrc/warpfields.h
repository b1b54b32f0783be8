// Mode-9 warp fields: host-side displacer placement and the device records the field
// kernels consume.  Mirrors the reference's src/caffe/WarpFields.cpp (WF):
// Supports::Gaussian2D (WF:88-112), Displacers::{Translation, Rotation, Zoom} (WF:191-260),
// CropGenerator::worker_thread_loop's placement on a triangular grid (WF:556-610), with
// the std::random_device seed replaced by an explicit seed (the reference's fields are
// not reproducible, SURVEY F-8).
#pragma once
#include <stdint.h>

#include <vector>

namespace ofdg {

// One displacer with its Gaussian2D support; every constant is the float the
// reference's constructors compute (host libm), so the device only does fp32 algebra.
struct DevDisplacer {
  int32_t type;  // 0 Translation, 1 Rotation, 2 Zoom
  float cx, cy;
  float dx, dy;                                          // Translation
  float sin_omega, cos_omega, sin_nomega, cos_nomega;    // Rotation
  float factor, ifactor;                                 // Zoom
  // Gaussian2D support
  float scx, scy, a, b, c, d, ratio_x_y, two_sigma_sq, gauss_prefactor, normalizer;
  float pad;
};

struct DisplacerParams {  // the 9 doubles per displacer the placement loop draws
  double type, p0, p1, p2, sup_cx, sup_cy, sup_sx, sup_sy, sup_angle;
};

// Placement + parameter draws for one big field of side 3*max(W,H).
std::vector<DisplacerParams> make_displacer_params(int W, int H, uint32_t seed);
std::vector<DevDisplacer> make_device_displacers(const std::vector<DisplacerParams>& p);

// One served crop: (W+1) x (H+1) (background: 2W x 2H) planes flow x, flow y, iflow x, iflow y.
struct DevCrop {
  const float* data;  // 4 planes of w*h floats
  int32_t w, h;
  float max_disp;     // max |iflow| over the crop (NaNs ignored), for box dilation
  int32_t pad;
};

}  // namespace ofdg
