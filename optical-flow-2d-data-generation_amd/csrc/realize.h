// Host "realize" step: blueprints -> device records (fp64 affines in the
// reference's operation order, texture placement, z-order).
//
// Mirrors DataGenerator::RealizeObjectBlueprint and the object set-up half of
// Process_TaskBucket (reference src/caffe/DataGenerator.cpp:1065-1173, 1183-1211)
// and MovingObjectBase::setIntrinsicTransform / setMotion / addBackgroundMotion
// (:302-335).  The 2x3 algebra restates agg::trans_affine (AGG 2.4
// agg_trans_affine.h): multiply = "apply this, then m"; members sx,shy,shx,sy,tx,ty.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "../../include/ofdg.h"
#include "ofdg_device.h"

namespace ofdg {

inline Mat mat_identity() { return Mat{1.0, 0.0, 0.0, 1.0, 0.0, 0.0}; }
inline Mat mat_rotation(double a) { return Mat{std::cos(a), std::sin(a), -std::sin(a), std::cos(a), 0.0, 0.0}; }
inline Mat mat_scaling(double s) { return Mat{s, 0.0, 0.0, s, 0.0, 0.0}; }
inline Mat mat_translation(double x, double y) { return Mat{1.0, 0.0, 0.0, 1.0, x, y}; }
inline Mat mat_mul(const Mat& a, const Mat& m) {  // a *= m
  Mat r;
  r.sx = a.sx * m.sx + a.shy * m.shx;
  r.shx = a.shx * m.sx + a.sy * m.shx;
  r.tx = a.tx * m.sx + a.ty * m.shx + m.tx;
  r.shy = a.sx * m.shy + a.shy * m.sy;
  r.sy = a.shx * m.shy + a.sy * m.sy;
  r.ty = a.tx * m.shy + a.ty * m.sy + m.ty;
  return r;
}
inline Mat mat_invert(const Mat& a) {
  Mat r;
  const double d = 1.0 / (a.sx * a.sy - a.shy * a.shx);
  const double t0 = a.sy * d;
  r.sy = a.sx * d;
  r.shy = -a.shy * d;
  r.shx = -a.shx * d;
  const double t4 = -a.tx * t0 - a.ty * r.shx;
  r.ty = -a.tx * r.shy - a.ty * r.sy;
  r.sx = t0;
  r.tx = t4;
  return r;
}

// A warp-crop use of the batch (mode 9): entry k of the batch's crop table.
struct CropUse {
  int32_t crop;        // index of the served crop
  int32_t background;  // 1: the 2W x 2H upscaled copy is needed (DataGenerator.cpp:1194-1202)
};

struct RealizedBatch {
  std::vector<DevShape> shapes;
  std::vector<DevObject> objects;
  std::vector<DevSample> samples;
  std::vector<CropUse> crops;  // DevObject/DevShape.deform - 1 indexes this table
  std::vector<DevBgPrep> bgprep;  // per sample, if RealizeConfig.background_prep
};

// CropGenerator::get_crop (WarpFields.cpp:516-538): crops are served in order, each
// reuse_same + 1 = 3 times (DataGenerator.cpp:1018); the set is cycled when exhausted.
struct CropServer {
  int n_crops = 0, head = 0, counter = 0, reuse_same = 2;
  int get() {
    const int c = head % n_crops;
    if (++counter > reuse_same) { ++head; counter = 0; }
    return c;
  }
};

struct RealizeConfig {
  int W, H, mode;
  int pool_n, pool_w, pool_h;
  int background_prep = 0;
  // texture sources (TexSource of the ctx); 0 stride = "centre crops of the pool images" computed from pool_w/h
  uint64_t fg_stride = 0, fg_origin = 0, bg_stride = 0, bg_origin = 0;
  // background_prep: where the whole images are - a uniform pool at pool_addr, or one entry per image (mixed pool)
  uint64_t pool_addr = 0;
  const DevTexEntry* tex_table = nullptr;
};

// getRandomizedCrop(2W, 2H, angle, zoom, shift) of a pool image as one coordinate map (DG:87-109).
DevBgPrep make_bg_prep(int pool_w, int pool_h, int W, int H, float angle, float zoom, int shift_x, int shift_y, uint64_t image_addr);
// The texels of the 2W x 2H background texture compose reads: the centre W x H window (frame 0)
// and the window mapped through the texture warp `tex_inv` (frame 1, bilinear), with a margin;
// the whole texture if that leaves it (reflection).
inline void bg_prep_region(const Mat& tex_inv, int W, int H, int32_t* rx0, int32_t* ry0, int32_t* rx1, int32_t* ry1) {
  double lox = W / 2., hix = 3 * W / 2., loy = H / 2., hiy = 3 * H / 2.;
  const double cx[4] = {W / 2., 3 * W / 2., W / 2., 3 * W / 2.}, cy[4] = {H / 2., H / 2., 3 * H / 2., 3 * H / 2.};
  for (int k = 0; k < 4; ++k) {
    const double x = cx[k] * tex_inv.sx + cy[k] * tex_inv.shx + tex_inv.tx, y = cx[k] * tex_inv.shy + cy[k] * tex_inv.sy + tex_inv.ty;
    lox = x < lox ? x : lox; hix = x > hix ? x : hix; loy = y < loy ? y : loy; hiy = y > hiy ? y : hiy;
  }
  const bool finite = lox == lox && hix == hix && loy == loy && hiy == hiy && hix - lox < 1e9 && hiy - loy < 1e9;
  const int m = 3;
  if (!finite || lox - m < 0 || loy - m < 0 || hix + m > 2 * W - 1 || hiy + m > 2 * H - 1) {
    *rx0 = 0; *ry0 = 0; *rx1 = 2 * W - 1; *ry1 = 2 * H - 1;
  } else {
    *rx0 = (int32_t)lox - m; *ry0 = (int32_t)loy - m; *rx1 = (int32_t)hix + m + 1; *ry1 = (int32_t)hiy + m + 1;
    if (*rx1 > 2 * W - 1) *rx1 = 2 * W - 1;
    if (*ry1 > 2 * H - 1) *ry1 = 2 * H - 1;
  }
}

// Returns OFDG_OK or an error code; *msg explains failures.
int realize_batch(const RealizeConfig& cfg, const ofdg_task* tasks, int n_tasks, const ofdg_blueprint* bps,
                  int n_bps, RealizedBatch* out, std::string* msg, CropServer* crops = nullptr);

}  // namespace ofdg
