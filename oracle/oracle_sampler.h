// TEST INFRASTRUCTURE -- parity oracle, not product code.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// CPU restatement of the reference's blueprint sampler.  It calls libstdc++'s
// <random> exactly the way the reference does (include/caffe/data_generation/
// SimpleRandom.h:15-145), so the draw sequences are those of the reference
// built with the same libstdc++.  Citations: DG = src/caffe/DataGenerator.cpp.
#pragma once
#include <cmath>
#include <limits>
#include <random>
#include <stdexcept>
#include <vector>

#include "../include/ofdg.h"

namespace oracle {

static const double kAggPi = 3.14159265358979323846;  // agg::pi

// ---- SimpleRandom.h:75-142 -------------------------------------------------
struct FixedRangeUniformInt {   // SimpleRandom.h:75-89
  std::mt19937 eng;
  std::uniform_int_distribution<> dist;
  FixedRangeUniformInt(int a, int b, int seed) : eng(seed), dist(a, b) {}
  int operator()() { return dist(eng); }
};
struct FixedRangeUniformFloat { // SimpleRandom.h:95-109 (double dist, float result)
  std::mt19937 eng;
  std::uniform_real_distribution<> dist;
  FixedRangeUniformFloat(float a, float b, int seed) : eng(seed), dist(a, b) {}
  float operator()() { return dist(eng); }
};
struct FixedMeanStddevNormalFloat { // SimpleRandom.h:130-142
  std::mt19937 eng;
  std::normal_distribution<float> dist;
  FixedMeanStddevNormalFloat(float mean, float stddev, int seed)
      : eng(seed), dist(mean, stddev) {}
  float operator()() { return dist(eng); }
};

// ---- FlyingChairsRandom, DG:826-922 ----------------------------------------
inline float baseGauss(float a, float b, float input, float normalize) {  // DG:828-831
  float sample{input * ((b + a) / 2.f - a) / normalize + (b + a) / 2.f};
  return ((a <= sample and sample <= b) ? sample : (b + a) / 2.);
}
struct Uniform {  // DG:864-870
  FixedRangeUniformFloat r;
  Uniform(float a, float b, int seed) : r(a, b, seed) {}
  float operator()() { return r(); }
};
struct Trigger {  // DG:840-849 (Trigger<Uniform>)
  float threshold;
  Uniform r;
  Trigger(float a, float b, float threshold, int seed) : threshold(threshold), r(a, b, seed) {}
  bool operator()() { return (r() < threshold); }
};
template <typename T>
struct Choice {  // DG:852-861
  std::vector<T> options;
  FixedRangeUniformInt r;
  Choice(std::vector<T> o, int seed) : options(o), r(0, (int)options.size() - 1, seed) {}
  T operator()() { return options[r()]; }
};
struct GaussianSq {  // DG:882-890
  float a, b;
  FixedMeanStddevNormalFloat r;
  GaussianSq(float a, float b, int seed) : a(a), b(b), r(0, 1, seed) {}
  float operator()() {
    float tmp = r();
    tmp = ((tmp > 0) ? std::pow(tmp, 2) : -std::pow(tmp, 2));
    return baseGauss(a, b, tmp, 6);
  }
};
struct Gaussian3 {  // DG:893-900
  float a, b;
  FixedMeanStddevNormalFloat r;
  Gaussian3(float a, float b, int seed) : a(a), b(b), r(0, 1, seed) {}
  float operator()() {
    float tmp = std::pow(r(), 3);
    return baseGauss(a, b, tmp, 10);
  }
};
struct Gaussian4 {  // DG:903-911
  float a, b;
  FixedMeanStddevNormalFloat r;
  Gaussian4(float a, float b, int seed) : a(a), b(b), r(0, 1, seed) {}
  float operator()() {
    float tmp = r();
    tmp = ((tmp > 0) ? std::pow(tmp, 4) : -std::pow(tmp, 4));
    return baseGauss(a, b, tmp, 15);
  }
};
struct GaussianMeanSigmaRange {  // DG:914-921
  float a, b, mean, sigma;
  FixedMeanStddevNormalFloat r;
  GaussianMeanSigmaRange(float a, float b, float mean, float sigma, int seed)
      : a(a), b(b), mean(mean), sigma(sigma), r(0, 1, seed) {}
  float operator()() {
    float tmp = r() * sigma + mean;
    return (((a <= tmp) and (tmp <= b)) ? tmp : mean);
  }
};

// ---- ObjectParametersGenerator, DG:1358-2835 --------------------------------
// The 13 per-mode constructor tables (DG:1363-2001) differ from mode 7's only in
// the entries collected in ModeTable below.
struct ModeTable {
  float bg_rot_trig_a, bg_rot_trig_b, bg_rot_trig_thr;
  double bg_rot_deg;                 // GaussianSq(-d*pi/180, d*pi/180)
  float bg_trans;                    // Gaussian4(-t, t)
  float bg_scale_trig_a, bg_scale_trig_b, bg_scale_trig_thr;
  float bg_scale_a, bg_scale_b;
  int obj_types;                     // bitmask: 1 ellipse, 2 polygon, 4 composite
  float obj_trans;                   // Gaussian3(-t, t)
  bool obj_init_rot;                 // false: Uniform(0,0) (mode 1)
  float obj_rot_trig_a, obj_rot_trig_b, obj_rot_trig_thr;
  double obj_rot_deg;
  float obj_scale_trig_a, obj_scale_trig_b, obj_scale_trig_thr;
  float obj_scale_a, obj_scale_b;
  float deform_thr;
};

inline ModeTable mode_table(int mode) {
  // mode 7 (DG:1654-1703)
  ModeTable t{0, 1, 0.3f, 10., 40, 0, 1, 0.6f, 0.93f, 1.07f, 7, 120, true,
              0, 1, 0.7f, 30., 0, 1, 0.7f, 0.8f, 1.2f, 0.f};
  auto no_bg_rot = [&] { t.bg_rot_trig_a = 0; t.bg_rot_trig_b = 0; t.bg_rot_trig_thr = 1; t.bg_rot_deg = 0; };
  auto no_bg_scale = [&] { t.bg_scale_trig_a = 0; t.bg_scale_trig_b = 0; t.bg_scale_trig_thr = 1; t.bg_scale_a = 1; t.bg_scale_b = 1; };
  auto no_obj_rot = [&] { t.obj_rot_trig_a = 0; t.obj_rot_trig_b = 0; t.obj_rot_trig_thr = 1; t.obj_rot_deg = 0; };
  auto no_obj_scale = [&] { t.obj_scale_trig_a = 0; t.obj_scale_trig_b = 0; t.obj_scale_trig_thr = 1; t.obj_scale_a = 1; t.obj_scale_b = 1; };
  switch (mode) {
    case 1: no_bg_rot(); no_bg_scale(); no_obj_rot(); no_obj_scale(); t.obj_types = 2; t.obj_init_rot = false; break;  // DG:1364-1411
    case 2: no_bg_rot(); no_bg_scale(); no_obj_rot(); no_obj_scale(); t.obj_types = 2; break;  // DG:1412-1459
    case 3: no_bg_rot(); no_bg_scale(); no_obj_rot(); no_obj_scale(); t.obj_types = 1; break;  // DG:1460-1507
    case 4: no_bg_scale(); no_obj_scale(); t.obj_types = 3; break;                             // DG:1508-1555
    case 5: t.obj_types = 3; break;                                                           // DG:1556-1603
    case 6: break;                                                                            // DG:1604-1653
    case 7: break;
    case 8: no_bg_rot(); no_bg_scale(); no_obj_rot(); no_obj_scale(); t.obj_types = 3; break;  // DG:1704-1751
    case 9: t.deform_thr = 0.2f; break;                                                       // DG:1752-1801
    case 10:  // DG:1802-1851
      t.bg_rot_trig_thr = 0.176f; t.bg_rot_deg = 5; t.bg_trans = 20; t.bg_scale_trig_thr = 0.429f;
      t.bg_scale_a = 0.965f; t.bg_scale_b = 1.035f; t.obj_trans = 60; t.obj_rot_trig_thr = 0.539f;
      t.obj_rot_deg = 15; t.obj_scale_trig_thr = 0.539f; t.obj_scale_a = 0.9f; t.obj_scale_b = 1.1f; break;
    case 11:  // DG:1852-1901
      t.bg_rot_trig_thr = 0.462f; t.bg_rot_deg = 20; t.bg_trans = 80; t.bg_scale_trig_thr = 0.75f;
      t.bg_scale_a = 0.86f; t.bg_scale_b = 1.14f; t.obj_trans = 240; t.obj_rot_trig_thr = 0.824f;
      t.obj_rot_deg = 60; t.obj_scale_trig_thr = 0.824f; t.obj_scale_a = 0.6f; t.obj_scale_b = 1.4f; break;
    case 12:  // DG:1902-1951
      t.bg_rot_trig_thr = 0.125f; t.bg_rot_deg = 3.3; t.bg_trans = 13.3f; t.bg_scale_trig_thr = 0.333f;
      t.bg_scale_a = 0.976f; t.bg_scale_b = 1.023f; t.obj_trans = 40; t.obj_rot_trig_thr = 0.437f;
      t.obj_rot_deg = 10; t.obj_scale_trig_thr = 0.437f; t.obj_scale_a = 0.933f; t.obj_scale_b = 1.066f; break;
    case 13:  // DG:1952-2001
      t.bg_rot_trig_thr = 0.563f; t.bg_rot_deg = 30; t.bg_trans = 120; t.bg_scale_trig_thr = 0.818f;
      t.bg_scale_a = 0.79f; t.bg_scale_b = 1.21f; t.obj_trans = 360; t.obj_rot_trig_thr = 0.875f;
      t.obj_rot_deg = 90; t.obj_scale_trig_thr = 0.875f; t.obj_scale_a = 0.4f; t.obj_scale_b = 1.6f; break;
    default: throw std::runtime_error("BAD MODE");  // DG:2004
  }
  return t;
}

inline std::vector<int> type_options(int mask) {
  std::vector<int> v;
  if (mask & 1) v.push_back(OFDG_OBJ_ELLIPSE);
  if (mask & 2) v.push_back(OFDG_OBJ_POLYGON);
  if (mask & 4) v.push_back(OFDG_OBJ_COMPOSITE);
  return v;
}

class Sampler {
 public:
  int MODE, W, H, num_objects_override;
  ModeTable t;
  int seed = 0;
  // declaration order == seed order (DG:1365-1409)
  FixedRangeUniformInt RNG_BgTexID;
  Uniform RNG_BgInitRot;
  Choice<int> RNG_BgInitTransX, RNG_BgInitTransY;
  Trigger RNG_BgRotTrigger;
  GaussianSq RNG_BgRot;
  Gaussian4 RNG_BgTransX, RNG_BgTransY;
  Trigger RNG_BgScaleTrigger;
  Uniform RNG_BgInitScale;
  GaussianSq RNG_BgScale;
  Uniform RNG_NumberOfFgObjects;
  Choice<int> RNG_ObjType;
  FixedRangeUniformInt RNG_ObjTexID;
  Uniform RNG_ObjInitTransX, RNG_ObjInitTransY;
  Gaussian3 RNG_ObjTransX, RNG_ObjTransY;
  Uniform RNG_ObjInitRot;
  Trigger RNG_ObjRotTrigger;
  GaussianSq RNG_ObjRot;
  GaussianMeanSigmaRange RNG_ObjInitScale;
  Trigger RNG_ObjScaleTrigger;
  GaussianSq RNG_ObjScale;
  FixedRangeUniformInt RNG_ObjTexShiftX, RNG_ObjTexShiftY;
  FixedRangeUniformFloat RNG_ObjTexRot, RNG_ObjTexZoom;
  Uniform RNG_ElliObj_ScaleX, RNG_ElliObj_ScaleY;
  FixedRangeUniformInt RNG_PolyObj_spokes;
  Uniform RNG_PolyObj_dphi, RNG_PolyObj_r, RNG_PolyObj_ScaleX, RNG_PolyObj_ScaleY;
  Trigger RNG_PolyObj_CurveTrigger;
  Uniform RNG_CompObjInitTransX, RNG_CompObjInitTransY;
  FixedRangeUniformInt RNG_CompObiNumberOfComponents;
  Trigger RNG_ComponentIsAdditive;
  Uniform RNG_ComponentOffset;
  Trigger RNG_ObjIsExtraThin;
  Trigger RNG_ObjDeformsNonrigidly;
  Uniform RNG_GenericUniform;
  Trigger RNG_GenericTrigger;

  Sampler(int mode, int W, int H, int num_objects)
      : MODE(mode), W(W), H(H), num_objects_override(num_objects), t(mode_table(mode)),
        RNG_BgTexID(0, std::numeric_limits<int>::max(), seed++),
        RNG_BgInitRot(-kAggPi, kAggPi, seed++),
        RNG_BgInitTransX({0, W}, seed++),
        RNG_BgInitTransY({0, H}, seed++),
        RNG_BgRotTrigger(t.bg_rot_trig_a, t.bg_rot_trig_b, t.bg_rot_trig_thr, seed++),
        RNG_BgRot(t.bg_rot_deg ? -t.bg_rot_deg * kAggPi / 180. : 0., t.bg_rot_deg * kAggPi / 180., seed++),
        RNG_BgTransX(-t.bg_trans, t.bg_trans, seed++),
        RNG_BgTransY(-t.bg_trans, t.bg_trans, seed++),
        RNG_BgScaleTrigger(t.bg_scale_trig_a, t.bg_scale_trig_b, t.bg_scale_trig_thr, seed++),
        RNG_BgInitScale(0.8, 1.2, seed++),
        RNG_BgScale(t.bg_scale_a, t.bg_scale_b, seed++),
        RNG_NumberOfFgObjects(16, 24, seed++),
        RNG_ObjType(type_options(t.obj_types), seed++),
        RNG_ObjTexID(0, std::numeric_limits<int>::max(), seed++),
        RNG_ObjInitTransX(-W / 2. - 50, W * 3. / 2. + 50, seed++),
        RNG_ObjInitTransY(-H / 2. - 50, H * 3. / 2. + 50, seed++),
        RNG_ObjTransX(-t.obj_trans, t.obj_trans, seed++),
        RNG_ObjTransY(-t.obj_trans, t.obj_trans, seed++),
        RNG_ObjInitRot(t.obj_init_rot ? -kAggPi : 0, t.obj_init_rot ? kAggPi : 0, seed++),
        RNG_ObjRotTrigger(t.obj_rot_trig_a, t.obj_rot_trig_b, t.obj_rot_trig_thr, seed++),
        RNG_ObjRot(t.obj_rot_deg ? -t.obj_rot_deg * kAggPi / 180. : 0., t.obj_rot_deg * kAggPi / 180., seed++),
        RNG_ObjInitScale(0.2, 2.5, 0.8, 0.8, seed++),
        RNG_ObjScaleTrigger(t.obj_scale_trig_a, t.obj_scale_trig_b, t.obj_scale_trig_thr, seed++),
        RNG_ObjScale(t.obj_scale_a, t.obj_scale_b, seed++),
        RNG_ObjTexShiftX(-W / 2, W / 2, seed++),
        RNG_ObjTexShiftY(-W / 2, W / 2, seed++),
        RNG_ObjTexRot(-kAggPi, kAggPi, seed++),
        RNG_ObjTexZoom(0.5, 2.0, seed++),
        RNG_ElliObj_ScaleX(0.5, 2, seed++),
        RNG_ElliObj_ScaleY(0.5, 2, seed++),
        RNG_PolyObj_spokes(3, 20, seed++),
        RNG_PolyObj_dphi(-10, 10, seed++),
        RNG_PolyObj_r(20, 80, seed++),
        RNG_PolyObj_ScaleX(0.5, 2, seed++),
        RNG_PolyObj_ScaleY(0.5, 2, seed++),
        RNG_PolyObj_CurveTrigger(0, 1, 0.33, seed++),
        RNG_CompObjInitTransX(-15, 15, seed++),
        RNG_CompObjInitTransY(-15, 15, seed++),
        RNG_CompObiNumberOfComponents(1, 7, seed++),
        RNG_ComponentIsAdditive(0, 1, 0.5, seed++),
        RNG_ComponentOffset(-20, 20, seed++),
        RNG_ObjIsExtraThin(0, 1, 0.2, seed++),
        RNG_ObjDeformsNonrigidly(0, 1, t.deform_thr, seed++),
        RNG_GenericUniform(0, 1, seed++),
        RNG_GenericTrigger(0, 1, 0.5, seed++) {}

  static void clear(ofdg_blueprint* b) {
    *b = ofdg_blueprint();
    b->obj_type = OFDG_OBJ_DUMMY;  // ObjectBlueprint ctor, DG:930-932
  }

  void generateBackground(ofdg_blueprint* b) {  // DG:2105-2143
    b->rot = (RNG_BgRotTrigger() ? RNG_BgRot() : 0.);
    b->scale = (RNG_BgScaleTrigger() ? RNG_BgScale() : 1.);
    float pre_transx = RNG_BgTransX();
    float pre_transy = RNG_BgTransY();
    b->trans_x = std::cos(-b->rot) * pre_transx - std::sin(-b->rot) * pre_transy;
    b->trans_y = std::sin(-b->rot) * pre_transx + std::cos(-b->rot) * pre_transy;
    b->tex_id = RNG_BgTexID();
    b->tex_rot = RNG_BgInitRot();
    b->tex_scale = RNG_BgInitScale();
    b->tex_shift_x = RNG_BgInitTransX();
    b->tex_shift_y = RNG_BgInitTransY();
    b->do_warpfield_deformation = RNG_ObjDeformsNonrigidly();
  }

  int generateNumberOfFgObjects() {  // DG:2832-2835 (float -> int truncation)
    return RNG_NumberOfFgObjects();
  }

  void polygon(ofdg_blueprint* b, bool curves) {  // DG:2208-2228 / 2289-2315
    const unsigned int spokes = static_cast<unsigned int>(RNG_PolyObj_spokes());
    std::vector<float> phi(spokes), r(spokes);
    for (unsigned int i = 0; i < spokes; ++i) {
      phi[i] = (i * 360. / spokes + RNG_PolyObj_dphi()) * kAggPi / 180.;
      r[i] = RNG_PolyObj_r();
    }
    const float xscale = RNG_PolyObj_ScaleX();
    const float yscale = RNG_PolyObj_ScaleY();
    b->n_segments = spokes;
    for (unsigned int i = 0; i < spokes; ++i) {
      b->segment_x[i] = xscale * r[i] * std::cos(phi[i]);
      b->segment_y[i] = yscale * r[i] * std::sin(phi[i]);
    }
    b->segment_type[0] = OFDG_SEG_DUMMY;
    for (unsigned int i = 1; i < spokes; ++i) {
      if (curves and (i < spokes - 1) and RNG_PolyObj_CurveTrigger()) {
        b->segment_type[i] = OFDG_SEG_CURVE3;
        b->segment_type[i + 1] = OFDG_SEG_DUMMY;
        ++i;
      } else {
        b->segment_type[i] = OFDG_SEG_LINE;
      }
    }
  }

  void common_head(ofdg_blueprint* b) {  // e.g. DG:2150-2160
    b->init_rot = RNG_ObjInitRot();
    b->init_trans_x = RNG_ObjInitTransX();
    b->init_trans_y = RNG_ObjInitTransY();
    b->rot = (RNG_ObjRotTrigger() ? RNG_ObjRot() : 0.);
    b->scale = (RNG_ObjScaleTrigger() ? RNG_ObjScale() : 1.);
    b->trans_x = RNG_ObjTransX();
    b->trans_y = RNG_ObjTransY();
    b->tex_id = RNG_ObjTexID();
  }

  // Blueprints live in `pool`; `bi` is the index of the one to fill.  Component
  // blueprints are appended to the pool.
  void generateForegroundObject(std::vector<ofdg_blueprint>& pool, size_t bi) {  // DG:2145-2830
    switch (MODE) {
      case 1: {  // DG:2148-2190
        ofdg_blueprint* b = &pool[bi];
        b->obj_type = RNG_ObjType();
        common_head(b);
        const float radius = RNG_PolyObj_r();
        const float xscale = radius * RNG_PolyObj_ScaleX();
        const float yscale = radius * RNG_PolyObj_ScaleY();
        b->n_segments = 4;
        b->segment_x[0] = xscale;  b->segment_x[1] = xscale;
        b->segment_x[2] = -xscale; b->segment_x[3] = -xscale;
        b->segment_y[0] = -yscale; b->segment_y[1] = yscale;
        b->segment_y[2] = yscale;  b->segment_y[3] = -yscale;
        b->segment_type[0] = OFDG_SEG_DUMMY;
        for (int i = 1; i < 4; ++i) b->segment_type[i] = OFDG_SEG_LINE;
        break;
      }
      case 2: {  // DG:2191-2236
        ofdg_blueprint* b = &pool[bi];
        b->obj_type = RNG_ObjType();
        common_head(b);
        polygon(b, false);
        break;
      }
      case 3: {  // DG:2237-2263
        ofdg_blueprint* b = &pool[bi];
        b->obj_type = RNG_ObjType();
        common_head(b);
        b->ellipse_scale_x = RNG_ElliObj_ScaleX() * 50;
        b->ellipse_scale_y = RNG_ElliObj_ScaleY() * 50;
        break;
      }
      case 4: case 5: case 8: {  // DG:2264-2323
        ofdg_blueprint* b = &pool[bi];
        b->obj_type = RNG_ObjType();
        common_head(b);
        if (b->obj_type == OFDG_OBJ_ELLIPSE) {
          b->ellipse_scale_x = RNG_ElliObj_ScaleX() * 50;
          b->ellipse_scale_y = RNG_ElliObj_ScaleY() * 50;
        } else if (b->obj_type == OFDG_OBJ_POLYGON) {
          polygon(b, true);
        } else {
          throw std::runtime_error("Bad object type");
        }
        break;
      }
      case 6: case 7: case 9: case 10: case 11: case 12: case 13:
        composite_modes(pool, bi);
        break;
      default: throw std::runtime_error("BAD MODE");  // DG:2769
    }
  }

  // DG:2324-2434 (mode 6), DG:2435-2600 (7, 10-13), DG:2601-2767 (9)
  void composite_modes(std::vector<ofdg_blueprint>& pool, size_t bi) {
    const bool thin_modes = (MODE != 6);
    const bool is_component = (pool[bi].obj_type == OFDG_OBJ_COMPOSITE);
    {
      ofdg_blueprint* b = &pool[bi];
      do {
        b->obj_type = RNG_ObjType();
      } while (is_component and (b->obj_type == OFDG_OBJ_COMPOSITE));
      common_head(b);
      if (MODE == 9) b->do_warpfield_deformation = RNG_ObjDeformsNonrigidly();  // DG:2619
      if (b->obj_type == OFDG_OBJ_ELLIPSE) {
        b->ellipse_scale_x = RNG_ElliObj_ScaleX() * 50;
        b->ellipse_scale_y = RNG_ElliObj_ScaleY() * 50;
        if (thin_modes and not is_component and RNG_ObjIsExtraThin()) b->ellipse_scale_x *= 0.05;  // DG:2462
        return;
      }
      if (b->obj_type == OFDG_OBJ_POLYGON) {
        polygon(b, true);
        if (thin_modes and not is_component and RNG_ObjIsExtraThin()) {  // DG:2496
          for (int i = 0; i < b->n_segments; ++i) b->segment_x[i] *= 0.05;
        }
        return;
      }
    }
    // Composite
    auto inherit = [&](ofdg_blueprint& c, const ofdg_blueprint& b) {  // DG:2556-2563
      c.init_rot = b.init_rot; c.init_trans_x = b.init_trans_x; c.init_trans_y = b.init_trans_y;
      c.rot = b.rot; c.scale = b.scale; c.trans_x = b.trans_x; c.trans_y = b.trans_y;
    };
    auto new_component = [&]() -> size_t {
      ofdg_blueprint c;
      clear(&c);
      c.obj_type = OFDG_OBJ_COMPOSITE;  // pre-mark as component
      pool.push_back(c);
      size_t ci = pool.size() - 1;
      generateForegroundObject(pool, ci);  // "Prefill values"
      return ci;
    };
    std::vector<size_t> comps;
    if (thin_modes and RNG_ObjIsExtraThin()) {  // DG:2504-2547 / DG:2668-2713 "outline"
      size_t c1 = new_component();
      inherit(pool[c1], pool[bi]);
      pool[c1].is_additive_component = true;
      if (MODE == 9) pool[c1].do_warpfield_deformation = pool[bi].do_warpfield_deformation;
      comps.push_back(c1);
      const ofdg_blueprint c1_copy = pool[c1];
      pool.push_back(c1_copy);  // ObjectBlueprint(*c1), DG:2520
      size_t c2 = pool.size() - 1;
      const ofdg_blueprint b = pool[bi];
      ofdg_blueprint& C2 = pool[c2];
      if (pool[c1].obj_type == OFDG_OBJ_ELLIPSE) {
        if (RNG_GenericTrigger()) {
          C2.init_trans_x = b.init_trans_x + RNG_CompObjInitTransX();
          C2.init_trans_y = b.init_trans_y + RNG_CompObjInitTransY();
        } else {
          C2.init_trans_x = b.init_trans_x;
          C2.init_trans_y = b.init_trans_y;
          C2.ellipse_scale_x *= 0.9;
          C2.ellipse_scale_y *= 0.9;
        }
      } else {
        C2.init_trans_x = b.init_trans_x;
        C2.init_trans_y = b.init_trans_y;
        for (int si = 0; si < C2.n_segments; ++si) {
          C2.segment_x[si] *= 0.9;
          C2.segment_y[si] *= 0.9;
        }
      }
      C2.scale = b.scale; C2.rot = b.rot; C2.trans_x = b.trans_x; C2.trans_y = b.trans_y;
      C2.is_additive_component = false;
      if (MODE == 9) C2.do_warpfield_deformation = b.do_warpfield_deformation;
      comps.push_back(c2);
    } else {  // DG:2384-2426 / DG:2549-2591 / DG:2715-2758
      const unsigned int parts = RNG_CompObiNumberOfComponents();
      for (unsigned int part_idx = 0; part_idx < parts; ++part_idx) {
        size_t ci = new_component();
        inherit(pool[ci], pool[bi]);
        ofdg_blueprint& c = pool[ci];
        if (part_idx == 0) {
          c.is_additive_component = true;
        } else {
          c.init_rot = RNG_ObjInitRot();
          c.init_trans_x += RNG_ComponentOffset();
          c.init_trans_y += RNG_ComponentOffset();
          if (c.obj_type == OFDG_OBJ_ELLIPSE) {
            c.ellipse_scale_x *= 0.2;
            c.ellipse_scale_y *= 0.2;
          } else if (c.obj_type == OFDG_OBJ_POLYGON) {
            for (int si = 0; si < c.n_segments; ++si) {
              c.segment_x[si] *= 0.2;
              c.segment_y[si] *= 0.2;
            }
          } else {
            throw std::runtime_error("Bad component object type");
          }
          c.is_additive_component = RNG_ComponentIsAdditive();
        }
        if (MODE == 9) c.do_warpfield_deformation = pool[bi].do_warpfield_deformation;
        comps.push_back(ci);
      }
    }
    // components of one composite are contiguous?  Not necessarily in creation
    // order with nested prefill, but components are never composites, so each
    // new_component() appends exactly one blueprint (plus the c2 copy).
    pool[bi].first_component = (int)comps.front();
    pool[bi].n_components = (int)comps.size();
  }

  // One task, the way load_batch drives the sampler (LAY:197-213).
  void next_task(std::vector<ofdg_blueprint>& pool, ofdg_task* task) {
    ofdg_blueprint bg;
    clear(&bg);
    bg.obj_id = OFDG_BACKGROUND_ID;
    generateBackground(&bg);
    pool.push_back(bg);
    task->background = (int)pool.size() - 1;
    int fg_objs = generateNumberOfFgObjects();
    if (num_objects_override > 0) fg_objs = num_objects_override;
    task->first_object = (int)pool.size();
    task->n_objects = fg_objs;
    task->reserved = 0;
    for (int i = 0; i < fg_objs; ++i) {
      ofdg_blueprint b;
      clear(&b);
      b.obj_id = i + 10;
      pool.push_back(b);
    }
    for (int i = 0; i < fg_objs; ++i) generateForegroundObject(pool, task->first_object + i);
  }
};

}  // namespace oracle
