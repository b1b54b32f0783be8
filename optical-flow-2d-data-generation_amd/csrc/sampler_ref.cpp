#include "sampler_ref.h"

#include <cmath>
#include <limits>

namespace ofdg {

// ---------------------------------------------------------------------------
// mt19937 (32-bit Mersenne Twister, std::mt19937 parameters)
// ---------------------------------------------------------------------------
void Mt19937::reseed(uint32_t seed) {
  mt_[0] = seed;
  for (int i = 1; i < 624; ++i) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + (uint32_t)i;
  idx_ = 624;
}
void Mt19937::refill() {
  for (int i = 0; i < 624; ++i) {
    const uint32_t y = (mt_[i] & 0x80000000u) | (mt_[(i + 1) % 624] & 0x7fffffffu);
    mt_[i] = mt_[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  idx_ = 0;
}
uint32_t Mt19937::next() {
  if (idx_ >= 624) refill();
  uint32_t y = mt_[idx_++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// ---------------------------------------------------------------------------
// libstdc++ (GCC 11) distribution algorithms over a 32-bit engine
// ---------------------------------------------------------------------------
// std::uniform_int_distribution<int>(a, b): Lemire's nearly-divisionless method.
int RefSampler::draw_int(Stream& s, int a, int b) {
  const uint32_t urange = (uint32_t)b - (uint32_t)a;
  if (urange == 0xffffffffu) return (int)((uint32_t)a + s.eng.next());
  const uint32_t range = urange + 1u;
  uint64_t product = (uint64_t)s.eng.next() * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)s.eng.next() * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (int)((uint32_t)a + (uint32_t)(product >> 32));
}

// std::uniform_real_distribution<double>(a, b) on floats widened to double, the
// result narrowed to float (SimpleRandom.h:95-109).
// generate_canonical<double, 53>: two 32-bit draws.
float RefSampler::draw_uniform(Stream& s) {
  const double r = 4294967296.0;
  double sum = (double)s.eng.next();
  sum += (double)s.eng.next() * r;
  double u = sum / (r * r);
  if (u >= 1.0) u = std::nextafter(1.0, 0.0);
  const double a = (double)s.a, b = (double)s.b;
  return (float)(u * (b - a) + a);
}

// std::normal_distribution<float>(0, 1): Marsaglia polar, second variate cached.
// generate_canonical<float, 24>: one draw, fp32 arithmetic.
float RefSampler::draw_normal(Stream& s) {
  if (s.saved_avail) {
    s.saved_avail = false;
    return s.saved * 1.0f + 0.0f;
  }
  float x, y, r2;
  do {
    float u1 = (float)s.eng.next() / 4294967296.0f;
    if (u1 >= 1.0f) u1 = std::nextafterf(1.0f, 0.0f);
    float u2 = (float)s.eng.next() / 4294967296.0f;
    if (u2 >= 1.0f) u2 = std::nextafterf(1.0f, 0.0f);
    x = (float)(2.0f * u1 - 1.0);
    y = (float)(2.0f * u2 - 1.0);
    r2 = x * x + y * y;
  } while (r2 > 1.0 || r2 == 0.0);
  const float mult = std::sqrt(-2 * std::log(r2) / r2);
  s.saved = x * mult;
  s.saved_avail = true;
  return (y * mult) * 1.0f + 0.0f;
}

// FlyingChairsRandom::GaussianSq / Gaussian3 / Gaussian4 + baseGauss
// (DataGenerator.cpp:828-911): n -> sign-preserving n^power, squeezed into [a, b];
// samples falling outside map to the midpoint.
float RefSampler::draw_gauss_pow(Stream& s, int power, float normalize) {
  float tmp = draw_normal(s);
  if (power == 3) {
    tmp = (float)std::pow((double)tmp, 3.0);
  } else {
    const double p = std::pow((double)tmp, (double)power);
    tmp = (float)((tmp > 0) ? p : -p);
  }
  const float a = s.a, b = s.b;
  const float sample = tmp * ((b + a) / 2.f - a) / normalize + (b + a) / 2.f;
  return (float)((a <= sample && sample <= b) ? (double)sample : (double)(b + a) / 2.);
}

// ---------------------------------------------------------------------------
// the 13 modes (DataGenerator.cpp:1363-2001) as deltas to mode 7
// ---------------------------------------------------------------------------
RefSampler::RefSampler(int mode, int W, int H, int num_objects_override)
    : mode_(mode), W_(W), H_(H), num_objects_(num_objects_override) {
  if (mode < 1 || mode > 13) return;  // "BAD MODE"
  for (int i = 0; i < kNumStreams; ++i) st_[i].eng.reseed((uint32_t)i);
  const double pi = 3.14159265358979323846;  // agg::pi
  auto range = [&](int id, double a, double b) { st_[id].a = (float)a; st_[id].b = (float)b; };
  auto trig = [&](int id, bool enabled, double thr) {
    // a disabled motion is Trigger(0, 0, 1): always fires, still consumes its draws
    st_[id].a = 0.f; st_[id].b = enabled ? 1.f : 0.f; st_[id].thr = enabled ? (float)thr : 1.f;
  };
  // motion magnitudes: {bg rot deg, bg trans, bg scale lo/hi, obj trans, obj rot deg, obj scale lo/hi,
  //                     trigger thresholds bg rot / bg scale / obj rot / obj scale}
  struct Mag { double bg_rot, bg_trans, bg_s0, bg_s1, obj_trans, obj_rot, obj_s0, obj_s1, t_bgr, t_bgs, t_or, t_os; };
  Mag g{10, 40, 0.93, 1.07, 120, 30, 0.8, 1.2, 0.3, 0.6, 0.7, 0.7};
  if (mode == 10) g = Mag{5, 20, 0.965, 1.035, 60, 15, 0.9, 1.1, 0.176, 0.429, 0.539, 0.539};
  if (mode == 11) g = Mag{20, 80, 0.86, 1.14, 240, 60, 0.6, 1.4, 0.462, 0.75, 0.824, 0.824};
  if (mode == 12) g = Mag{3.3, 13.3, 0.976, 1.023, 40, 10, 0.933, 1.066, 0.125, 0.333, 0.437, 0.437};
  if (mode == 13) g = Mag{30, 120, 0.79, 1.21, 360, 90, 0.4, 1.6, 0.563, 0.818, 0.875, 0.875};
  const bool translation_only = (mode == 1 || mode == 2 || mode == 3 || mode == 8);
  const bool bg_rot = !translation_only;
  const bool bg_scale = !translation_only && mode != 4;
  const bool obj_rot = !translation_only;
  const bool obj_scale = !translation_only && mode != 4;

  range(kBgInitRot, -pi, pi);
  trig(kBgRotTrigger, bg_rot, g.t_bgr);
  if (bg_rot) range(kBgRot, -g.bg_rot * pi / 180., g.bg_rot * pi / 180.); else range(kBgRot, 0, 0);
  range(kBgTransX, -g.bg_trans, g.bg_trans);
  range(kBgTransY, -g.bg_trans, g.bg_trans);
  trig(kBgScaleTrigger, bg_scale, g.t_bgs);
  range(kBgInitScale, 0.8, 1.2);
  if (bg_scale) range(kBgScale, g.bg_s0, g.bg_s1); else range(kBgScale, 1, 1);
  range(kNumberOfFgObjects, 16, 24);
  switch (mode) {
    case 1: case 2: type_mask_ = 2; break;
    case 3: type_mask_ = 1; break;
    case 4: case 5: case 8: type_mask_ = 3; break;
    default: type_mask_ = 7; break;
  }
  if (type_mask_ & 1) types_[n_types_++] = OFDG_OBJ_ELLIPSE;
  if (type_mask_ & 2) types_[n_types_++] = OFDG_OBJ_POLYGON;
  if (type_mask_ & 4) types_[n_types_++] = OFDG_OBJ_COMPOSITE;
  range(kObjInitTransX, -W / 2. - 50, W * 3. / 2. + 50);
  range(kObjInitTransY, -H / 2. - 50, H * 3. / 2. + 50);
  range(kObjTransX, -g.obj_trans, g.obj_trans);
  range(kObjTransY, -g.obj_trans, g.obj_trans);
  if (mode == 1) range(kObjInitRot, 0, 0); else range(kObjInitRot, -pi, pi);
  trig(kObjRotTrigger, obj_rot, g.t_or);
  if (obj_rot) range(kObjRot, -g.obj_rot * pi / 180., g.obj_rot * pi / 180.); else range(kObjRot, 0, 0);
  trig(kObjScaleTrigger, obj_scale, g.t_os);
  if (obj_scale) range(kObjScale, g.obj_s0, g.obj_s1); else range(kObjScale, 1, 1);
  range(kElliScaleX, 0.5, 2);
  range(kElliScaleY, 0.5, 2);
  range(kPolyDphi, -10, 10);
  range(kPolyR, 20, 80);
  range(kPolyScaleX, 0.5, 2);
  range(kPolyScaleY, 0.5, 2);
  trig(kPolyCurveTrigger, true, 0.33);
  range(kCompInitTransX, -15, 15);
  range(kCompInitTransY, -15, 15);
  trig(kComponentIsAdditive, true, 0.5);
  range(kComponentOffset, -20, 20);
  trig(kObjIsExtraThin, true, 0.2);
  trig(kObjDeformsNonrigidly, true, mode == 9 ? 0.2 : 0.0);
  range(kGenericUniform, 0, 1);
  trig(kGenericTrigger, true, 0.5);
  // kObjInitScale, kObjTexShiftX/Y, kObjTexRot, kObjTexZoom are constructed by the
  // reference but never drawn from (DataGenerator.cpp:1678-1684): nothing to set up.
  ok_ = true;
}

// generateBackground (DataGenerator.cpp:2105-2143)
void RefSampler::background(ofdg_blueprint* b) {
  b->rot = draw_trigger(st_[kBgRotTrigger]) ? draw_gauss_pow(st_[kBgRot], 2, 6) : 0.f;
  b->scale = draw_trigger(st_[kBgScaleTrigger]) ? draw_gauss_pow(st_[kBgScale], 2, 6) : 1.f;
  const float px = draw_gauss_pow(st_[kBgTransX], 4, 15);
  const float py = draw_gauss_pow(st_[kBgTransY], 4, 15);
  // translation pre-rotated by -rot, fp32 (std::cos(float))
  b->trans_x = std::cos(-b->rot) * px - std::sin(-b->rot) * py;
  b->trans_y = std::sin(-b->rot) * px + std::cos(-b->rot) * py;
  b->tex_id = draw_int(st_[kBgTexID], 0, std::numeric_limits<int>::max());
  b->tex_rot = draw_uniform(st_[kBgInitRot]);
  b->tex_scale = draw_uniform(st_[kBgInitScale]);
  b->tex_shift_x = draw_int(st_[kBgInitTransX], 0, 1) ? W_ : 0;
  b->tex_shift_y = draw_int(st_[kBgInitTransY], 0, 1) ? H_ : 0;
  b->do_warpfield_deformation = draw_trigger(st_[kObjDeformsNonrigidly]) ? 1 : 0;
}

// the block every mode starts a foreground object with (e.g. DataGenerator.cpp:2446-2455)
void RefSampler::motion_and_texture(ofdg_blueprint* b) {
  b->init_rot = draw_uniform(st_[kObjInitRot]);
  b->init_trans_x = draw_uniform(st_[kObjInitTransX]);
  b->init_trans_y = draw_uniform(st_[kObjInitTransY]);
  b->rot = draw_trigger(st_[kObjRotTrigger]) ? draw_gauss_pow(st_[kObjRot], 2, 6) : 0.f;
  b->scale = draw_trigger(st_[kObjScaleTrigger]) ? draw_gauss_pow(st_[kObjScale], 2, 6) : 1.f;
  b->trans_x = draw_gauss_pow(st_[kObjTransX], 3, 10);
  b->trans_y = draw_gauss_pow(st_[kObjTransY], 3, 10);
  b->tex_id = draw_int(st_[kObjTexID], 0, std::numeric_limits<int>::max());
}

// star-shaped polygon with optional quadratic curve segments (DataGenerator.cpp:2469-2495)
void RefSampler::star_polygon(ofdg_blueprint* b, bool with_curves) {
  const double pi = 3.14159265358979323846;
  const unsigned spokes = (unsigned)draw_int(st_[kPolySpokes], 3, 20);
  float phi[OFDG_MAX_SEGMENTS], r[OFDG_MAX_SEGMENTS];
  for (unsigned i = 0; i < spokes; ++i) {
    phi[i] = (float)((i * 360. / spokes + draw_uniform(st_[kPolyDphi])) * pi / 180.);
    r[i] = draw_uniform(st_[kPolyR]);
  }
  const float xs = draw_uniform(st_[kPolyScaleX]);
  const float ys = draw_uniform(st_[kPolyScaleY]);
  b->n_segments = (int)spokes;
  for (unsigned i = 0; i < spokes; ++i) {
    b->segment_x[i] = xs * r[i] * std::cos(phi[i]);
    b->segment_y[i] = ys * r[i] * std::sin(phi[i]);
  }
  b->segment_type[0] = OFDG_SEG_DUMMY;
  unsigned i = 1;
  while (i < spokes) {
    if (with_curves && i < spokes - 1 && draw_trigger(st_[kPolyCurveTrigger])) {
      b->segment_type[i] = OFDG_SEG_CURVE3;
      b->segment_type[i + 1] = OFDG_SEG_DUMMY;
      i += 2;
    } else {
      b->segment_type[i] = OFDG_SEG_LINE;
      i += 1;
    }
  }
}

static void scale_shape(ofdg_blueprint* c, double f) {  // "*= 0.2" / "*= 0.9" on float fields
  if (c->obj_type == OFDG_OBJ_ELLIPSE) {
    c->ellipse_scale_x = (float)(c->ellipse_scale_x * f);
    c->ellipse_scale_y = (float)(c->ellipse_scale_y * f);
  } else {
    for (int i = 0; i < c->n_segments; ++i) {
      c->segment_x[i] = (float)(c->segment_x[i] * f);
      c->segment_y[i] = (float)(c->segment_y[i] * f);
    }
  }
}

// generateForegroundObject (DataGenerator.cpp:2145-2830).  `is_component` marks the
// recursive "prefill" call for a part of a composite (the reference pre-sets
// obj_type = Composite as the marker, :2441, :2506).
int RefSampler::foreground(std::vector<ofdg_blueprint>* bps, size_t bi, bool is_component, std::string* msg) {
  const bool has_thin = (mode_ == 7 || mode_ >= 9);
  const bool curves = (mode_ >= 4);
  int type;
  do {
    type = types_[draw_int(st_[kObjType], 0, n_types_ - 1)];
  } while (is_component && type == OFDG_OBJ_COMPOSITE);
  {
    ofdg_blueprint* b = &(*bps)[bi];
    b->obj_type = type;
    motion_and_texture(b);
    if (mode_ == 9) b->do_warpfield_deformation = draw_trigger(st_[kObjDeformsNonrigidly]) ? 1 : 0;
    if (mode_ == 1) {  // axis-aligned box (DataGenerator.cpp:2165-2182)
      const float radius = draw_uniform(st_[kPolyR]);
      const float xs = radius * draw_uniform(st_[kPolyScaleX]);
      const float ys = radius * draw_uniform(st_[kPolyScaleY]);
      b->n_segments = 4;
      const float bx[4] = {xs, xs, -xs, -xs}, by[4] = {-ys, ys, ys, -ys};
      for (int i = 0; i < 4; ++i) {
        b->segment_x[i] = bx[i];
        b->segment_y[i] = by[i];
        b->segment_type[i] = i ? OFDG_SEG_LINE : OFDG_SEG_DUMMY;
      }
      return OFDG_OK;
    }
    if (type == OFDG_OBJ_ELLIPSE) {
      b->ellipse_scale_x = draw_uniform(st_[kElliScaleX]) * 50;
      b->ellipse_scale_y = draw_uniform(st_[kElliScaleY]) * 50;
      if (has_thin && !is_component && draw_trigger(st_[kObjIsExtraThin]))  // "needle"
        b->ellipse_scale_x = (float)(b->ellipse_scale_x * 0.05);
      return OFDG_OK;
    }
    if (type == OFDG_OBJ_POLYGON) {
      star_polygon(b, curves);
      if (has_thin && !is_component && draw_trigger(st_[kObjIsExtraThin]))
        for (int i = 0; i < b->n_segments; ++i) b->segment_x[i] = (float)(b->segment_x[i] * 0.05);
      return OFDG_OK;
    }
  }
  // ---- composite (modes 6, 7, 9-13) ----
  auto add_component = [&](size_t* ci) -> int {
    ofdg_blueprint c = ofdg_blueprint();
    bps->push_back(c);
    *ci = bps->size() - 1;
    int rc = foreground(bps, *ci, true, msg);  // prefill; most fields are overwritten below
    if (rc != OFDG_OK) return rc;
    ofdg_blueprint& C = (*bps)[*ci];
    const ofdg_blueprint& B = (*bps)[bi];
    C.init_rot = B.init_rot; C.init_trans_x = B.init_trans_x; C.init_trans_y = B.init_trans_y;
    C.rot = B.rot; C.scale = B.scale; C.trans_x = B.trans_x; C.trans_y = B.trans_y;
    if (mode_ == 9) C.do_warpfield_deformation = B.do_warpfield_deformation;
    return OFDG_OK;
  };
  const int first = (int)bps->size();
  int count = 0;
  if (has_thin && draw_trigger(st_[kObjIsExtraThin])) {
    // "outline": a shape minus a slightly smaller / shifted copy (DataGenerator.cpp:2504-2547)
    size_t c1;
    int rc = add_component(&c1);
    if (rc != OFDG_OK) return rc;
    (*bps)[c1].is_additive_component = 1;
    ofdg_blueprint c2 = (*bps)[c1];
    const ofdg_blueprint B = (*bps)[bi];
    c2.init_trans_x = B.init_trans_x;
    c2.init_trans_y = B.init_trans_y;
    if (c2.obj_type == OFDG_OBJ_ELLIPSE) {
      if (draw_trigger(st_[kGenericTrigger])) {
        c2.init_trans_x = B.init_trans_x + draw_uniform(st_[kCompInitTransX]);
        c2.init_trans_y = B.init_trans_y + draw_uniform(st_[kCompInitTransY]);
      } else {
        scale_shape(&c2, 0.9);
      }
    } else {
      scale_shape(&c2, 0.9);
    }
    c2.is_additive_component = 0;
    bps->push_back(c2);
    count = 2;
  } else {
    const unsigned parts = (unsigned)draw_int(st_[kCompNumberOfComponents], 1, 7);
    for (unsigned k = 0; k < parts; ++k) {
      size_t ci;
      int rc = add_component(&ci);
      if (rc != OFDG_OK) return rc;
      ofdg_blueprint& C = (*bps)[ci];
      if (k == 0) {
        C.is_additive_component = 1;
      } else {  // small satellite part: own rotation, +-20 px offset, 0.2x size, random add/subtract
        C.init_rot = draw_uniform(st_[kObjInitRot]);
        C.init_trans_x += draw_uniform(st_[kComponentOffset]);
        C.init_trans_y += draw_uniform(st_[kComponentOffset]);
        scale_shape(&C, 0.2);
        C.is_additive_component = draw_trigger(st_[kComponentIsAdditive]) ? 1 : 0;
      }
      ++count;
    }
  }
  (*bps)[bi].first_component = first;
  (*bps)[bi].n_components = count;
  return OFDG_OK;
}

int RefSampler::next_task(std::vector<ofdg_blueprint>* bps, ofdg_task* task, std::string* msg) {
  if (!ok_) { *msg = "BAD MODE"; return OFDG_EBADMODE; }
  ofdg_blueprint bg = ofdg_blueprint();
  bg.obj_id = OFDG_BACKGROUND_ID;  // data_generation_layer.cpp:201
  background(&bg);
  bps->push_back(bg);
  task->background = (int)bps->size() - 1;
  int n = (int)draw_uniform(st_[kNumberOfFgObjects]);  // float -> int (DataGenerator.cpp:2832-2835)
  if (num_objects_ > 0) n = num_objects_;
  task->first_object = (int)bps->size();
  task->n_objects = n;
  task->reserved = 0;
  for (int i = 0; i < n; ++i) {
    ofdg_blueprint b = ofdg_blueprint();
    b.obj_id = i + 10;  // data_generation_layer.cpp:210
    bps->push_back(b);
  }
  for (int i = 0; i < n; ++i) {
    int rc = foreground(bps, (size_t)task->first_object + i, false, msg);
    if (rc != OFDG_OK) return rc;
  }
  return OFDG_OK;
}

}  // namespace ofdg
