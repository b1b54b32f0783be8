// Host "reference-stream" blueprint sampler (OFDG_SAMPLER_REF).
//
// Produces, draw for draw, the blueprints the reference's
// ObjectParametersGenerator produces (src/caffe/DataGenerator.cpp:1358-2835,
// driven like load_batch, src/caffe/layers/data_generation_layer.cpp:197-213):
// 45 independent mt19937 streams seeded 0..44 in declaration order.
//
// The generator and the three libstdc++ (GCC 11) distribution algorithms the
// reference instantiates through include/caffe/data_generation/SimpleRandom.h
// are written out here (mt19937 tempering, Lemire's bounded integers,
// generate_canonical<double,53> / <float,24>, Marsaglia's polar method with a
// cached second variate), so the stream does not depend on the C++ runtime the
// library is built against and can be moved to other hosts unchanged.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/ofdg.h"

namespace ofdg {

class Mt19937 {
 public:
  explicit Mt19937(uint32_t seed = 5489u) { reseed(seed); }
  void reseed(uint32_t seed);
  uint32_t next();

 private:
  void refill();
  uint32_t mt_[624];
  int idx_;
};

enum StreamId {
  kBgTexID = 0, kBgInitRot, kBgInitTransX, kBgInitTransY, kBgRotTrigger, kBgRot, kBgTransX, kBgTransY,
  kBgScaleTrigger, kBgInitScale, kBgScale, kNumberOfFgObjects, kObjType, kObjTexID, kObjInitTransX,
  kObjInitTransY, kObjTransX, kObjTransY, kObjInitRot, kObjRotTrigger, kObjRot, kObjInitScale,
  kObjScaleTrigger, kObjScale, kObjTexShiftX, kObjTexShiftY, kObjTexRot, kObjTexZoom, kElliScaleX,
  kElliScaleY, kPolySpokes, kPolyDphi, kPolyR, kPolyScaleX, kPolyScaleY, kPolyCurveTrigger,
  kCompInitTransX, kCompInitTransY, kCompNumberOfComponents, kComponentIsAdditive, kComponentOffset,
  kObjIsExtraThin, kObjDeformsNonrigidly, kGenericUniform, kGenericTrigger, kNumStreams
};

class RefSampler {
 public:
  // Throws nothing; check ok() (bad mode => "BAD MODE", DataGenerator.cpp:2004).
  RefSampler(int mode, int W, int H, int num_objects_override);
  bool ok() const { return ok_; }

  // Appends one task (background + foreground objects + their components).
  // Returns OFDG_OK or an error code.
  int next_task(std::vector<ofdg_blueprint>* bps, ofdg_task* task, std::string* msg);

 private:
  struct Stream {
    Mt19937 eng;
    float a = 0, b = 0;     // range of uniform / gaussian-shaped streams
    float thr = 0;          // trigger threshold
    bool saved_avail = false;
    float saved = 0;        // normal_distribution's cached variate
  };
  // distributions
  int draw_int(Stream& s, int a, int b);
  float draw_uniform(Stream& s);             // FixedRangeUniformFloat(a, b)
  float draw_normal(Stream& s);              // normal_distribution<float>(0, 1)
  bool draw_trigger(Stream& s) { return draw_uniform(s) < s.thr; }
  float draw_gauss_pow(Stream& s, int power, float normalize);  // GaussianSq / Gaussian3 / Gaussian4

  void background(ofdg_blueprint* b);
  int foreground(std::vector<ofdg_blueprint>* bps, size_t bi, bool is_component, std::string* msg);
  void motion_and_texture(ofdg_blueprint* b);
  void star_polygon(ofdg_blueprint* b, bool with_curves);

  int mode_, W_, H_, num_objects_;
  bool ok_ = false;
  unsigned type_mask_ = 0;  // bit0 ellipse, bit1 polygon, bit2 composite
  int n_types_ = 0;
  int types_[3];
  Stream st_[kNumStreams];
};

}  // namespace ofdg
