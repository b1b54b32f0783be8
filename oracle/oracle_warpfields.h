// TEST INFRASTRUCTURE -- parity oracle, not product code.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// CPU restatement of the mode-9 warp-field math: src/caffe/WarpFields.cpp (WF)
// of the reference plus the CImg accessors it calls (linear_atXY, _linear_atXY,
// resize(...,3)) -- CImg is an un-vendored, un-pinned (">= 2.0.0") dependency of
// the reference; its arithmetic is restated from the published source of CImg 2.x:
// "parity unpinned".
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "oracle_core.h"

namespace oracle {

// CImg<unsigned char>::linear_atXY(fx,fy,0,c,out_value=0) -> (unsigned char) (Dirichlet).
inline uint8_t cimg_linear_atXY_dirichlet_u8(const uint8_t* img, int w, int h, float fx, float fy) {
  if (!(fx == fx) || !(fy == fy)) return 0;  // NaN displacement: defined as "outside" (SURVEY F-9)
  if (fx < -2.f || fy < -2.f || fx > w + 2.f || fy > h + 2.f) return 0;  // all four taps outside
  const int x = (int)fx - (fx >= 0 ? 0 : 1), nx = x + 1;
  const int y = (int)fy - (fy >= 0 ? 0 : 1), ny = y + 1;
  const float dx = fx - x, dy = fy - y;
  auto at = [&](int xx, int yy) -> float {
    return (xx < 0 || yy < 0 || xx >= w || yy >= h) ? 0.f : (float)img[(size_t)yy * w + xx];
  };
  const float Icc = at(x, y), Inc = at(nx, y), Icn = at(x, ny), Inn = at(nx, ny);
  const float v = Icc + dx * (Inc - Icc + dy * (Icc + Inn - Icn - Inc)) + dy * (Icn - Icc);
  return (uint8_t)v;
}

// CImg<float>::linear_atXY(fx,fy,z) == _linear_atXY (Neumann), one plane of w x h.
inline float cimg_linear_atXY_neumann(const float* img, int w, int h, float fx, float fy) {
  const float nfx = fx <= 0 ? 0 : (fx >= w - 1 ? (float)(w - 1) : fx);  // cimg::cut
  const float nfy = fy <= 0 ? 0 : (fy >= h - 1 ? (float)(h - 1) : fy);
  const unsigned int x = (unsigned int)nfx, y = (unsigned int)nfy;
  const float dx = nfx - x, dy = nfy - y;
  const unsigned int nx = dx > 0 ? x + 1 : x, ny = dy > 0 ? y + 1 : y;
  const float Icc = img[(size_t)y * w + x], Inc = img[(size_t)y * w + nx];
  const float Icn = img[(size_t)ny * w + x], Inn = img[(size_t)ny * w + nx];
  return Icc + dx * (Inc - Icc + dy * (Icc + Inn - Icn - Inc)) + dy * (Icn - Icc);
}

struct WarpCrop {
  int w = 0, h = 0;
  std::vector<float> flow, iflow;  // each 2 planes of w*h
};

// CropGenerator::get_crop (WF:516-538): each crop is served reuse_same+1 times.
struct WarpSource {
  std::vector<WarpCrop> crops;
  int reuse, counter = 0;
  size_t head = 0;
  WarpSource(const float* data, int n, int w, int h, int reuse_same) : reuse(reuse_same) {
    const size_t plane2 = (size_t)2 * w * h;
    for (int i = 0; i < n; ++i) {
      WarpCrop c;
      c.w = w; c.h = h;
      c.flow.assign(data + (size_t)i * 2 * plane2, data + (size_t)i * 2 * plane2 + plane2);
      c.iflow.assign(data + (size_t)i * 2 * plane2 + plane2, data + (size_t)(i + 1) * 2 * plane2);
      crops.push_back(c);
    }
  }
  const WarpCrop* get_crop() {
    if (crops.empty()) return nullptr;
    const WarpCrop* c = &crops[head % crops.size()];
    ++counter;
    if (counter > reuse) { ++head; counter = 0; }
    return c;
  }
  // reuse < 0: the blueprint's deformation flag names the crop (flag - 1) instead of the
  // CropGenerator's serving order - the device counter sampler assigns crops that way
  // (a pure function of seed and sample index; ofdg_sample_counter returns flag = 1 + crop).
  const WarpCrop* get_crop_for(int flag) {
    if (reuse >= 0) return get_crop();
    return crops.empty() ? nullptr : &crops[(size_t)(flag - 1) % crops.size()];
  }
};

// CImg<float>::resize(sx,sy,-100,-100,3) (linear, boundary 0, upscaling branch),
// then *= 2  (DataGenerator.cpp:1197-1200).
inline std::vector<float> cimg_resize_linear_plane(const std::vector<float>& in, int w, int h, int sx, int sy) {
  // X pass
  std::vector<float> rx((size_t)sx * h);
  {
    const double fx = (sx > w) ? (sx > 1 ? (w - 1.) / (sx - 1) : 0) : (double)w / sx;
    std::vector<unsigned int> off(sx);
    std::vector<double> foff(sx);
    double curr = 0, old = 0;
    for (int x = 0; x < sx; ++x) {
      foff[x] = curr - (unsigned int)curr;
      old = curr;
      curr = std::min(w - 1., curr + fx);
      off[x] = (unsigned int)curr - (unsigned int)old;
    }
    for (int y = 0; y < h; ++y) {
      const float* ptrs = &in[(size_t)y * w];
      const float* ptrsmax = ptrs + w - 1;
      for (int x = 0; x < sx; ++x) {
        const double alpha = foff[x];
        const float val1 = *ptrs, val2 = ptrs < ptrsmax ? *(ptrs + 1) : val1;
        rx[(size_t)y * sx + x] = (float)((1 - alpha) * val1 + alpha * val2);
        ptrs += off[x];
      }
    }
  }
  std::vector<float> ry((size_t)sx * sy);
  {
    const double fy = (sy > h) ? (sy > 1 ? (h - 1.) / (sy - 1) : 0) : (double)h / sy;
    std::vector<unsigned int> off(sy);
    std::vector<double> foff(sy);
    double curr = 0, old = 0;
    for (int y = 0; y < sy; ++y) {
      foff[y] = curr - (unsigned int)curr;
      old = curr;
      curr = std::min(h - 1., curr + fy);
      off[y] = sx * ((unsigned int)curr - (unsigned int)old);
    }
    for (int x = 0; x < sx; ++x) {
      const float* ptrs = &rx[x];
      const float* ptrsmax = ptrs + (size_t)(h - 1) * sx;
      for (int y = 0; y < sy; ++y) {
        const double alpha = foff[y];
        const float val1 = *ptrs, val2 = ptrs < ptrsmax ? *(ptrs + sx) : val1;
        ry[(size_t)y * sx + x] = (float)((1 - alpha) * val1 + alpha * val2);
        ptrs += off[y];
      }
    }
  }
  return ry;
}

inline WarpCrop upscale_warp_for_background(const WarpCrop& c, int sx, int sy) {
  WarpCrop o;
  o.w = sx; o.h = sy;
  const size_t plane = (size_t)c.w * c.h;
  for (int f = 0; f < 2; ++f) {
    const std::vector<float>& src = f ? c.iflow : c.flow;
    std::vector<float>& dst = f ? o.iflow : o.flow;
    for (int ch = 0; ch < 2; ++ch) {
      std::vector<float> p(src.begin() + ch * plane, src.begin() + (ch + 1) * plane);
      std::vector<float> r = cimg_resize_linear_plane(p, c.w, c.h, sx, sy);
      for (float& v : r) v = (float)(v * 2.);  // warpflow *= 2.
      dst.insert(dst.end(), r.begin(), r.end());
    }
  }
  return o;
}

// ---- Supports / Displacers / DisplacementComposer (WF:88-112, 191-260, 296-316) ----
struct DisplacerSpec {
  double type;          // 0 Translation, 1 Rotation, 2 Zoom
  double p0, p1, p2;    // (dx,dy,-) | (cx,cy,omega) | (cx,cy,factor)
  double sup_cx, sup_cy, sup_sx, sup_sy, sup_angle;  // Gaussian2D
};

struct Gaussian2D {  // WF:88-112
  float cx, cy, a, b, c, d, ratio_x_y, sigma_sq, gauss_prefactor, normalizer;
  Gaussian2D(float cx_, float cy_, float sigma_x, float sigma_y, float angle)
      : cx(cx_), cy(cy_), a(std::cos(angle)), b(-std::sin(angle)), c(std::sin(angle)), d(std::cos(angle)),
        ratio_x_y(sigma_x / sigma_y), sigma_sq(sigma_x * sigma_x),
        gauss_prefactor(1 / std::sqrt(2 * M_PI * sigma_sq)), normalizer(0) {
    normalizer = 1 / raw_at(cx, cy);
  }
  float raw_at(float x, float y) const {
    const float rx = a * (x - cx) + b * (y - cy);
    const float ry = (c * (x - cx) + d * (y - cy)) * ratio_x_y;
    const float dist_sq{rx * rx + ry * ry};
    const float arg = -dist_sq / (2 * sigma_sq);
    return gauss_prefactor * (detmath_flag() ? ofdg_det_expf(arg) : std::exp(arg));
  }
  float at(float x, float y) const { return normalizer * raw_at(x, y); }
};

struct Displacer {
  int type;
  float cx, cy;                                            // DisplacerBase
  float dx, dy;                                            // Translation (WF:191-205)
  float omega, sin_omega, cos_omega, sin_nomega, cos_nomega;  // Rotation (WF:211-236)
  float factor, ifactor;                                   // Zoom (WF:242-260)
  Gaussian2D sup;
  explicit Displacer(const DisplacerSpec& s)
      : type((int)s.type), cx(0), cy(0), dx(0), dy(0), omega(0), sin_omega(0), cos_omega(0), sin_nomega(0),
        cos_nomega(0), factor(0), ifactor(0),
        sup((float)s.sup_cx, (float)s.sup_cy, (float)s.sup_sx, (float)s.sup_sy, (float)s.sup_angle) {
    if (type == 0) { dx = (float)s.p0; dy = (float)s.p1; }
    else if (type == 1) {
      cx = (float)s.p0; cy = (float)s.p1; omega = (float)s.p2;
      sin_omega = std::sin(omega); cos_omega = std::cos(omega);
      sin_nomega = std::sin(-omega); cos_nomega = std::cos(-omega);
    } else {
      cx = (float)s.p0; cy = (float)s.p1; factor = (float)s.p2; ifactor = 1. / factor;
    }
  }
  void raw_flow(float x, float y, float* u, float* v) const {
    if (type == 0) { *u = dx; *v = dy; return; }
    const float ddx{x - cx}, ddy{y - cy};
    if (type == 1) {
      const float rot_dx{cos_nomega * ddx - sin_nomega * ddy};
      const float rot_dy{sin_nomega * ddx + cos_nomega * ddy};
      *u = rot_dx - ddx; *v = rot_dy - ddy;
    } else { *u = factor * ddx - ddx; *v = factor * ddy - ddy; }
  }
  void raw_iflow(float x, float y, float* u, float* v) const {
    if (type == 0) { *u = -dx; *v = -dy; return; }
    const float ddx{x - cx}, ddy{y - cy};
    if (type == 1) {
      const float rot_dx{cos_omega * ddx - sin_omega * ddy};
      const float rot_dy{sin_omega * ddx + cos_omega * ddy};
      *u = rot_dx - ddx; *v = rot_dy - ddy;
    } else { *u = ifactor * ddx - ddx; *v = ifactor * ddy - ddy; }
  }
};

// FlowField (WF:337-455).  `iters` is 17 in the reference (WF:366, 406).
struct FlowField {
  int W = 0, H = 0;
  std::vector<float> flow, iflow;  // 2 planes each

  static void compose(std::vector<float>& f, int W, int H, int iters) {
    const size_t n = (size_t)W * H;
    std::vector<float> tmp(f);
    std::vector<uint8_t> flagged(n, 0);
    for (int iter = iters; iter > 0; --iter) {
      std::vector<float>& from = (iter % 2 == 1 ? tmp : f);
      std::vector<float>& to = (iter % 2 == 1 ? f : tmp);
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const float fx = from[(size_t)y * W + x];
          const float fy = from[n + (size_t)y * W + x];
          if (x + fx < 0 or x + fx >= W or y + fy < 0 or y + fy >= H) {
            flagged[(size_t)y * W + x] = 255;
            to[(size_t)y * W + x] = fx;
            to[n + (size_t)y * W + x] = fy;
            continue;
          }
          to[(size_t)y * W + x] = fx + cimg_linear_atXY_neumann(from.data(), W, H, x + fx, y + fy);
          to[n + (size_t)y * W + x] = fy + cimg_linear_atXY_neumann(from.data() + n, W, H, x + fx, y + fy);
        }
    }
    // the last pass (iter == 1) always writes into f, whatever the parity of iters
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        if (x + f[i] < 0 or x + f[i] >= W or y + f[n + i] < 0 or y + f[n + i] >= H) flagged[i] = 255;
        if (flagged[i]) {
          f[i] = std::numeric_limits<float>::quiet_NaN();
          f[n + i] = std::numeric_limits<float>::quiet_NaN();
        }
      }
  }

  void init_from_displacers(int W_, int H_, const std::vector<DisplacerSpec>& specs, int iters) {
    W = W_; H = H_;
    const size_t n = (size_t)W * H;
    flow.assign(2 * n, 0.f);
    iflow.assign(2 * n, 0.f);
    std::vector<Displacer> ds;
    for (const auto& s : specs) ds.emplace_back(s);
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float fu = 0, fv = 0, iu = 0, iv = 0;  // DisplacementComposer::flow_at / iflow_at
        for (const Displacer& d : ds) {
          float u, v;
          d.raw_flow(x, y, &u, &v);
          const float w{d.sup.at(x, y)};
          fu += u * w; fv += v * w;
        }
        for (const Displacer& d : ds) {
          float u, v;
          d.raw_iflow(x, y, &u, &v);
          const float w{d.sup.at(x, y)};
          iu += u * w; iv += v * w;
        }
        flow[(size_t)y * W + x] = fu; flow[n + (size_t)y * W + x] = fv;
        iflow[(size_t)y * W + x] = iu; iflow[n + (size_t)y * W + x] = iv;
      }
    compose(flow, W, H, iters);
    compose(iflow, W, H, iters);
  }
  void clamp_near_zeros() {  // WF:444-455
    const float threshold{1e-3};
    for (float& v : flow) if (std::abs(v) < threshold) v = 0.f;
    for (float& v : iflow) if (std::abs(v) < threshold) v = 0.f;
  }
};


// CropGenerator::worker_thread_loop (WF:540-641) for ONE big field, with the reference's
// std::random_device seed replaced by `seed` (the reference is not reproducible here,
// SURVEY F-8).  Returns the displacer list in placement order.
inline std::vector<DisplacerSpec> make_displacers(int W, int H, unsigned seed) {
  std::mt19937 mersenne(seed);
  std::uniform_int_distribution<> displacer_type(0, 2);
  std::uniform_real_distribution<> generic_param(-1, 1);
  const int big_size{std::max(W, H) * 3};
  std::vector<DisplacerSpec> out;
  const int spacing{200};
  const int isosceles_spacing{(int)(spacing / 2. * std::sqrt(3.))};
  const int rows{(big_size + isosceles_spacing - 1) / isosceles_spacing};
  const int cols{(big_size) / spacing};
  for (int yidx = 0; yidx < rows; ++yidx) {
    for (int xidx = 0; xidx < cols; ++xidx) {
      const int x = xidx * spacing + (yidx % 2 == 1 ? spacing / 2 : 0) + spacing / 2;
      const int y = yidx * isosceles_spacing + spacing / 2;
      DisplacerSpec d;
      d.type = displacer_type(mersenne);
      // argument evaluation order of the reference's constructor calls: GCC evaluates
      // function arguments right to left, so the LAST parameter draws first.
      if (d.type == 0) {
        const double b = generic_param(mersenne) * 3e-4;
        const double a = generic_param(mersenne) * 3e-4;
        d.p0 = a; d.p1 = b; d.p2 = 0;
      } else if (d.type == 1) {
        const double c = generic_param(mersenne) * M_PI * 2e-6;
        const double b = y + generic_param(mersenne) * 10;
        const double a = x + generic_param(mersenne) * 10;
        d.p0 = a; d.p1 = b; d.p2 = c;
      } else {
        const double c = 1 + generic_param(mersenne) * 2e-6;
        const double b = y + generic_param(mersenne) * 10;
        const double a = x + generic_param(mersenne) * 10;
        d.p0 = a; d.p1 = b; d.p2 = c;
      }
      const double s4 = generic_param(mersenne) * M_PI;
      const double s3 = 50 + generic_param(mersenne) * 20;
      const double s2 = 50 + generic_param(mersenne) * 20;
      const double s1 = y + generic_param(mersenne) * 10;
      const double s0 = x + generic_param(mersenne) * 10;
      d.sup_cx = s0; d.sup_cy = s1; d.sup_sx = s2; d.sup_sy = s3; d.sup_angle = s4;
      out.push_back(d);
    }
  }
  return out;
}

// crop origins of one big field (WF:617-633): y outer, x inner.
inline std::vector<std::pair<int, int>> crop_origins(int W, int H) {
  const int big_size{std::max(W, H) * 3};
  std::vector<std::pair<int, int>> o;
  for (int y = H / 4; y < big_size - 5 * H / 4; y += H / 3)
    for (int x = W / 4; x < big_size - 5 * W / 4; x += W / 3) o.push_back({x, y});
  return o;
}

}  // namespace oracle
