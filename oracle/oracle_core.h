// TEST INFRASTRUCTURE -- parity oracle, not product code.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// CPU restatement of the per-pixel engines the reference's hot path calls:
//  * Anti-Grain Geometry 2.4 (un-vendored dependency of the reference, pinned by
//    cmake/Dependencies.cmake:7-8, MD5 863d9992fd83c5d40fe1c011501ecf0e):
//    trans_affine, ellipse, curve3_div, rasterizer_cells_aa/rasterizer_scanline_aa,
//    pixfmt_gray8 solid blending, span_interpolator_linear + dda2_line_interpolator,
//    span_image_filter_rgb_bilinear, image_accessor_wrap<wrap_mode_reflect>.
//    Restated from the library's published algorithm; the rasteriser / curve
//    subdivision / DDA are pinned bit-exactly against matplotlib's compiled AGG
//    (tests/golden/gen_agg_goldens.py).
//  * CImg: draw_image (integer form), linear_atXY.
// Everything is sequential and written the way the libraries iterate (incremental
// stepping), so it is independent of the closed forms the HIP kernels use.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/ofdg_detmath.h"

namespace oracle {

// Which elementary functions build the affines (and the Gaussian supports of mode 9):
// 0 = the host libm, as the reference does (reference-stream path); 1 = ofdg_det_* of
// include/ofdg_detmath.h, the functions the DEVICE counter-sampler path is defined with (that path
// has no reference bit stream; see the header).  Set by ofdg_oracle_set_detmath().
inline int& detmath_flag() { static int f = 0; return f; }
inline int& lean_flag() { static int f = 0; return f; }  // CPU-baseline cost model, see ofdg_oracle_set_lean

// ---------------------------------------------------------------------------
// agg::trans_affine (AGG 2.4 agg_trans_affine.h / .cpp), used at
// DataGenerator.cpp:302-335, 203-205, 673.
// ---------------------------------------------------------------------------
struct Affine {
  double sx = 1, shy = 0, shx = 0, sy = 1, tx = 0, ty = 0;
  Affine() {}
  Affine(double a, double b, double c, double d, double e, double f)
      : sx(a), shy(b), shx(c), sy(d), tx(e), ty(f) {}
  static Affine rotation(double a) {
    if (detmath_flag()) {
      double s, c;
      ofdg_det_sincos(a, &s, &c);
      return Affine(c, s, -s, c, 0.0, 0.0);
    }
    return Affine(std::cos(a), std::sin(a), -std::sin(a), std::cos(a), 0.0, 0.0);
  }
  static Affine scaling(double s) { return Affine(s, 0.0, 0.0, s, 0.0, 0.0); }
  static Affine translation(double x, double y) { return Affine(1.0, 0.0, 0.0, 1.0, x, y); }
  Affine& multiply(const Affine& m) {
    double t0 = sx * m.sx + shy * m.shx;
    double t2 = shx * m.sx + sy * m.shx;
    double t4 = tx * m.sx + ty * m.shx + m.tx;
    shy = sx * m.shy + shy * m.sy;
    sy = shx * m.shy + sy * m.sy;
    ty = tx * m.shy + ty * m.sy + m.ty;
    sx = t0;
    shx = t2;
    tx = t4;
    return *this;
  }
  Affine& operator*=(const Affine& m) { return multiply(m); }
  Affine operator*(const Affine& m) const { return Affine(*this).multiply(m); }
  Affine& invert() {
    double d = 1.0 / (sx * sy - shy * shx);
    double t0 = sy * d;
    sy = sx * d;
    shy = -shy * d;
    shx = -shx * d;
    double t4 = -tx * t0 - ty * shx;
    ty = -tx * shy - ty * sy;
    sx = t0;
    tx = t4;
    return *this;
  }
  void transform(double* x, double* y) const {
    double tmp = *x;
    *x = tmp * sx + *y * shx + tx;
    *y = tmp * shy + *y * sy + ty;
  }
};

inline int iround(double v) { return int((v < 0.0) ? v - 0.5 : v + 0.5); }  // agg_basics.h

struct PointD { double x, y; };

// ---------------------------------------------------------------------------
// agg::curve3_div (agg_curves.cpp), approximation_scale = 1, angle_tolerance = 0.
// Reached through conv_curve at DataGenerator.cpp:525-527.
// ---------------------------------------------------------------------------
struct Curve3Div {
  std::vector<PointD> pts;
  double tol_sq;
  Curve3Div() {
    tol_sq = 0.5 / 1.0;
    tol_sq *= tol_sq;
  }
  void recursive_bezier(double x1, double y1, double x2, double y2, double x3, double y3, unsigned level) {
    if (level > 32) return;  // curve_recursion_limit
    double x12 = (x1 + x2) / 2;
    double y12 = (y1 + y2) / 2;
    double x23 = (x2 + x3) / 2;
    double y23 = (y2 + y3) / 2;
    double x123 = (x12 + x23) / 2;
    double y123 = (y12 + y23) / 2;
    double dx = x3 - x1;
    double dy = y3 - y1;
    double d = std::fabs(((x2 - x3) * dy - (y2 - y3) * dx));
    double da;
    if (d > 1e-30) {  // curve_collinearity_epsilon
      if (d * d <= tol_sq * (dx * dx + dy * dy)) {
        // angle_tolerance (0) < curve_angle_tolerance_epsilon (0.01)
        pts.push_back({x123, y123});
        return;
      }
    } else {
      da = dx * dx + dy * dy;
      if (da == 0) {
        d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);  // calc_sq_distance(x1,y1,x2,y2)
      } else {
        d = ((x2 - x1) * dx + (y2 - y1) * dy) / da;
        if (d > 0 && d < 1) return;
        if (d <= 0) d = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2);
        else if (d >= 1) d = (x3 - x2) * (x3 - x2) + (y3 - y2) * (y3 - y2);
        else {
          double px = x1 + d * dx, py = y1 + d * dy;
          d = (px - x2) * (px - x2) + (py - y2) * (py - y2);
        }
      }
      if (d < tol_sq) {
        pts.push_back({x2, y2});
        return;
      }
    }
    recursive_bezier(x1, y1, x12, y12, x123, y123, level + 1);
    recursive_bezier(x123, y123, x23, y23, x3, y3, level + 1);
  }
  void init(double x1, double y1, double x2, double y2, double x3, double y3) {
    pts.clear();
    pts.push_back({x1, y1});
    recursive_bezier(x1, y1, x2, y2, x3, y3, 0);
    pts.push_back({x3, y3});
  }
};

// ---------------------------------------------------------------------------
// agg::rasterizer_cells_aa<cell_aa> + rasterizer_scanline_aa<> (no clip box,
// non-zero fill) -- agg_rasterizer_cells_aa.h, agg_rasterizer_scanline_aa.h.
// Reached through MovingObjectBase::draw, DataGenerator.cpp:351-368.
// ---------------------------------------------------------------------------
struct Cell { int x, y, cover, area; };

struct Rasterizer {
  enum { shift = 8, scale = 256, mask = 255 };
  std::vector<Cell> cells;
  Cell cur;
  // rasterizer_scanline_aa state
  int start_x = 0, start_y = 0, x1 = 0, y1 = 0;
  int status = 0;  // 0 initial, 1 move_to, 2 line_to, 3 closed
  bool dx_limit_hit = false;

  Rasterizer() { reset(); }
  void reset() {
    cells.clear();
    cur.x = 0x7FFFFFFF; cur.y = 0x7FFFFFFF; cur.cover = 0; cur.area = 0;
    status = 0;
  }
  void add_curr_cell() { if (cur.area | cur.cover) cells.push_back(cur); }
  void set_curr_cell(int x, int y) {
    if (cur.x != x || cur.y != y) {
      add_curr_cell();
      cur.x = x; cur.y = y; cur.cover = 0; cur.area = 0;
    }
  }
  void render_hline(int ey, int x1, int y1, int x2, int y2) {
    int ex1 = x1 >> shift;
    int ex2 = x2 >> shift;
    int fx1 = x1 & mask;
    int fx2 = x2 & mask;
    int delta, p, first, dx;
    int incr, lift, mod, rem;
    if (y1 == y2) { set_curr_cell(ex2, ey); return; }
    if (ex1 == ex2) {
      delta = y2 - y1;
      cur.cover += delta;
      cur.area += (fx1 + fx2) * delta;
      return;
    }
    p = (scale - fx1) * (y2 - y1);
    first = scale;
    incr = 1;
    dx = x2 - x1;
    if (dx < 0) {
      p = fx1 * (y2 - y1);
      first = 0;
      incr = -1;
      dx = -dx;
    }
    delta = p / dx;
    mod = p % dx;
    if (mod < 0) { delta--; mod += dx; }
    cur.cover += delta;
    cur.area += (fx1 + first) * delta;
    ex1 += incr;
    set_curr_cell(ex1, ey);
    y1 += delta;
    if (ex1 != ex2) {
      p = scale * (y2 - y1 + delta);
      lift = p / dx;
      rem = p % dx;
      if (rem < 0) { lift--; rem += dx; }
      mod -= dx;
      while (ex1 != ex2) {
        delta = lift;
        mod += rem;
        if (mod >= 0) { mod -= dx; delta++; }
        cur.cover += delta;
        cur.area += scale * delta;
        y1 += delta;
        ex1 += incr;
        set_curr_cell(ex1, ey);
      }
    }
    delta = y2 - y1;
    cur.cover += delta;
    cur.area += (fx2 + scale - first) * delta;
  }
  void line(int x1, int y1, int x2, int y2) {
    const int dx_limit = 16384 << shift;
    int dx = x2 - x1;
    if (dx >= dx_limit || dx <= -dx_limit) {
      // AGG 2.4 splits here and then falls through; blueprints never get close
      // (|dx| would have to exceed 16384 px).  Flag it instead of restating a quirk.
      dx_limit_hit = true;
      return;
    }
    int dy = y2 - y1;
    int ex1 = x1 >> shift;
    int ey1 = y1 >> shift;
    int ey2 = y2 >> shift;
    int fy1 = y1 & mask;
    int fy2 = y2 & mask;
    int x_from, x_to;
    int p, rem, mod, lift, delta, first, incr;
    set_curr_cell(ex1, ey1);
    if (ey1 == ey2) { render_hline(ey1, x1, fy1, x2, fy2); return; }
    incr = 1;
    if (dx == 0) {
      int ex = x1 >> shift;
      int two_fx = (x1 - (ex << shift)) << 1;
      int area;
      first = scale;
      if (dy < 0) { first = 0; incr = -1; }
      x_from = x1;
      delta = first - fy1;
      cur.cover += delta;
      cur.area += two_fx * delta;
      ey1 += incr;
      set_curr_cell(ex, ey1);
      delta = first + first - scale;
      area = two_fx * delta;
      while (ey1 != ey2) {
        cur.cover = delta;
        cur.area = area;
        ey1 += incr;
        set_curr_cell(ex, ey1);
      }
      delta = fy2 - scale + first;
      cur.cover += delta;
      cur.area += two_fx * delta;
      return;
    }
    p = (scale - fy1) * dx;
    first = scale;
    if (dy < 0) { p = fy1 * dx; first = 0; incr = -1; dy = -dy; }
    delta = p / dy;
    mod = p % dy;
    if (mod < 0) { delta--; mod += dy; }
    x_from = x1 + delta;
    render_hline(ey1, x1, fy1, x_from, first);
    ey1 += incr;
    set_curr_cell(x_from >> shift, ey1);
    if (ey1 != ey2) {
      p = scale * dx;
      lift = p / dy;
      rem = p % dy;
      if (rem < 0) { lift--; rem += dy; }
      mod -= dy;
      while (ey1 != ey2) {
        delta = lift;
        mod += rem;
        if (mod >= 0) { mod -= dy; delta++; }
        x_to = x_from + delta;
        render_hline(ey1, x_from, scale - first, x_to, first);
        x_from = x_to;
        ey1 += incr;
        set_curr_cell(x_from >> shift, ey1);
      }
    }
    render_hline(ey1, x_from, scale - first, x2, fy2);
  }
  // rasterizer_scanline_aa<>::move_to_d / line_to_d / close_polygon with
  // rasterizer_sl_clip_int, clipping disabled (DataGenerator.cpp:355-356).
  void close_polygon() {
    if (status == 2) {
      line(x1, y1, start_x, start_y);
      x1 = start_x; y1 = start_y;
      status = 3;
    }
  }
  void move_to_d(double x, double y) {
    close_polygon();  // m_auto_close
    start_x = x1 = iround(x * scale);
    start_y = y1 = iround(y * scale);
    status = 1;
  }
  void line_to_d(double x, double y) {
    int x2 = iround(x * scale), y2 = iround(y * scale);
    line(x1, y1, x2, y2);
    x1 = x2; y1 = y2;
    status = 2;
  }
  // Sweep all scanlines and write the raw coverage (before any pixel-format
  // blending) of the pixels inside [0,w)x[0,h); gamma = identity or threshold.
  // out must be zero-initialised by the caller (renderer_base::clear).
  void sweep(int w, int h, uint8_t* out, bool threshold_gamma) {
    close_polygon();      // rewind_scanlines(): m_auto_close
    add_curr_cell();      // sort_cells()
    cur.x = 0x7FFFFFFF; cur.y = 0x7FFFFFFF; cur.cover = 0; cur.area = 0;
    std::vector<Cell> sorted(cells);
    std::stable_sort(sorted.begin(), sorted.end(), [](const Cell& a, const Cell& b) {
      return a.y != b.y ? a.y < b.y : a.x < b.x;
    });
    size_t i = 0;
    while (i < sorted.size()) {
      int y = sorted[i].y;
      size_t j = i;
      while (j < sorted.size() && sorted[j].y == y) ++j;
      // sweep_scanline
      int cover = 0;
      size_t k = i;
      while (k < j) {
        int x = sorted[k].x;
        int area = sorted[k].area;
        cover += sorted[k].cover;
        ++k;
        while (k < j && sorted[k].x == x) {
          area += sorted[k].area;
          cover += sorted[k].cover;
          ++k;
        }
        if (area) {
          unsigned alpha = calculate_alpha((cover << (shift + 1)) - area, threshold_gamma);
          if (alpha) put(out, w, h, x, y, alpha);
          x++;
        }
        if (k < j && sorted[k].x > x) {
          unsigned alpha = calculate_alpha(cover << (shift + 1), threshold_gamma);
          if (alpha) for (int xx = x; xx < sorted[k].x; ++xx) put(out, w, h, xx, y, alpha);
        }
      }
      i = j;
    }
  }
  static unsigned calculate_alpha(int area, bool threshold_gamma) {
    int cover = area >> (shift * 2 + 1 - 8);
    if (cover < 0) cover = -cover;
    if (cover > 255) cover = 255;
    // gamma_none: identity LUT; gamma_threshold(0.5): uround((i/255.0 < 0.5 ? 0 : 1) * 255)
    if (threshold_gamma) return (cover / 255.0 < 0.5) ? 0u : 255u;
    return (unsigned)cover;
  }
  static void put(uint8_t* out, int w, int h, int x, int y, unsigned alpha) {
    if (x < 0 || y < 0 || x >= w || y >= h) return;  // renderer_base clipping
    out[(size_t)y * w + x] = (uint8_t)alpha;
  }
};

// renderer_scanline_aa_solid on pixfmt_gray8 with colour gray8(255) onto a
// cleared buffer (AGG 2.4 agg_pixfmt_gray.h blend_solid_hspan + blender_gray):
//   alpha = (255 * (cover + 1)) >> 8;  alpha == 255 ? 255 : ((255 - 0) * alpha + (0 << 8)) >> 8
inline uint8_t gray8_solid_on_clear(uint8_t cover) {
  if (cover == 0) return 0;  // no span emitted
  unsigned alpha = (255u * (unsigned(cover) + 1)) >> 8;
  if (alpha == 255) return 255;
  unsigned p = 0;
  return (uint8_t)((((255u - p) * alpha) + (p << 8)) >> 8);
}

// ---------------------------------------------------------------------------
// Outline of a shape (vertex source) under an affine, the way conv_transform
// (+ conv_curve for polygons) feed the rasteriser: DataGenerator.cpp:465-479,
// 520-534, 1080, 1091-1114.
// ---------------------------------------------------------------------------
struct ShapeGeom {
  int type = 0;  // OFDG_OBJ_ELLIPSE / OFDG_OBJ_POLYGON
  double rx = 0, ry = 0;
  int n_seg = 0;
  int seg_type[32];
  double seg_x[32], seg_y[32];
};

// Returns the flattened outline (screen space, doubles) -- one closed polygon.
inline std::vector<PointD> outline(const ShapeGeom& g, const Affine& m) {
  std::vector<PointD> out;
  if (g.type == 1) {
    // agg::ellipse::init(0,0,rx,ry,100) ; vertex(): angle = step/num * 2*pi
    const double pi = 3.14159265358979323846;
    for (int step = 0; step < 100; ++step) {
      double angle = double(step) / double(100) * 2.0 * pi;
      double x = 0.0 + std::cos(angle) * g.rx;
      double y = 0.0 + std::sin(angle) * g.ry;
      m.transform(&x, &y);
      out.push_back({x, y});
    }
    return out;
  }
  // polygon: move_to(seg0); Line -> line_to; Curve3 -> curve3(ctrl=seg[i], to=seg[i+1]), ++i
  double lx = g.seg_x[0], ly = g.seg_y[0];
  m.transform(&lx, &ly);
  out.push_back({lx, ly});
  Curve3Div c3;
  for (int i = 1; i < g.n_seg; ++i) {
    if (g.seg_type[i] == 1) {
      double x = g.seg_x[i], y = g.seg_y[i];
      m.transform(&x, &y);
      out.push_back({x, y});
      lx = x; ly = y;
    } else if (g.seg_type[i] == 3) {
      double cx = g.seg_x[i], cy = g.seg_y[i];
      double ex = g.seg_x[i + 1], ey = g.seg_y[i + 1];
      m.transform(&cx, &cy);
      m.transform(&ex, &ey);
      c3.init(lx, ly, cx, cy, ex, ey);
      // conv_curve: first point is the (already emitted) start; the rest are line_to
      for (size_t k = 1; k < c3.pts.size(); ++k) out.push_back(c3.pts[k]);
      lx = ex; ly = ey;
      ++i;
    }
    // Dummy inside the list cannot occur (DataGenerator.cpp:1095-1098)
  }
  return out;
}

inline bool rasterize_polygon(const std::vector<PointD>& poly, int w, int h, uint8_t* cov, bool threshold_gamma) {
  Rasterizer r;
  std::memset(cov, 0, (size_t)w * h);
  if (poly.empty()) return true;
  r.move_to_d(poly[0].x, poly[0].y);
  for (size_t i = 1; i < poly.size(); ++i) r.line_to_d(poly[i].x, poly[i].y);
  r.sweep(w, h, cov, threshold_gamma);
  return !r.dx_limit_hit;
}

// ---------------------------------------------------------------------------
// dda2_line_interpolator (agg_dda_line.h) + span_interpolator_linear::begin
// ---------------------------------------------------------------------------
struct Dda2 {
  int cnt, lft, rem, mod, y;
  Dda2(int y1, int y2, int count)
      : cnt(count <= 0 ? 1 : count), lft((y2 - y1) / cnt), rem((y2 - y1) % cnt), mod(rem), y(y1) {
    if (mod <= 0) { mod += count; rem += count; lft--; }
    mod -= count;
  }
  void operator++() {
    mod += rem;
    y += lft;
    if (mod > 0) { mod -= cnt; y++; }
  }
};

// wrap_mode_reflect (agg_image_accessors.h)
struct WrapReflect {
  unsigned size, size2, add, value;
  explicit WrapReflect(unsigned s) : size(s), size2(s * 2), add(size2 * (0x3FFFFFFF / size2)), value(0) {}
  unsigned operator()(int v) {
    value = (unsigned(v) + add) % size2;
    if (value >= size) return size2 - value - 1;
    return value;
  }
  unsigned inc() {
    ++value;
    if (value >= size2) value = 0;
    if (value >= size) return size2 - value - 1;
    return value;
  }
};

// getTransformedTexture (DataGenerator.cpp:168-231): in/out planar u8 [3][th][tw].
// kRegion (the "lean" CPU baseline only): the same values, but only for the pixels of rows ry0..ry1, columns
// rx0..rx1 (the reference always renders the full rectangle).
template <bool kRegion>
inline void transformed_texture_impl(const uint8_t* in, int tw, int th, const Affine& tf, uint8_t* out, int rx0, int ry0, int rx1, int ry1) {
  Affine inv = tf;
  inv.invert();
  const size_t plane = (size_t)tw * th;
  for (int y = kRegion ? ry0 : 0; y < (kRegion ? ry1 + 1 : th); ++y) {
    // one span per row: x = 0, len = tw (the rendered path is the image rectangle)
    double tx = 0 + 0.5, ty = y + 0.5;
    inv.transform(&tx, &ty);
    int x1 = iround(tx * 256), y1 = iround(ty * 256);
    tx = 0 + 0.5 + tw; ty = y + 0.5;
    inv.transform(&tx, &ty);
    int x2 = iround(tx * 256), y2 = iround(ty * 256);
    Dda2 lix(x1, x2, tw), liy(y1, y2, tw);
    WrapReflect wx(tw), wy(th);
    for (int x = 0; x < (kRegion ? rx1 + 1 : tw); ++x) {
      if (kRegion && x < rx0) { ++lix; ++liy; continue; }
      int x_hr = lix.y - 128, y_hr = liy.y - 128;
      int x_lr = x_hr >> 8, y_lr = y_hr >> 8;
      unsigned fg[3] = {256 * 256 / 2, 256 * 256 / 2, 256 * 256 / 2};
      x_hr &= 255; y_hr &= 255;
      unsigned row = wy(y_lr);
      unsigned col = wx(x_lr);
      unsigned weight = (256 - x_hr) * (256 - y_hr);
      for (int c = 0; c < 3; ++c) fg[c] += weight * in[c * plane + (size_t)row * tw + col];
      unsigned col1 = wx.inc();
      weight = x_hr * (256 - y_hr);
      for (int c = 0; c < 3; ++c) fg[c] += weight * in[c * plane + (size_t)row * tw + col1];
      unsigned row1 = wy.inc();
      col = wx(x_lr);
      weight = (256 - x_hr) * y_hr;
      for (int c = 0; c < 3; ++c) fg[c] += weight * in[c * plane + (size_t)row1 * tw + col];
      col1 = wx.inc();
      weight = x_hr * y_hr;
      for (int c = 0; c < 3; ++c) fg[c] += weight * in[c * plane + (size_t)row1 * tw + col1];
      for (int c = 0; c < 3; ++c) out[c * plane + (size_t)y * tw + x] = (uint8_t)(fg[c] >> 16);
      ++lix; ++liy;
    }
  }
}
inline void transformed_texture(const uint8_t* in, int tw, int th, const Affine& tf, uint8_t* out) {
  transformed_texture_impl<false>(in, tw, th, tf, out, 0, 0, tw - 1, th - 1);
}
inline void transformed_texture_region(const uint8_t* in, int tw, int th, const Affine& tf, uint8_t* out, int rx0, int ry0, int rx1, int ry1) {
  transformed_texture_impl<true>(in, tw, th, tf, out, rx0, ry0, rx1, ry1);
}

// CImg<unsigned char>::draw_image(0,0,sprite,mask,1,255) per value
// (DataGenerator.cpp:782,792):  d = (T)((|m|*s + d*(255 - max(m,0))) / 255.f).
// All operands are exact in fp32; evaluated in fp32 like CImg does.
inline uint8_t draw_image_value(uint8_t d, uint8_t s, uint8_t m) {
  const float mopacity = (float)m * 1.f;
  const float nopacity = std::fabs(mopacity), copacity = 255.f - std::max(mopacity, 0.f);
  return (uint8_t)((nopacity * s + d * copacity) / 255.f);
}

// MovingObjectComposite::renderMasks per byte (DataGenerator.cpp:606, 626), fp32.
inline uint8_t composite_add(uint8_t u, uint8_t v) {
  return static_cast<unsigned char>(255.f * (1.f - (1.f - u / 255.f) * (1.f - v / 255.f)));
}
inline uint8_t composite_sub(uint8_t u, uint8_t v) {
  return static_cast<unsigned char>(255.f * ((u / 255.f) * (1.f - v / 255.f)));
}

}  // namespace oracle
