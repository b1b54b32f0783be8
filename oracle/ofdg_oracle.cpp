// TEST INFRASTRUCTURE -- parity oracle, not product code.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// import, call, link or execute anything under oracle/.  The product
// (optical-flow-2d-data-generation_amd/) never links this file.
//
// CPU restatement of the reference's hot path, one sample at a time, in the
// reference's own structure (per-object full-frame masks and textures, painter's
// blit, per-pixel flow):  DataGenerator::Process_TaskBucket and everything it
// calls (src/caffe/DataGenerator.cpp:1175-1254).  Citations "DG:" are into
// src/caffe/DataGenerator.cpp of the reference.
//
// Parity status: the sampler's RNG layer is pinned against the reference's own
// SimpleRandom.h compiled here (oracle/_ref, tests/golden/rng_goldens.json) and
// against the blueprint values recorded in SURVEY.md Appendix E.1; rasteriser,
// curve flattening and DDA are pinned against matplotlib's compiled AGG
// (tests/golden/agg_goldens.npz).  The reference as a whole cannot be built here
// (Caffe, AGG 2.4 and CImg are absent), so the remaining pieces
// (gray8 blend byte mapping, bilinear weights, reflect wrap, CImg draw_image)
// follow the libraries' published source: "parity unpinned" for those.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <thread>

#include "../include/ofdg.h"
#include "oracle_core.h"
#include "oracle_sampler.h"
#include "oracle_warpfields.h"

using namespace oracle;

namespace {

struct Pool {
  int n = 0, w = 0, h = 0;
  const uint8_t* data = nullptr;  // n * 3 * h * w, planar B,G,R per texture
  const uint8_t* tex(int raw_index) const {  // TextureCollection::getTexturePtr, DG:158-161
    return data + (size_t)(raw_index % n) * 3 * w * h;
  }
};

// CImg<unsigned char>::get_resize(sx, sy, -100, -100, 3) of one plane (CImg 2.x, recalled; PARITY
// UNPINNED): X pass then Y pass, the intermediate image is u8 again.  Per axis: same size ->
// copy; enlarging -> linear with step (w - 1) / (s - 1) (boundary 0), value (T)((1-a)*v1 + a*v2);
// shrinking -> moving average (interpolation 2): every destination pixel integrates the source
// pixels it overlaps on the w*s grid, in float, divided by w, truncated.
std::vector<uint8_t> cimg_resize_axis_u8(const std::vector<uint8_t>& in, int w, int h, int s, bool along_x) {
  const int n = along_x ? w : h;            // source length along the axis
  const int ow = along_x ? s : w, oh = along_x ? h : s;
  std::vector<uint8_t> out((size_t)ow * oh);
  if (s == n) return in;
  auto src = [&](int line, int k) -> uint8_t { return along_x ? in[(size_t)line * w + k] : in[(size_t)k * w + line]; };
  auto dst = [&](int line, int k) -> uint8_t& { return along_x ? out[(size_t)line * ow + k] : out[(size_t)k * ow + line]; };
  const int lines = along_x ? h : w;
  if (s > n) {
    const double f = s > 1 ? (n - 1.) / (s - 1) : 0;
    std::vector<unsigned int> off(s);
    std::vector<double> foff(s);
    double curr = 0, old = 0;
    for (int x = 0; x < s; ++x) {
      foff[x] = curr - (unsigned int)curr;
      old = curr;
      curr = std::min(n - 1., curr + f);
      off[x] = (unsigned int)curr - (unsigned int)old;
    }
    for (int line = 0; line < lines; ++line) {
      int at = 0;
      for (int x = 0; x < s; ++x) {
        const double alpha = foff[x];
        const uint8_t v1 = src(line, at), v2 = at < n - 1 ? src(line, at + 1) : v1;
        dst(line, x) = (uint8_t)((1 - alpha) * v1 + alpha * v2);
        at += (int)off[x];
      }
    }
  } else {
    std::vector<float> tmp(s);
    for (int line = 0; line < lines; ++line) {
      std::fill(tmp.begin(), tmp.end(), 0.f);
      for (unsigned int a = (unsigned)n * s, b = n, c = s, si = 0, t = 0; a;) {
        const unsigned int d = std::min(b, c);
        a -= d; b -= d; c -= d;
        tmp[t] += (float)src(line, si) * d;
        if (!b) { tmp[t] /= n; ++t; b = n; }
        if (!c) { ++si; c = s; }
      }
      for (int x = 0; x < s; ++x) dst(line, x) = (uint8_t)tmp[x];
    }
  }
  return out;
}
std::vector<uint8_t> cimg_resize_u8(const uint8_t* plane, int w, int h, int sx, int sy) {
  std::vector<uint8_t> in(plane, plane + (size_t)w * h);
  std::vector<uint8_t> rx = cimg_resize_axis_u8(in, w, h, sx, true);
  return cimg_resize_axis_u8(rx, sx, h, sy, false);
}

// Texture::getRandomizedCrop with default arguments (DG:87-109 called at
// DG:1149-1150): get_shift(0,0), rotate(0), then - image at least cw x ch -
// crop(w/2-cw/2, h/2-ch/2, +cw-1, +ch-1), resize(cw,ch): all identities except the centre
// crop; - smaller image - resize(cw, ch) of the whole image.
std::vector<uint8_t> centre_crop(const Pool& pool, int raw_index, int cw, int ch) {
  std::vector<uint8_t> out((size_t)3 * cw * ch);
  const uint8_t* t = pool.tex(raw_index);
  if (pool.w >= cw && pool.h >= ch) {
    const int x0 = pool.w / 2 - cw / 2, y0 = pool.h / 2 - ch / 2;
    for (int c = 0; c < 3; ++c)
      for (int y = 0; y < ch; ++y)
        std::memcpy(&out[((size_t)c * ch + y) * cw], t + ((size_t)c * pool.h + (y0 + y)) * pool.w + x0, cw);
  } else {
    for (int c = 0; c < 3; ++c) {
      const std::vector<uint8_t> r = cimg_resize_u8(t + (size_t)c * pool.w * pool.h, pool.w, pool.h, cw, ch);
      std::memcpy(&out[(size_t)c * cw * ch], r.data(), r.size());
    }
  }
  return out;
}

// Texture::getRandomizedCrop(2W, 2H, angle, zoom, x_shift, y_shift) for the background
// (DG:87-109, 1186-1192), branch "image at least 2W x 2H":
//   get_shift(sx, sy, 0, 0, 3).rotate(angle, 1, 3).crop(w/2-W, h/2-H, w/2-W+2W/zoom-1, h/2-H+2H/zoom-1, 3)
//   .resize(2W, 2H, -100, -100, 3)
// CImg is not in the image and its boundary code 3 / rotate / resize details are version
// dependent: PARITY UNPINNED.  Restated here as ONE resampling along the
// composed coordinate chain of CImg 2.x (the reference interpolates twice - rotate, then
// resize - and box-filters when zoom < 1; same geometry, one interpolation less of blur):
//   resize   : source column of destination u = min(cw - 1, u * f), f = (cw - 1) / (2W - 1) when
//              enlarging, cw / 2W otherwise                                       (CImg get_resize case 3, boundary 0)
//   crop     : + (x0, y0) = (w/2 - W, h/2 - H) inside the ROTATED image
//   rotate   : about the centres, the rotated image grown to round(1 + |(w-1)ca| + |(h-1)sa|) x ...;
//              angle is taken in DEGREES although the sampler draws radians (reference quirk, DG:1656)
//              x = w2 + xc*ca + yc*sa, y = h2 - xc*sa + yc*ca, mirrored into [0, w) x [0, h) (boundary 3)
//   sample   : _linear_atXY (Neumann) on the SHIFTED image; a shifted texel (i, j) is pool texel
//              (mirror(i - sx), mirror(j - sy)) (get_shift with boundary 3 = mirrored crop)
//   store    : truncation to u8
struct BgPrep {
  float ca, sa, w2, h2, rw2, rh2, fx, fy;
  int x0, y0, cw, ch, shx, shy;
  int rw, rh;  // size of the rotated image
};
inline float cimg_modf(float x, float m) { return (float)(x - m * std::floor((double)x / m)); }
inline int cimg_modi(int x, int m) { const int r = x % m; return r < 0 ? r + m : r; }
BgPrep make_bg_prep(int pw, int ph, int W, int H, float angle, float zoom, int shx, int shy) {
  BgPrep p;
  const int TW = 2 * W, TH = 2 * H;
  const float nangle = cimg_modf(angle, 360.0f);
  const float rad = (float)(nangle * 3.14159265358979323846 / 180.0);
  if (detmath_flag()) {  // the device counter-sampler path: include/ofdg_detmath.h's fp64 sin / cos, rounded to float
    double sa, ca;
    ofdg_det_sincos((double)rad, &sa, &ca);
    p.ca = (float)ca; p.sa = (float)sa;
  } else {
    p.ca = std::cos(rad); p.sa = std::sin(rad);  // (float overloads, as in CImg)
  }
  const float ux = std::fabs((pw - 1) * p.ca), uy = std::fabs((pw - 1) * p.sa);
  const float vx = std::fabs((ph - 1) * p.sa), vy = std::fabs((ph - 1) * p.ca);
  const int rw = (int)std::floor(1 + ux + vx + 0.5f), rh = (int)std::floor(1 + uy + vy + 0.5f);
  p.w2 = 0.5f * (pw - 1); p.h2 = 0.5f * (ph - 1);
  p.rw2 = 0.5f * (rw - 1); p.rh2 = 0.5f * (rh - 1);
  p.rw = rw; p.rh = rh;
  if (pw >= TW && ph >= TH) {
    p.x0 = pw / 2 - TW / 2; p.y0 = ph / 2 - TH / 2;
    const int x1 = (int)((float)p.x0 + (float)TW / zoom - 1.0f), y1 = (int)((float)p.y0 + (float)TH / zoom - 1.0f);
    p.cw = x1 - p.x0 + 1; p.ch = y1 - p.y0 + 1;
  } else {  // smaller image: no crop, the whole rotated image is resized (DG:102-106)
    p.x0 = 0; p.y0 = 0; p.cw = rw; p.ch = rh;
  }
  // get_resize(.., 3) with boundary 0: step (w - 1) / (sx - 1) when enlarging, w / sx otherwise
  p.fx = TW > p.cw ? (float)((p.cw - 1.0) / (TW - 1.0)) : (float)((double)p.cw / TW);
  p.fy = TH > p.ch ? (float)((p.ch - 1.0) / (TH - 1.0)) : (float)((double)p.ch / TH);
  p.shx = shx; p.shy = shy;
  return p;
}
std::vector<uint8_t> prepared_background(const Pool& pool, int raw_index, int W, int H, float angle, float zoom, int shx, int shy) {
  const int TW = 2 * W, TH = 2 * H, pw = pool.w, ph = pool.h;
  const BgPrep p = make_bg_prep(pw, ph, W, H, angle, zoom, shx, shy);
  const uint8_t* t = pool.tex(raw_index);
  std::vector<uint8_t> out((size_t)3 * TW * TH);
  const float ww = 2.0f * pw, hh = 2.0f * ph;
  for (int v = 0; v < TH; ++v)
    for (int u = 0; u < TW; ++u) {
      const float cxf = std::min((float)(p.cw - 1), (float)u * p.fx), cyf = std::min((float)(p.ch - 1), (float)v * p.fy);
      const float xc = ((float)p.x0 + cxf) - p.rw2, yc = ((float)p.y0 + cyf) - p.rh2;
      float mx = cimg_modf((p.w2 + xc * p.ca) + yc * p.sa, ww), my = cimg_modf((p.h2 - xc * p.sa) + yc * p.ca, hh);
      mx = mx < (float)pw ? mx : (ww - mx) - 1.0f;
      my = my < (float)ph ? my : (hh - my) - 1.0f;
      // _linear_atXY (Neumann)
      const float nfx = mx <= 0 ? 0.f : (mx >= (float)(pw - 1) ? (float)(pw - 1) : mx);
      const float nfy = my <= 0 ? 0.f : (my >= (float)(ph - 1) ? (float)(ph - 1) : my);
      const int x = (int)nfx, y = (int)nfy;
      const float dx = nfx - x, dy = nfy - y;
      const int nx = dx > 0 ? x + 1 : x, ny = dy > 0 ? y + 1 : y;
      // the four texels of the shifted image
      auto sxm = [&](int i) { const int m = cimg_modi(i - p.shx, 2 * pw); return m < pw ? m : 2 * pw - m - 1; };
      auto sym = [&](int j) { const int m = cimg_modi(j - p.shy, 2 * ph); return m < ph ? m : 2 * ph - m - 1; };
      const int xa = sxm(x), xb = sxm(nx), ya = sym(y), yb = sym(ny);
      for (int c = 0; c < 3; ++c) {
        const uint8_t* pl = t + (size_t)c * pw * ph;
        const float Icc = pl[(size_t)ya * pw + xa], Inc = pl[(size_t)ya * pw + xb];
        const float Icn = pl[(size_t)yb * pw + xa], Inn = pl[(size_t)yb * pw + xb];
        const float val = Icc + dx * (Inc - Icc + dy * (Icc + Inn - Icn - Inc)) + dy * (Icn - Icc);
        out[((size_t)c * TH + v) * TW + u] = (uint8_t)val;
      }
    }
  return out;
}

// The same chain the way CImg 2.x really runs it (ofdg_params.background_prep = 1; recalled source, SURVEY
// App. C - PARITY UNPINNED): four images, each rounded to unsigned char before the next stage.
//   S = get_shift(sx, sy, 0, 0, 3)   = get_crop(-sx, -sy, w-sx-1, h-sy-1, 3): S(i,j) = T(mirror(i-sx), mirror(j-sy))
//   R = S.rotate(angle, 1, 3)        rw x rh = round(1 + |(w-1)ca| + |(h-1)sa|) x round(1 + |(w-1)sa| + |(h-1)ca|);
//                                    R(x,y) = (uchar)S._linear_atXY(mirror_f(w2 + xc*ca + yc*sa), mirror_f(h2 - xc*sa + yc*ca)),
//                                    xc = x - rw2, yc = y - rh2  (angle == 0 mod 360: R = S, which the formula also yields)
//   C = R.crop(x0, y0, x1, y1, 3)    cw x ch, C(i,j) = R(mirror(x0+i), mirror(y0+j)) (the crop leaves R when zoom < 1)
//   B = C.resize(2W, 2H, -100, -100, 3): per axis - enlarging: linear (running double sums), shrinking: moving
//                                    average, same size: copy; unsigned char between the X and the Y pass.
std::vector<uint8_t> prepared_background_two_pass(const Pool& pool, int raw_index, int W, int H, float angle, float zoom, int shx, int shy) {
  const int TW = 2 * W, TH = 2 * H, pw = pool.w, ph = pool.h;
  const BgPrep p = make_bg_prep(pw, ph, W, H, angle, zoom, shx, shy);
  const int rw = p.rw, rh = p.rh;
  const uint8_t* t = pool.tex(raw_index);
  const float ww = 2.0f * pw, hh = 2.0f * ph;
  auto sxm = [&](int i) { const int m = cimg_modi(i - p.shx, 2 * pw); return m < pw ? m : 2 * pw - m - 1; };
  auto sym = [&](int j) { const int m = cimg_modi(j - p.shy, 2 * ph); return m < ph ? m : 2 * ph - m - 1; };
  std::vector<uint8_t> out((size_t)3 * TW * TH);
  std::vector<uint8_t> C((size_t)p.cw * p.ch);
  for (int c = 0; c < 3; ++c) {
    const uint8_t* pl = t + (size_t)c * pw * ph;
    for (int j = 0; j < p.ch; ++j)
      for (int i = 0; i < p.cw; ++i) {
        int rx = cimg_modi(p.x0 + i, 2 * rw), ry = cimg_modi(p.y0 + j, 2 * rh);
        rx = rx < rw ? rx : 2 * rw - rx - 1;
        ry = ry < rh ? ry : 2 * rh - ry - 1;
        const float xc = (float)rx - p.rw2, yc = (float)ry - p.rh2;
        float mx = cimg_modf((p.w2 + xc * p.ca) + yc * p.sa, ww), my = cimg_modf((p.h2 - xc * p.sa) + yc * p.ca, hh);
        mx = mx < (float)pw ? mx : (ww - mx) - 1.0f;
        my = my < (float)ph ? my : (hh - my) - 1.0f;
        const float nfx = mx <= 0 ? 0.f : (mx >= (float)(pw - 1) ? (float)(pw - 1) : mx);
        const float nfy = my <= 0 ? 0.f : (my >= (float)(ph - 1) ? (float)(ph - 1) : my);
        const int x = (int)nfx, y = (int)nfy;
        const float dx = nfx - x, dy = nfy - y;
        const int nx = dx > 0 ? x + 1 : x, ny = dy > 0 ? y + 1 : y;
        const int xa = sxm(x), xb = sxm(nx), ya = sym(y), yb = sym(ny);
        const float Icc = pl[(size_t)ya * pw + xa], Inc = pl[(size_t)ya * pw + xb];
        const float Icn = pl[(size_t)yb * pw + xa], Inn = pl[(size_t)yb * pw + xb];
        C[(size_t)j * p.cw + i] = (uint8_t)(Icc + dx * (Inc - Icc + dy * (Icc + Inn - Icn - Inc)) + dy * (Icn - Icc));
      }
    const std::vector<uint8_t> B = cimg_resize_u8(C.data(), p.cw, p.ch, TW, TH);
    std::memcpy(&out[(size_t)c * TW * TH], B.data(), B.size());
  }
  return out;
}

ShapeGeom geom_of(const ofdg_blueprint& p) {  // DG:1073-1117
  ShapeGeom g;
  g.type = p.obj_type;
  g.rx = p.ellipse_scale_x;
  g.ry = p.ellipse_scale_y;
  g.n_seg = p.n_segments;
  for (int i = 0; i < p.n_segments; ++i) {
    g.seg_type[i] = p.segment_type[i];
    g.seg_x[i] = p.segment_x[i];
    g.seg_y[i] = p.segment_y[i];
  }
  return g;
}

// One MovingObject* (DG:256-718) with its four masks and two warped textures.
struct Object {
  int id = 0;
  bool is_background = false;
  Affine intrinsic, intrinsic_inv, motion, motion_inv;
  std::vector<uint8_t> mask_noAA[2], mask_AA[2];
  std::vector<uint8_t> tex[3];  // m_textures[0..2]
  bool has_warp = false;
  const WarpCrop* warp = nullptr;
  std::unique_ptr<WarpCrop> own_warp;  // background's upscaled copy
  // lean cost model only: pixels outside this box are zero in both masks of frame f (inclusive; empty: x0 > x1)
  int bb[2][4] = {{0, 0, -1, -1}, {0, 0, -1, -1}};
};

struct Ctx {
  int W, H, mode;
  bool use_AA;
  // true: the reference's work pattern - 4 rasterisations and 2 full-frame texture warps per shape, full-frame
  // composites and blits for every object, visible or not (DG:337-349, 465-479, 762-799).  false ("lean", CPU
  // baseline only, same output bit for bit): one rasterisation per frame, work restricted to the outline's box,
  // identity warps are copies, objects entirely off-screen are skipped.
  bool faithful;
  // Texture::getRandomizedCrop with the sampled rotation / zoom / shift (DG:1186-1192): 0 off (centre crop),
  // 1 the CImg chain stage by stage (four u8 images), 2 one resampling along the composed coordinate map
  int background_prep = 0;
};

void set_intrinsic(Object& o, float alpha, float xs, float ys) {  // DG:302-310
  o.intrinsic = Affine();
  o.intrinsic *= Affine::rotation(alpha);
  o.intrinsic *= Affine::translation(xs, ys);
  o.intrinsic_inv = o.intrinsic;
  o.intrinsic_inv.invert();
}
void set_motion(Object& o, float alpha, float scale, float xs, float ys) {  // DG:312-322
  o.motion = Affine();
  o.motion *= Affine::rotation(alpha);
  o.motion *= Affine::scaling(scale);
  o.motion *= Affine::translation(xs, ys);
  o.motion_inv = o.motion;
  o.motion_inv.invert();
}
void add_background_motion(Object& o, const Affine& bg_motion, int W, int H) {  // DG:324-335
  Affine bg_n = Affine::translation(-W / 2., -H / 2.);
  bg_n *= bg_motion;
  bg_n *= Affine::translation(W / 2., H / 2.);
  o.motion *= bg_n;
  o.motion_inv = o.motion;
  o.motion_inv.invert();
}

// MovingObjectBase::draw (DG:351-368): rasterise, gray8-blend onto cleared scratch.
void draw(const Ctx& c, const std::vector<PointD>& poly, bool AA, std::vector<uint8_t>& mask, bool* ok) {
  std::vector<uint8_t> cov((size_t)c.W * c.H);
  if (!rasterize_polygon(poly, c.W, c.H, cov.data(), !AA)) *ok = false;
  mask.resize(cov.size());
  for (size_t i = 0; i < cov.size(); ++i) mask[i] = gray8_solid_on_clear(cov[i]);
}

// applyWarpFieldToTexture (DG:237-252) for `channels` planes of W x H.
void apply_warp(const std::vector<uint8_t>& in, int W, int H, int channels, const WarpCrop& wc, bool inverse,
                std::vector<uint8_t>& out) {
  out.resize(in.size());
  const std::vector<float>& f = inverse ? wc.iflow : wc.flow;
  for (int ch = 0; ch < channels; ++ch)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float fx = x + f[(size_t)y * wc.w + x];
        float fy = y + f[(size_t)wc.w * wc.h + (size_t)y * wc.w + x];
        out[((size_t)ch * H + y) * W + x] = cimg_linear_atXY_dirichlet_u8(&in[(size_t)ch * W * H], W, H, fx, fy);
      }
}

// lean: ONE rasterisation gives both masks of a frame (gamma_threshold(0.5) of the same raw coverage, AGG B.4),
// computed inside the outline's pixel box only
void draw_lean(const Ctx& c, const std::vector<PointD>& poly, Object& o, int f, bool* ok) {
  const size_t n = (size_t)c.W * c.H;
  o.mask_AA[f].assign(n, 0);
  o.mask_noAA[f].assign(n, 0);
  double mnx = 1e300, mny = 1e300, mxx = -1e300, mxy = -1e300;
  for (const PointD& p : poly) { mnx = std::min(mnx, p.x); mxx = std::max(mxx, p.x); mny = std::min(mny, p.y); mxy = std::max(mxy, p.y); }
  int x0 = (int)std::floor(std::max(mnx, -4.0)) - 2, y0 = (int)std::floor(std::max(mny, -4.0)) - 2;
  int x1 = (int)std::floor(std::min(mxx, c.W + 4.0)) + 2, y1 = (int)std::floor(std::min(mxy, c.H + 4.0)) + 2;
  x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, c.W - 1); y1 = std::min(y1, c.H - 1);
  if (poly.empty() || x0 > x1 || y0 > y1) { o.bb[f][0] = 0; o.bb[f][1] = 0; o.bb[f][2] = -1; o.bb[f][3] = -1; return; }
  std::vector<uint8_t> cov(n);
  if (!rasterize_polygon(poly, c.W, c.H, cov.data(), false)) *ok = false;
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x) {
      const uint8_t cv = cov[(size_t)y * c.W + x];
      o.mask_AA[f][(size_t)y * c.W + x] = gray8_solid_on_clear(cv);
      o.mask_noAA[f][(size_t)y * c.W + x] = cv >= 128 ? 255 : 0;
    }
  o.bb[f][0] = x0; o.bb[f][1] = y0; o.bb[f][2] = x1; o.bb[f][3] = y1;
}
void full_box(const Ctx& c, Object& o, int f) { o.bb[f][0] = 0; o.bb[f][1] = 0; o.bb[f][2] = c.W - 1; o.bb[f][3] = c.H - 1; }

// renderMasks (DG:465-479 ellipse, DG:520-534 polygon, DG:370-386 warp part).
void render_shape_masks(const Ctx& c, Object& o, const ShapeGeom& g, bool* ok) {
  Affine save = o.intrinsic;
  save *= o.motion;
  std::vector<PointD> p0 = outline(g, o.intrinsic);
  std::vector<PointD> p1 = outline(g, save);
  if (!c.faithful) {
    draw_lean(c, p0, o, 0, ok);
    draw_lean(c, p1, o, 1, ok);
    if (o.has_warp) {  // (the warped mask can be non-zero anywhere)
      std::vector<uint8_t> t;
      apply_warp(o.mask_noAA[1], c.W, c.H, 1, *o.warp, true, t);
      o.mask_noAA[1].swap(t);
      apply_warp(o.mask_AA[1], c.W, c.H, 1, *o.warp, true, t);
      o.mask_AA[1].swap(t);
      full_box(c, o, 1);
    }
    return;
  }
  draw(c, p0, true, o.mask_AA[0], ok);
  draw(c, p0, false, o.mask_noAA[0], ok);
  draw(c, p1, true, o.mask_AA[1], ok);
  draw(c, p1, false, o.mask_noAA[1], ok);
  if (o.has_warp) {
    std::vector<uint8_t> t;
    apply_warp(o.mask_noAA[1], c.W, c.H, 1, *o.warp, true, t);
    o.mask_noAA[1].swap(t);
    apply_warp(o.mask_AA[1], c.W, c.H, 1, *o.warp, true, t);
    o.mask_AA[1].swap(t);
  }
}

// renderTransformedTexture (DG:337-349)
void render_textures(const Ctx& c, Object& o) {
  o.tex[1].resize(o.tex[0].size());
  o.tex[2].resize(o.tex[0].size());
  if (!c.faithful) {
    if (o.bb[0][0] <= o.bb[0][2]) o.tex[1] = o.tex[0];  // identity warp == copy (all fractional weights are 0)
    if (o.has_warp) {
      transformed_texture(o.tex[0].data(), c.W, c.H, o.motion, o.tex[2].data());
      std::vector<uint8_t> t;
      apply_warp(o.tex[2], c.W, c.H, 3, *o.warp, true, t);
      o.tex[2].swap(t);
    } else if (o.bb[1][0] <= o.bb[1][2]) {
      transformed_texture_region(o.tex[0].data(), c.W, c.H, o.motion, o.tex[2].data(), o.bb[1][0], o.bb[1][1], o.bb[1][2], o.bb[1][3]);
    }
    return;
  }
  transformed_texture(o.tex[0].data(), c.W, c.H, Affine(), o.tex[1].data());
  transformed_texture(o.tex[0].data(), c.W, c.H, o.motion, o.tex[2].data());
  if (o.has_warp) {
    std::vector<uint8_t> t;
    apply_warp(o.tex[2], c.W, c.H, 3, *o.warp, true, t);
    o.tex[2].swap(t);
  }
}

// MovingObjectBackground (DG:654-718)
void render_background(const Ctx& c, Object& o) {
  const int W = c.W, H = c.H, W2 = 2 * W, H2 = 2 * H;
  std::vector<uint8_t> t1, t2(o.tex[0].size());
  Affine m = o.intrinsic_inv * o.motion * o.intrinsic;  // DG:673,677
  if (c.faithful) {
    t1.resize(o.tex[0].size());
    transformed_texture(o.tex[0].data(), W2, H2, Affine(), t1.data());
    transformed_texture(o.tex[0].data(), W2, H2, m, t2.data());
  } else {  // identity == copy; of the warped texture only the centre crop is ever used (unless a warp field re-samples it)
    t1 = o.tex[0];
    if (o.has_warp) transformed_texture(o.tex[0].data(), W2, H2, m, t2.data());
    else transformed_texture_region(o.tex[0].data(), W2, H2, m, t2.data(), (int)(W / 2.), (int)(H / 2.), (int)(W / 2.) + W - 1, (int)(H / 2.) + H - 1);
  }
  if (o.has_warp) {
    std::vector<uint8_t> t;
    apply_warp(t2, W2, H2, 3, *o.warp, true, t);
    t2.swap(t);
  }
  // crop(W/2., H/2., W*3./2.-1, H*3./2.-1) (DG:680-681)
  const int cx = (int)(W / 2.), cy = (int)(H / 2.);
  o.tex[1].resize((size_t)3 * W * H);
  o.tex[2].resize((size_t)3 * W * H);
  for (int ch = 0; ch < 3; ++ch)
    for (int y = 0; y < H; ++y) {
      std::memcpy(&o.tex[1][((size_t)ch * H + y) * W], &t1[((size_t)ch * H2 + y + cy) * W2 + cx], W);
      std::memcpy(&o.tex[2][((size_t)ch * H + y) * W], &t2[((size_t)ch * H2 + y + cy) * W2 + cx], W);
    }
  for (int f = 0; f < 2; ++f) {  // renderMasks, DG:684-690
    o.mask_AA[f].assign((size_t)W * H, 255);
    o.mask_noAA[f].assign((size_t)W * H, 255);
    full_box(c, o, f);
  }
}

// getPointFlow (DG:388-407 / DG:692-718)
void point_flow(const Ctx& c, const Object& o, float* x, float* y) {
  if (!o.is_background) {
    double ix = *x, iy = *y;
    float save_x = ix, save_y = iy;
    o.motion.transform(&ix, &iy);
    *x = ix - save_x;
    *y = iy - save_y;
    if (o.has_warp and ix >= 0 and ix < c.W and iy >= 0 and iy < c.H) {
      *x += cimg_linear_atXY_neumann(o.warp->flow.data(), o.warp->w, o.warp->h, ix, iy);
      *y += cimg_linear_atXY_neumann(o.warp->flow.data() + (size_t)o.warp->w * o.warp->h, o.warp->w, o.warp->h, ix, iy);
    }
  } else {
    double ix = *x + c.W / 2, iy = *y + c.H / 2;
    float save_x = ix, save_y = iy;
    o.intrinsic_inv.transform(&ix, &iy);
    o.motion.transform(&ix, &iy);
    o.intrinsic.transform(&ix, &iy);
    *x = ix - save_x;
    *y = iy - save_y;
    if (o.has_warp and ix >= 0 and ix < 2 * c.W and iy >= 0 and iy < 2 * c.H) {
      *x += cimg_linear_atXY_neumann(o.warp->flow.data(), o.warp->w, o.warp->h, ix, iy);
      *y += cimg_linear_atXY_neumann(o.warp->flow.data() + (size_t)o.warp->w * o.warp->h, o.warp->w, o.warp->h, ix, iy);
    }
  }
}

struct Scene {
  std::map<size_t, std::unique_ptr<Object>> objects;  // objects_map (ascending ID)
  std::vector<std::unique_ptr<Object>> components;    // kept alive for inspection
  std::vector<const Object*> shape_order;             // rasterised shapes in realisation order
};

// RealizeObjectBlueprint (DG:1065-1173)
Object* realize(const Ctx& c, const ofdg_blueprint* bps, int bi, const Affine& bg_motion, const Pool& pool,
                WarpSource* warps, Scene& scene, Object* parent, bool* ok) {
  const ofdg_blueprint& p = bps[bi];
  std::unique_ptr<Object> obj(new Object());
  Object* o = obj.get();
  o->id = parent ? 0 : p.obj_id;  // Component classes use ID 0 (DG:542-544, 555-557)
  std::vector<Object*> comps;
  std::vector<bool> comp_modes;
  if (p.obj_type == OFDG_OBJ_COMPOSITE) {
    if (c.mode == 9 and p.do_warpfield_deformation) {  // DG:1120-1128
      o->warp = warps->get_crop_for(p.do_warpfield_deformation);
      o->has_warp = true;
    }
    if (parent) scene.components.push_back(std::move(obj)); else scene.objects[o->id] = std::move(obj);
    for (int k = 0; k < p.n_components; ++k) {
      Object* co = realize(c, bps, p.first_component + k, bg_motion, pool, warps, scene, o, ok);
      comps.push_back(co);
      comp_modes.push_back(bps[p.first_component + k].is_additive_component != 0);
    }
  } else if (p.obj_type == OFDG_OBJ_ELLIPSE or p.obj_type == OFDG_OBJ_POLYGON) {
    if (parent) scene.components.push_back(std::move(obj)); else scene.objects[o->id] = std::move(obj);
  } else {
    *ok = false;  // "(RealizeObjectBlueprint) Bad object type" DG:1143
    return nullptr;
  }
  o->tex[0] = centre_crop(pool, p.tex_id, c.W, c.H);  // DG:1149-1150
  set_intrinsic(*o, p.init_rot, p.init_trans_x, p.init_trans_y);
  set_motion(*o, p.rot, p.scale, p.trans_x, p.trans_y);
  add_background_motion(*o, bg_motion, c.W, c.H);
  if (c.mode == 9 and p.do_warpfield_deformation) {  // DG:1157-1169
    if (parent) {
      o->warp = parent->warp;
      o->has_warp = parent->has_warp;
    } else if (not o->has_warp) {
      o->warp = warps->get_crop_for(p.do_warpfield_deformation);
      o->has_warp = true;
    }
  }
  // Process_UnfinishedObjectContainer (DG:726-732)
  if (!parent && c.faithful) render_textures(c, *o);  // Component*::renderTransformedTexture is empty (DG:546-560)
  if (p.obj_type == OFDG_OBJ_COMPOSITE && !c.faithful) {
    // lean: the same sequential fp32 chain, inside the union of the components' boxes (zero elsewhere)
    const size_t n = (size_t)c.W * c.H;
    for (int f = 0; f < 2; ++f) {
      o->mask_AA[f].assign(n, 0); o->mask_noAA[f].assign(n, 0);
      int x0 = c.W, y0 = c.H, x1 = -1, y1 = -1;
      for (Object* co : comps)
        if (co->bb[f][0] <= co->bb[f][2]) { x0 = std::min(x0, co->bb[f][0]); y0 = std::min(y0, co->bb[f][1]); x1 = std::max(x1, co->bb[f][2]); y1 = std::max(y1, co->bb[f][3]); }
      if (x0 > x1) continue;
      o->bb[f][0] = x0; o->bb[f][1] = y0; o->bb[f][2] = x1; o->bb[f][3] = y1;
      for (size_t ci = 0; ci < comps.size(); ++ci) {
        Object* co = comps[ci];
        for (int y = y0; y <= y1; ++y)
          for (int x = x0; x <= x1; ++x) {
            const size_t i = (size_t)y * c.W + x;
            if (comp_modes[ci]) {
              o->mask_noAA[f][i] = composite_add(o->mask_noAA[f][i], co->mask_noAA[f][i]);
              o->mask_AA[f][i] = composite_add(o->mask_AA[f][i], co->mask_AA[f][i]);
            } else {
              o->mask_noAA[f][i] = composite_sub(o->mask_noAA[f][i], co->mask_noAA[f][i]);
              o->mask_AA[f][i] = composite_sub(o->mask_AA[f][i], co->mask_AA[f][i]);
            }
          }
      }
    }
  } else if (p.obj_type == OFDG_OBJ_COMPOSITE) {
    // MovingObjectComposite::renderMasks (DG:591-646)
    const size_t n = (size_t)c.W * c.H;
    for (int f = 0; f < 2; ++f) { o->mask_AA[f].assign(n, 0); o->mask_noAA[f].assign(n, 0); }
    for (size_t ci = 0; ci < comps.size(); ++ci) {
      Object* co = comps[ci];
      for (int f = 0; f < 2; ++f) {
        for (size_t i = 0; i < n; ++i) {
          if (comp_modes[ci]) {
            o->mask_noAA[f][i] = composite_add(o->mask_noAA[f][i], co->mask_noAA[f][i]);
            o->mask_AA[f][i] = composite_add(o->mask_AA[f][i], co->mask_AA[f][i]);
          } else {
            o->mask_noAA[f][i] = composite_sub(o->mask_noAA[f][i], co->mask_noAA[f][i]);
            o->mask_AA[f][i] = composite_sub(o->mask_AA[f][i], co->mask_AA[f][i]);
          }
        }
      }
    }
  } else {
    scene.shape_order.push_back(o);
    render_shape_masks(c, *o, geom_of(p), ok);
  }
  if (!parent && !c.faithful) render_textures(c, *o);  // (lean: after the masks, whose boxes bound the texture work)
  return o;
}

// Process_TaskBucket (DG:1175-1254) for one task.  bg_tex: optional explicit
// 2W x 2H background texture m_textures[0] (planar BGR); if null, the parity
// boundary's "centre crop" preparation is used (angle 0, zoom 1, shift 0).
bool process_task(const Ctx& c, const ofdg_task& task, const ofdg_blueprint* bps, const Pool& pool,
                  WarpSource* warps, float* img0, float* img1, float* flow, Scene* keep_scene) {
  const int W = c.W, H = c.H;
  const size_t n = (size_t)W * H;
  bool ok = true;
  Scene local;
  Scene& scene = keep_scene ? *keep_scene : local;
  // background (DG:1183-1205)
  const ofdg_blueprint& pb = bps[task.background];
  {
    std::unique_ptr<Object> bg(new Object());
    bg->id = pb.obj_id;
    bg->is_background = true;
    set_intrinsic(*bg, 0.f, W, H);  // DG:662
    bg->tex[0] = c.background_prep == 1 ? prepared_background_two_pass(pool, pb.tex_id, W, H, pb.tex_rot, pb.tex_scale, pb.tex_shift_x, pb.tex_shift_y)
               : c.background_prep ? prepared_background(pool, pb.tex_id, W, H, pb.tex_rot, pb.tex_scale, pb.tex_shift_x, pb.tex_shift_y)
                                   : centre_crop(pool, pb.tex_id, 2 * W, 2 * H);
    set_motion(*bg, pb.rot, pb.scale, pb.trans_x, pb.trans_y);
    if (c.mode == 9 and pb.do_warpfield_deformation) {  // DG:1194-1202
      const WarpCrop* crop = warps->get_crop_for(pb.do_warpfield_deformation);
      bg->own_warp.reset(new WarpCrop(upscale_warp_for_background(*crop, 2 * W, 2 * H)));
      bg->warp = bg->own_warp.get();
      bg->has_warp = true;
    }
    render_background(c, *bg);
    scene.objects[bg->id] = std::move(bg);
  }
  const Affine bg_motion = scene.objects[pb.obj_id]->motion;
  for (int i = 0; i < task.n_objects; ++i)
    realize(c, bps, task.first_object + i, bg_motion, pool, warps, scene, nullptr, &ok);
  if (!ok) return false;

  // RenderCore::blitObject in ascending ID (DG:762-799, 1216-1223)
  std::vector<uint8_t> frame0(3 * n, 0), frame1(3 * n, 0);
  std::vector<size_t> index0(n, 0);
  for (auto& kv : scene.objects) {
    const Object& o = *kv.second;
    if (!c.faithful) {  // lean: the object's boxes only (its masks are zero elsewhere: draw_image leaves d as is)
      const std::vector<uint8_t>& m0 = c.use_AA ? o.mask_AA[0] : o.mask_noAA[0];
      const std::vector<uint8_t>& m1 = c.use_AA ? o.mask_AA[1] : o.mask_noAA[1];
      for (int y = o.bb[0][1]; y <= o.bb[0][3]; ++y)
        for (int x = o.bb[0][0]; x <= o.bb[0][2]; ++x) {
          const size_t i = (size_t)y * W + x;
          if (o.mask_noAA[0][i] == 255) index0[i] = o.id;
          if (m0[i]) for (int ch = 0; ch < 3; ++ch) frame0[ch * n + i] = draw_image_value(frame0[ch * n + i], o.tex[1][ch * n + i], m0[i]);
        }
      for (int y = o.bb[1][1]; y <= o.bb[1][3]; ++y)
        for (int x = o.bb[1][0]; x <= o.bb[1][2]; ++x) {
          const size_t i = (size_t)y * W + x;
          if (m1[i]) for (int ch = 0; ch < 3; ++ch) frame1[ch * n + i] = draw_image_value(frame1[ch * n + i], o.tex[2][ch * n + i], m1[i]);
        }
      continue;
    }
    for (size_t i = 0; i < n; ++i)
      if (o.mask_noAA[0][i] == 255) index0[i] = o.id;
    const std::vector<uint8_t>& m0 = c.use_AA ? o.mask_AA[0] : o.mask_noAA[0];
    const std::vector<uint8_t>& m1 = c.use_AA ? o.mask_AA[1] : o.mask_noAA[1];
    for (int ch = 0; ch < 3; ++ch)
      for (size_t i = 0; i < n; ++i) {
        frame0[ch * n + i] = draw_image_value(frame0[ch * n + i], o.tex[1][ch * n + i], m0[i]);
        frame1[ch * n + i] = draw_image_value(frame1[ch * n + i], o.tex[2][ch * n + i], m1[i]);
      }
  }
  // computeFlowImage(objects_map, false) (DG:801-818)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      size_t idx = index0[(size_t)y * W + x];
      float xf = x, yf = y;
      if (idx == 0) { flow[(size_t)y * W + x] = 0.f; flow[n + (size_t)y * W + x] = 0.f; continue; }
      point_flow(c, *scene.objects[idx], &xf, &yf);
      flow[(size_t)y * W + x] = xf;
      flow[n + (size_t)y * W + x] = yf;
    }
  for (size_t i = 0; i < 3 * n; ++i) {  // DG:1229-1244
    img0[i] = static_cast<float>(frame0[i]);
    img1[i] = static_cast<float>(frame1[i]);
  }
  return true;
}

}  // namespace

extern "C" {

// ---- RNG probes (G2) ----------------------------------------------------------
// kind: 0 FixedRangeUniformInt(a,b) 1 FixedRangeUniformFloat(a,b) 2 Normal(0,1)
//       3 GaussianSq(a,b) 4 Gaussian3(a,b) 5 Gaussian4(a,b) 6 Trigger(a,b,thr=c)
void ofdg_oracle_rng_draws(int kind, int seed, double a, double b, double c, int n, double* out) {
  switch (kind) {
    case 0: { FixedRangeUniformInt r((int)a, (int)b, seed); for (int i = 0; i < n; ++i) out[i] = r(); break; }
    case 1: { FixedRangeUniformFloat r(a, b, seed); for (int i = 0; i < n; ++i) out[i] = r(); break; }
    case 2: { FixedMeanStddevNormalFloat r(0, 1, seed); for (int i = 0; i < n; ++i) out[i] = r(); break; }
    case 3: { GaussianSq r(a, b, seed); for (int i = 0; i < n; ++i) out[i] = r(); break; }
    case 4: { Gaussian3 r(a, b, seed); for (int i = 0; i < n; ++i) out[i] = r(); break; }
    case 5: { Gaussian4 r(a, b, seed); for (int i = 0; i < n; ++i) out[i] = r(); break; }
    case 6: { Trigger r(a, b, c, seed); for (int i = 0; i < n; ++i) out[i] = r(); break; }
  }
}

// ---- sampler ---------------------------------------------------------------------
void* ofdg_oracle_sampler_create(int mode, int W, int H, int num_objects) {
  try { return new Sampler(mode, W, H, num_objects); } catch (...) { return nullptr; }
}
void ofdg_oracle_sampler_destroy(void* s) { delete (Sampler*)s; }
// Samples n_tasks tasks; returns the number of blueprints written or -needed.
int ofdg_oracle_sampler_next(void* s, int n_tasks, ofdg_task* tasks, ofdg_blueprint* bps, int cap) {
  Sampler* sm = (Sampler*)s;
  std::vector<ofdg_blueprint> pool;
  for (int i = 0; i < n_tasks; ++i) sm->next_task(pool, &tasks[i]);
  if ((int)pool.size() > cap) return -(int)pool.size();
  std::memcpy(bps, pool.data(), pool.size() * sizeof(ofdg_blueprint));
  return (int)pool.size();
}

// ---- geometry / rasteriser probes (G3, G4, G5) ----------------------------------
int ofdg_oracle_rasterize(const double* xy, int n, int w, int h, uint8_t* cov) {
  std::vector<PointD> p(n);
  for (int i = 0; i < n; ++i) p[i] = {xy[2 * i], xy[2 * i + 1]};
  return rasterize_polygon(p, w, h, cov, false) ? 0 : -1;
}
int ofdg_oracle_curve3(double x1, double y1, double x2, double y2, double x3, double y3, double* out_xy, int cap) {
  Curve3Div c;
  c.init(x1, y1, x2, y2, x3, y3);
  if ((int)c.pts.size() > cap) return -(int)c.pts.size();
  for (size_t i = 0; i < c.pts.size(); ++i) { out_xy[2 * i] = c.pts[i].x; out_xy[2 * i + 1] = c.pts[i].y; }
  return (int)c.pts.size();
}
// Outline of a blueprint's shape under m (6 doubles sx,shy,shx,sy,tx,ty).
int ofdg_oracle_outline(const ofdg_blueprint* bp, const double* m, double* out_xy, int cap) {
  Affine a(m[0], m[1], m[2], m[3], m[4], m[5]);
  std::vector<PointD> p = outline(geom_of(*bp), a);
  if ((int)p.size() > cap) return -(int)p.size();
  for (size_t i = 0; i < p.size(); ++i) { out_xy[2 * i] = p[i].x; out_xy[2 * i + 1] = p[i].y; }
  return (int)p.size();
}
// span_interpolator_linear + dda2: per-pixel (x_hr, y_hr) of row y, before the -128.
void ofdg_oracle_dda_row(const double* inv, int y, int len, int* out_xy) {
  Affine a(inv[0], inv[1], inv[2], inv[3], inv[4], inv[5]);
  double tx = 0.5, ty = y + 0.5;
  a.transform(&tx, &ty);
  int x1 = iround(tx * 256), y1 = iround(ty * 256);
  tx = 0.5 + len; ty = y + 0.5;
  a.transform(&tx, &ty);
  int x2 = iround(tx * 256), y2 = iround(ty * 256);
  Dda2 lx(x1, x2, len), ly(y1, y2, len);
  for (int i = 0; i < len; ++i) { out_xy[2 * i] = lx.y; out_xy[2 * i + 1] = ly.y; ++lx; ++ly; }
}
void ofdg_oracle_transformed_texture(const uint8_t* in, int tw, int th, const double* m, uint8_t* out) {
  transformed_texture(in, tw, th, Affine(m[0], m[1], m[2], m[3], m[4], m[5]), out);
}
// exhaustive tables for the composite formulas and the blends (65536 entries each)
void ofdg_oracle_tables(uint8_t* add_tbl, uint8_t* sub_tbl, uint8_t* aa_byte) {
  for (int u = 0; u < 256; ++u)
    for (int v = 0; v < 256; ++v) {
      add_tbl[u * 256 + v] = composite_add(u, v);
      sub_tbl[u * 256 + v] = composite_sub(u, v);
    }
  for (int c = 0; c < 256; ++c) aa_byte[c] = gray8_solid_on_clear(c);
}
uint8_t ofdg_oracle_draw_image_value(uint8_t d, uint8_t s, uint8_t m) { return draw_image_value(d, s, m); }

// ---- warp fields (mode 9) ----------------------------------------------------------
// displacers: n x 11 doubles {type, p0, p1, p2, sup_cx, sup_cy, sup_sx, sup_sy, sup_angle, 0, 0}
// (type 0 Translation(dx,dy); 1 Rotation(cx,cy,omega); 2 Zoom(cx,cy,factor)).
int ofdg_oracle_flowfield(int size, const double* displacers, int n, int iters, float* flow, float* iflow) {
  std::vector<DisplacerSpec> d(n);
  for (int i = 0; i < n; ++i) std::memcpy(&d[i], displacers + (size_t)i * 11, sizeof(double) * 9);
  FlowField ff;
  ff.init_from_displacers(size, size, d, iters);
  ff.clamp_near_zeros();
  std::memcpy(flow, ff.flow.data(), ff.flow.size() * 4);
  std::memcpy(iflow, ff.iflow.data(), ff.iflow.size() * 4);
  return 0;
}

// Displacer list of one big field for (W, H, seed): n x 9 doubles (DisplacerSpec); returns n.
int ofdg_oracle_displacers(int W, int H, unsigned seed, double* out, int cap) {
  std::vector<DisplacerSpec> d = make_displacers(W, H, seed);
  if ((int)d.size() > cap) return -(int)d.size();
  for (size_t i = 0; i < d.size(); ++i) std::memcpy(out + 9 * i, &d[i], sizeof(double) * 9);
  return (int)d.size();
}
// One seeded big field -> all its (W+1)x(H+1) crops in the reference's order.
// crops: n x {flow x, flow y, iflow x, iflow y} planes of (H+1)*(W+1) floats. Returns n (or -needed).
int ofdg_oracle_warp_crops(int W, int H, unsigned seed, int iters, float* crops, int cap) {
  const int big = std::max(W, H) * 3;
  std::vector<std::pair<int, int>> org = crop_origins(W, H);
  if ((int)org.size() > cap) return -(int)org.size();
  FlowField ff;
  ff.init_from_displacers(big, big, make_displacers(W, H, seed), iters);
  ff.clamp_near_zeros();
  const int cw = W + 1, ch = H + 1;
  const size_t plane = (size_t)cw * ch, bplane = (size_t)big * big;
  for (size_t k = 0; k < org.size(); ++k) {
    float* dst = crops + k * 4 * plane;
    for (int f = 0; f < 4; ++f) {
      const float* src = (f < 2 ? ff.flow.data() : ff.iflow.data()) + (f & 1) * bplane;
      for (int y = 0; y < ch; ++y)  // get_crop(x, y, x+W, y+H): inclusive
        std::memcpy(dst + f * plane + (size_t)y * cw, src + (size_t)(org[k].second + y) * big + org[k].first, sizeof(float) * cw);
    }
  }
  return (int)org.size();
}

// ---- the hot path ---------------------------------------------------------------------
// flags: bit0 = faithful cost (unused yet: always the reference's per-object structure)
// warp_crops: for mode 9, n_crops crops of (W+1)x(H+1) {flow[2], iflow[2]} served in
// order, each `reuse` times in a row (CropGenerator::get_crop, WarpFields.cpp:516-538).
int ofdg_oracle_render(const ofdg_params* prm, const ofdg_task* tasks, int n_tasks, const ofdg_blueprint* bps,
                       int n_bps, const uint8_t* pool_data, int pool_n, int pool_w, int pool_h,
                       const float* warp_crops, int n_crops, int reuse,
                       float* img0, float* img1, float* flow, int n_threads) {
  (void)n_bps;
  Ctx c{prm->width, prm->height, prm->mode, prm->use_antialiasing != 0, lean_flag() == 0};
  c.background_prep = prm->background_prep;
  Pool pool{pool_n, pool_w, pool_h, pool_data};
  const size_t n = (size_t)c.W * c.H;
  WarpSource warps(warp_crops, n_crops, c.W + 1, c.H + 1, reuse);
  int rc = OFDG_OK;
  if (n_threads <= 1 || c.mode == 9) {
    for (int i = 0; i < n_tasks; ++i)
      if (!process_task(c, tasks[i], bps, pool, &warps, img0 + 3 * n * i, img1 + 3 * n * i, flow + 2 * n * i, nullptr))
        rc = OFDG_EOBJTYPE;
  } else {
    // first_level_threads sample workers (DG:1023-1027); tasks are independent.
    std::vector<std::thread> th;
    std::vector<int> rcs(n_threads, OFDG_OK);
    for (int t = 0; t < n_threads; ++t)
      th.emplace_back([&, t] {
        WarpSource none(nullptr, 0, 0, 0, 0);
        for (int i = t; i < n_tasks; i += n_threads)
          if (!process_task(c, tasks[i], bps, pool, &none, img0 + 3 * n * i, img1 + 3 * n * i, flow + 2 * n * i, nullptr))
            rcs[t] = OFDG_EOBJTYPE;
      });
    for (auto& t : th) t.join();
    for (int r : rcs) if (r != OFDG_OK) rc = r;
  }
  return rc;
}

// Raw masks of the rasterised shapes of ONE task, in realisation order (components
// first, depth-first, as RealizeObjectBlueprint recurses).  kind: 0 AA0 1 AA1 2 noAA0 3 noAA1.
int ofdg_oracle_shape_masks(const ofdg_params* prm, const ofdg_task* task, const ofdg_blueprint* bps,
                            const uint8_t* pool_data, int pool_n, int pool_w, int pool_h,
                            uint8_t* masks, int max_shapes) {
  Ctx c{prm->width, prm->height, prm->mode, prm->use_antialiasing != 0, true};
  c.background_prep = prm->background_prep;
  Pool pool{pool_n, pool_w, pool_h, pool_data};
  const size_t n = (size_t)c.W * c.H;
  std::vector<float> a(3 * n), b(3 * n), f(2 * n);
  Scene scene;
  WarpSource none(nullptr, 0, 0, 0, 0);
  if (!process_task(c, *task, bps, pool, &none, a.data(), b.data(), f.data(), &scene)) return OFDG_EOBJTYPE;
  int k = 0;
  for (const Object* o : scene.shape_order) {
    if (k >= max_shapes) break;
    std::memcpy(masks + ((size_t)k * 4 + 0) * n, o->mask_AA[0].data(), n);
    std::memcpy(masks + ((size_t)k * 4 + 1) * n, o->mask_AA[1].data(), n);
    std::memcpy(masks + ((size_t)k * 4 + 2) * n, o->mask_noAA[0].data(), n);
    std::memcpy(masks + ((size_t)k * 4 + 3) * n, o->mask_noAA[1].data(), n);
    ++k;
  }
  return (int)scene.shape_order.size();
}

// 0 (default): libm, the reference's arithmetic; 1: include/ofdg_detmath.h (the device counter-sampler path's
// definition of sin / cos / expf).  Process-wide; returns the previous value.
int ofdg_oracle_set_detmath(int on) { const int old = detmath_flag(); detmath_flag() = on ? 1 : 0; return old; }
// the functions themselves (CPU tests: accuracy against libm; GPU tests: device == host bit for bit)
void ofdg_oracle_det_sincos(const double* a, int n, double* s, double* c) { for (int i = 0; i < n; ++i) ofdg_det_sincos(a[i], s + i, c + i); }
void ofdg_oracle_det_expf(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) y[i] = ofdg_det_expf(x[i]); }

// CPU-baseline cost model (process-wide; returns the previous value).  0 (default): "faithful" - the reference's
// work pattern; 1: "lean" - the same output bit for bit with one rasterisation per frame and work restricted to
// the outlines' boxes.  Parity tests always run the faithful form.
int ofdg_oracle_set_lean(int on) { const int old = lean_flag(); lean_flag() = on ? 1 : 0; return old; }

int ofdg_oracle_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

// the record of the background preparation chain (test hook; same layout as ofdg_host_bg_prep)
int ofdg_oracle_bg_prep(int pool_w, int pool_h, int width, int height, float angle, float zoom, int shift_x, int shift_y, float* f,
                        int* i) {
  const BgPrep p = make_bg_prep(pool_w, pool_h, width, height, angle, zoom, shift_x, shift_y);
  f[0] = p.ca; f[1] = p.sa; f[2] = p.w2; f[3] = p.h2; f[4] = p.rw2; f[5] = p.rh2; f[6] = p.fx; f[7] = p.fy;
  i[0] = p.x0; i[1] = p.y0; i[2] = p.cw; i[3] = p.ch; i[4] = p.shx; i[5] = p.shy;
  return 0;
}

}  // extern "C"
