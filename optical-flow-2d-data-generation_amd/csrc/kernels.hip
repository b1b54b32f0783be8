// HIP kernels of the render pipeline, written for gfx950 (MI355X, wave64).
//
//   geom_kernel     blueprint shape + 2x3 affine  -> 24.8 fixed-point outline + bbox
//   raster_kernel   outline -> AGG-exact coverage bytes (cells accumulated in LDS with
//                   integer atomics, per-row wavefront prefix sum of `cover`)
//   compose_kernel  background + painter's-order objects -> image0, image1, flow
//
// Everything the reference computes in integers is bit-exact here; fp64 affine
// algebra is evaluated with the reference's operation order and no contraction
// (the file is compiled with -ffp-contract=off).
//
// Reference (lmb-freiburg/optical-flow-2d-data-generation):
//   DG = src/caffe/DataGenerator.cpp.  AGG = Anti-Grain Geometry 2.4 (un-vendored
//   dependency, cmake/Dependencies.cmake:4-19); its algorithms are restated here in
//   closed form so that (edge, scanline) pairs can be processed independently.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/ofdg_detmath.h"
#include "ofdg_device.h"
#include "warpfields.h"

namespace ofdg {

// --------------------------------------------------------------------------
// small helpers
// --------------------------------------------------------------------------
__device__ __forceinline__ int iround_d(double v) {  // agg::iround
  return (int)((v < 0.0) ? v - 0.5 : v + 0.5);
}
__device__ __forceinline__ void xform(const Mat& m, double& x, double& y) {  // trans_affine::transform
  double t = x;
  x = t * m.sx + y * m.shx + m.tx;
  y = t * m.shy + y * m.sy + m.ty;
}
// floor(a / b) for b > 0, |a| < 2^52: fp64 quotient + exact integer correction.
__device__ __forceinline__ long long floordiv64(long long a, long long b) {
  long long q = (long long)floor((double)a / (double)b);
  long long r = a - q * b;
  if (r < 0) --q;
  else if (r >= b) ++q;
  return q;
}

// --------------------------------------------------------------------------
// wave-level helpers (wave64)
// --------------------------------------------------------------------------
// Issue priority.  Every kernel of a step but the background preparation raises its waves' priority: sampler, geom and raster
// are latency-bound (a few hundred waves that gate their chain), compose is memory-bound (few instructions, long waits) - what
// they can issue should go out at once - while the ALU-bound preparation of the OTHER chains fills the issue slots they leave.
// +2.3 % on the headline step, +2.1 % with the composite objects of config 5 (priority for compose alone: +2.9 % / -0.7 .. -3 %;
// profiles/r04_experiments_log.md section 10).  Without the preparation every wave has the same priority and nothing changes.
__device__ __forceinline__ void step_kernel_priority() { __builtin_amdgcn_s_setprio(3); }
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
  return v;
}
// inclusive prefix sum over the 64 lanes: DPP row shifts + row broadcasts (gfx9 DPP)
__device__ __forceinline__ int wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// --------------------------------------------------------------------------
// geom_kernel: one WAVE per (shape, frame), four per workgroup.
// Reference: RealizeObjectBlueprint geometry (DG:1073-1117), conv_transform +
// conv_curve feeding rasterizer_scanline_aa::add_path (DG:465-479, 520-534),
// agg::ellipse (100 steps), agg::curve3_div, ras_conv_int::upscale = iround(v*256).
// --------------------------------------------------------------------------
constexpr int kTileW = 64, kTileH = 16, kPx = 4;
constexpr int kChunkW = 128;  // columns one raster item covers
constexpr int kGeomWaves = 1;   // single-wave workgroups: any retiring compose wave makes room for one
constexpr int kCurveSlots = 10;  // a polygon of <= 20 segments holds <= 9 curve3 segments

// curve3_div::recursive_bezier as an explicit depth-first walk (left subtree first).
// Writes the subdivision points followed by the end point to `out` (24.8 fixed point)
// and returns how many.  `stack` holds the pending right halves: the start of a popped
// half is the end of the half just finished, so 4 doubles + the level suffice.
__device__ __forceinline__ int flatten_curve3(double x1, double y1, double x2, double y2, double x3, double y3,
                                              double (*stack)[5], int2* out, int cap, bool* overflow) {
  const double ex = x3, ey = y3;
  const double tol_sq = 0.25;  // (0.5 / approximation_scale)^2
  int cnt = 0, sp = 0, level = 0;
  for (;;) {
    bool subdivide = false;
    if (level <= 32) {  // curve_recursion_limit
      const double x12 = (x1 + x2) / 2, y12 = (y1 + y2) / 2;
      const double x23 = (x2 + x3) / 2, y23 = (y2 + y3) / 2;
      const double x123 = (x12 + x23) / 2, y123 = (y12 + y23) / 2;
      const double dx = x3 - x1, dy = y3 - y1;
      double d = fabs(((x2 - x3) * dy - (y2 - y3) * dx));
      bool emit = false;
      double px = 0, py = 0;
      if (d > 1e-30) {  // curve_collinearity_epsilon
        if (d * d <= tol_sq * (dx * dx + dy * dy)) { emit = true; px = x123; py = y123; }
        else subdivide = true;
      } else {
        const double da = dx * dx + dy * dy;
        bool stop = false;
        if (da == 0) {
          d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
        } else {
          d = ((x2 - x1) * dx + (y2 - y1) * dy) / da;
          if (d > 0 && d < 1) stop = true;
          else if (d <= 0) d = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2);
          else d = (x3 - x2) * (x3 - x2) + (y3 - y2) * (y3 - y2);
        }
        if (!stop) {
          if (d < tol_sq) { emit = true; px = x2; py = y2; }
          else subdivide = true;
        }
      }
      if (emit) {
        if (cnt < cap - 1) out[cnt++] = make_int2(iround_d(px * 256.0), iround_d(py * 256.0));
        else *overflow = true;
      }
      if (subdivide) {
        if (sp < kCurveMaxDepth) {
          double* st = stack[sp++];
          st[0] = x23; st[1] = y23; st[2] = x3; st[3] = y3; st[4] = (double)(level + 1);
          x3 = x123; y3 = y123; x2 = x12; y2 = y12;  // descend into the left half
          ++level;
          continue;
        }
        *overflow = true;
      }
    }
    if (sp == 0) break;
    const double* st = stack[--sp];
    x1 = x3; y1 = y3;  // the right half starts where the left half ended
    x2 = st[0]; y2 = st[1]; x3 = st[2]; y3 = st[3]; level = (int)st[4];
  }
  out[cnt++] = make_int2(iround_d(ex * 256.0), iround_d(ey * 256.0));
  return cnt;
}

// Block masks: for every 64 x 8 block of every sample two 64-bit words (frame 0, frame 1);
// bit k = the (dilated) box of an outline of foreground object k touches the block.  compose
// reads its block's pair with one scalar load and visits exactly these objects.

// A path of up to 64 segments (lane = segment; type 1 = line_to, 3 = curve3 control point followed by its end point,
// segment 0 = the move_to vertex) flattened the way conv_curve + curve3_div do it.  type_of(i) / point(i, x, y) give a
// segment's type and its point in OUTPUT coordinates.  Returns the vertex count; the box is per lane (reduce it).
template <class OutPtr, class TypeFn, class PointFn>
__device__ __forceinline__ int path_verts(int n_seg, TypeFn type_of, PointFn point, OutPtr out, double (*stack)[kCurveMaxDepth][5],
                                          int2 (*stage)[kCurveMaxPts], uint32_t* __restrict__ err, int lane, int& minx, int& miny,
                                          int& maxx, int& maxy) {
  int n_verts = 0;
  // segment i: 0 = the move_to vertex, Line -> 1 vertex, Curve3 -> its flattening
  // (all but the first point), the Dummy after a Curve3 (its end point) -> nothing
  const int t = (lane >= n_seg) ? 0 : (lane == 0 ? 1 : type_of(lane));
  const bool is_curve = (t == 3);
  const unsigned long long cmask = __ballot(is_curve);
  const int slot = __popcll(cmask & ((1ull << lane) - 1ull));
  bool overflow = false;
  int cnt = 0;
  int2 v0 = make_int2(0, 0);
  if (t == 1) {
    double x, y;
    point(lane, x, y);
    v0 = make_int2(iround_d(x * 256.0), iround_d(y * 256.0));
    cnt = 1;
  } else if (is_curve) {
    if (slot < kCurveSlots) {
      // conv_curve: curve3(ctrl = seg[i], to = seg[i+1]) from the current point seg[i-1]
      double x1, y1, x2, y2, x3, y3;
      const int ie = (lane + 1 < n_seg) ? lane + 1 : lane;
      point(lane - 1, x1, y1); point(lane, x2, y2); point(ie, x3, y3);
      cnt = flatten_curve3(x1, y1, x2, y2, x3, y3, stack[slot], stage[slot], kCurveMaxPts, &overflow);
    } else {
      overflow = true;
    }
  }
  const int incl = wave_scan_incl(cnt);
  n_verts = __shfl(incl, 63, 64);
  const int off = incl - cnt;
  if (n_verts > kMaxVerts) {
    if (lane == 0) atomicOr(err, kErrVertCapacity);
    n_verts = 0;
  } else if (t == 1) {
    out[off] = v0;
    minx = v0.x; maxx = v0.x; miny = v0.y; maxy = v0.y;
  } else if (cnt > 0) {
    for (int k = 0; k < cnt; ++k) {
      const int2 v = stage[slot][k];
      out[off + k] = v;
      minx = min(minx, v.x); maxx = max(maxx, v.x);
      miny = min(miny, v.y); maxy = max(maxy, v.y);
    }
  }
  if (overflow) atomicOr(err, kErrCurveCapacity);
  return n_verts;
}

// The flattened outline of one (shape, frame): vertices in 24.8 fixed point written to `out` (global memory or LDS),
// their number and bounding box (wave-reduced).  One wave; lanes = ellipse steps or path segments.
// Reference: agg::ellipse (100 steps), conv_curve / curve3_div, ras_conv_int::upscale = iround(v * 256).
template <class OutPtr>
__device__ __forceinline__ int outline_verts(const DevShape& S, const Mat& M, const double* __restrict__ cs_tab, OutPtr out,
                                             double (*stack)[kCurveMaxDepth][5], int2 (*stage)[kCurveMaxPts],
                                             uint32_t* __restrict__ err, int lane, int& minx, int& miny, int& maxx, int& maxy) {
  int n_verts = 0;
  minx = 0x7FFFFFFF; miny = 0x7FFFFFFF; maxx = (int)0x80000000; maxy = (int)0x80000000;
  if (S.type == 1) {
    // agg::ellipse::vertex: x = cx + cos(angle)*rx with angle = step/100 * 2*pi; the
    // cos/sin table comes from the host's libm so the doubles match the CPU's.
    n_verts = 100;
    for (int k = lane; k < 100; k += 64) {
      double x = 0.0 + cs_tab[2 * k] * (double)S.rx;
      double y = 0.0 + cs_tab[2 * k + 1] * (double)S.ry;
      xform(M, x, y);
      const int2 v = make_int2(iround_d(x * 256.0), iround_d(y * 256.0));
      out[k] = v;
      minx = min(minx, v.x); maxx = max(maxx, v.x);
      miny = min(miny, v.y); maxy = max(maxy, v.y);
    }
  } else {
    n_verts = path_verts(S.n_seg, [&](int i) { return S.seg_type[i]; },
                         [&](int i, double& x, double& y) { x = (double)S.seg_x[i]; y = (double)S.seg_y[i]; xform(M, x, y); },
                         out, stack, stage, err, lane, minx, miny, maxx, maxy);
  }
  minx = wave_min(minx); miny = wave_min(miny); maxx = wave_max(maxx); maxy = wave_max(maxy);
  return n_verts;
}

// LDS of one geom wave: the curve3 stacks and the staging of their points
struct GeomWs {
  double stack[kCurveSlots][kCurveMaxDepth][5];
  int2 stage[kCurveSlots][kCurveMaxPts];
};
// one wave = one (outline, frame) sf
__device__ __forceinline__ void geom_wave(GeomWs& ws, int sf, int lane, const DevShape* __restrict__ shapes, int n_shapes,
                                          const double* __restrict__ cs_tab, int W, int H, DevShapeFrame* __restrict__ frames,
                                          int2* __restrict__ verts, unsigned long long* __restrict__ blockmask, uint32_t* __restrict__ err,
                                          int* __restrict__ item_count, int4* __restrict__ items, const DevCropRef* __restrict__ crops) {
  if (sf >= n_shapes * 2) return;  // wave-uniform
  const DevShape& S = shapes[sf >> 1];
  if (S.type == 0) return;  // unused slot of a device-sampled batch (wave-uniform)
  const Mat M = S.m[sf & 1];
  int2* out = verts + (size_t)sf * kMaxVerts;

  int minx, miny, maxx, maxy;
  int n_verts = outline_verts(S, M, cs_tab, out, ws.stack, ws.stage, err, lane, minx, miny, maxx, maxy);
  // AGG dx_limit: an edge spanning >= 16384 px takes a different code path in
  // rasterizer_cells_aa::line; blueprints never get close, flag it if one does.
  if (n_verts > 0 && ((long long)maxx - (long long)minx >= (16384LL << 8))) {
    if (lane == 0) atomicOr(err, kErrDxLimit);
    n_verts = 0;
  }
  int x0 = minx >> 8, y0 = miny >> 8, x1 = maxx >> 8, y1 = maxy >> 8;
  const bool visible = !(n_verts < 2 || x1 < 0 || y1 < 0 || x0 > W - 1 || y0 > H - 1);
  int bx0 = 1, by0 = 1, bx1 = 0, by1 = 0;  // box compose / raster work with (dilated for deforming shapes)
  if (visible) {
    x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, W - 1); y1 = min(y1, H - 1);
    bx0 = x0; by0 = y0; bx1 = x1; by1 = y1;
    if ((sf & 1) && S.deform > 0) {
      // mode 9: the frame-1 mask is re-sampled through the inverse warp field, so it can
      // reach max |iflow| (+ the bilinear footprint) beyond the outline's box
      const int d = (int)ceilf(__uint_as_float(*crops[S.deform - 1].max_bits)) + 2;
      bx0 = max(x0 - d, 0); by0 = max(y0 - d, 0); bx1 = min(x1 + d, W - 1); by1 = min(y1 + d, H - 1);
    }
  } else {
    x0 = 1; x1 = 0; y0 = 1; y1 = 0;  // nothing on screen
  }
  if (lane == 0) {
    DevShapeFrame f;
    f.n_verts = n_verts;
    f.x0 = x0; f.y0 = y0; f.x1 = x1; f.y1 = y1;
    f.pad[0] = S.sample; f.pad[1] = S.obj_local; f.pad[2] = 0;  // (raster_kernel marks the block masks with these)
    frames[sf] = f;
  }
  // Raster work list: one item per 8-row band x 128-column chunk of the 64 x 8 blocks the
  // (dilated) box touches.  Coverage outside these blocks is never read: compose tests the
  // block against the same box.
  if (visible) {
    const int band0 = by0 / kBandRows, band1 = by1 / kBandRows;
    // Block masks: raster_kernel marks the blocks in which the outline really has coverage.  Only
    // a deforming frame-1 outline (mode 9) is marked here, by its dilated box: its mask is
    // re-sampled from displaced positions.
    if ((sf & 1) && S.deform > 0) {
      const int nbx = (W + kTileW - 1) / kTileW, nby = (H + kBandRows - 1) / kBandRows;
      const int c0 = bx0 / kTileW, nc = bx1 / kTileW - c0 + 1;
      unsigned long long* m = blockmask + ((size_t)S.sample * nbx * nby) * 2 + (sf & 1);
      const unsigned long long bit = 1ull << S.obj_local;
      for (int i = lane; i < (band1 - band0 + 1) * nc; i += 64)
        atomicOr(m + (size_t)((band0 + i / nc) * nbx + c0 + i % nc) * 2, bit);
    }
    const int xa = (bx0 / kTileW) * kTileW, xb = min((bx1 / kTileW) * kTileW + kTileW - 1, W - 1);
    const int nchunks = (xb - xa + kChunkW) / kChunkW;
    const int n_items = (band1 - band0 + 1) * nchunks;
    int base = 0;
    if (lane == 0) base = atomicAdd(item_count, n_items);
    base = __shfl(base, 0, 64);
    for (int i = lane; i < n_items; i += 64) {
      const int b = band0 + i / nchunks, cx = xa + (i % nchunks) * kChunkW;
      items[base + i] = make_int4(sf, b, cx, min(cx + kChunkW - 1, xb));
    }
  }
}

__global__ __launch_bounds__(64 * kGeomWaves) void geom_kernel(const DevShape* __restrict__ shapes, int n_shapes,
                                                   const double* __restrict__ cs_tab, int W, int H,
                                                   DevShapeFrame* __restrict__ frames, int2* __restrict__ verts,
                                                   unsigned long long* __restrict__ blockmask, uint32_t* __restrict__ err,
                                                   int* __restrict__ item_count, int4* __restrict__ items,
                                                   const DevCropRef* __restrict__ crops) {
  __shared__ GeomWs s_ws[kGeomWaves];
  step_kernel_priority();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sf = __builtin_amdgcn_readfirstlane(blockIdx.x * kGeomWaves + wave);
  geom_wave(s_ws[wave], sf, lane, shapes, n_shapes, cs_tab, W, H, frames, verts, blockmask, err, item_count, items, crops);
}

// --------------------------------------------------------------------------
// raster_kernel: persistent workgroups walk the item list; one item = one outline x one
// band of kBandRows scanlines x the padded column range [X0, X1].
//
// AGG's rasterizer_cells_aa walks each edge scanline by scanline and cell by cell
// with incremental lift/rem/mod stepping.  Its per-cell sums of (cover, area) are
// order independent, and the stepping has a closed form (floor of the cumulative
// rational), so every (edge, scanline) pair is processed by its own thread and
// accumulated into LDS with integer atomics.  Column 0 of a row collects the cover
// of all cells left of the column range.
// Then rasterizer_scanline_aa::sweep_scanline: running cover prefix sum per row
// (wavefront DPP scan), alpha = min(|((C << 9) - area) >> 9|, 255) (non-zero rule,
// gamma_none).  The thresholded (gamma_threshold 0.5) mask is alpha >= 128.
// Reference: MovingObjectBase::draw, DG:351-368.
// --------------------------------------------------------------------------
struct CellAcc {
  int* cover;  // [kBandRows][pitch]
  int* area;
  int pitch, X0, X1;
  __device__ __forceinline__ void add(int r, int cell, int dcover, int darea) const {
    if (cell > X1 || dcover == 0) return;  // (darea is a multiple of dcover)
    if (cell < X0) {
      atomicAdd(&cover[r * pitch], dcover);
    } else {
      const int i = r * pitch + (cell - X0 + 1);
      atomicAdd(&cover[i], dcover);
      atomicAdd(&area[i], darea);
    }
  }
};

// ---- 32-bit fast path ---------------------------------------------------------------
// For edges shorter than 8192 px horizontally and 1024 px vertically every product in
// the closed forms fits 32 bits, and all floor divisions can use a float reciprocal
// followed by an exact integer correction.
// floor(num / den), den >= 1, |true quotient| <= 4096: one estimate, one +-1 fix.
__device__ __forceinline__ int fdiv_small(int num, int den, float rcp) {
  int q = (int)floorf((float)num * rcp);
  const int r = num - q * den;
  if (r < 0) --q;
  else if (r >= den) ++q;
  return q;
}
// floor(num / den), den >= 1, |num| < 2^29: two-step estimate (the first remainder is
// small enough for the second estimate to be within one of the truth).
__device__ __forceinline__ int fdiv_big(int num, int den, float rcp) {
  const int q0 = (int)floorf((float)num * rcp);
  const int r0 = num - q0 * den;
  return q0 + fdiv_small(r0, den, rcp);
}

// render_hline in closed form, both directions in one code path.
template <class Acc>
__device__ __forceinline__ void hline32(const Acc& acc, int r, int xa, int ya, int xb, int yb) {
  if (ya == yb) return;
  const int exa = xa >> 8, exb = xb >> 8;
  const int fxa = xa & 255, fxb = xb & 255;
  const int Dy = yb - ya;
  if (exa == exb) {
    acc.add(r, exa, Dy, (fxa + fxb) * Dy);
    return;
  }
  const int lo = min(exa, exb), hi = max(exa, exb);
  if (hi < acc.X0) { acc.add(r, acc.X0 - 1, Dy, 0); return; }  // entirely left: only its cover counts
  if (lo > acc.X1) return;                                      // entirely right: no effect
  const bool right = xb > xa;
  const int adx = right ? xb - xa : xa - xb;
  const float rcp = __frcp_rn((float)adx);
  const int m = hi - lo;
  const int f0 = right ? 256 - fxa : fxa;
  const int j_lo = right ? max(0, acc.X0 - exa) : max(0, exa - acc.X1);
  const int j_hi = right ? min(m, acc.X1 - exa) : min(m, exa - acc.X0);
  // y where the hline enters cell j (j = 1..m): ya + floor((f0 + 256 (j-1)) Dy / adx)
  int yprev = (j_lo == 0) ? ya : ya + fdiv_small((f0 + 256 * (j_lo - 1)) * Dy, adx, rcp);
  if (right && j_lo > 0) acc.add(r, acc.X0 - 1, yprev - ya, 0);  // cells left of the range
  for (int j = j_lo; j <= j_hi; ++j) {
    const int ynext = (j == m) ? yb : ya + fdiv_small((f0 + 256 * j) * Dy, adx, rcp);
    const int d = ynext - yprev;
    const int fin = (j == 0) ? fxa : (right ? 0 : 256);
    const int fout = (j == m) ? fxb : (right ? 256 : 0);
    acc.add(r, right ? exa + j : exa - j, d, (fin + fout) * d);
    yprev = ynext;
  }
  if (!right && j_hi < m) acc.add(r, acc.X0 - 1, yb - yprev, 0);  // what remains lies left of the range
}

// rasterizer_cells_aa::render_hline(ey, xa, ya, xb, yb) in closed form.
template <class Acc>
__device__ __forceinline__ void hline(const Acc& acc, int r, int xa, int ya, int xb, int yb) {
  if (ya == yb) return;
  const int exa = xa >> 8, exb = xb >> 8;
  const int fxa = xa & 255, fxb = xb & 255;
  const int Dy = yb - ya;
  if (exa == exb) {
    acc.add(r, exa, Dy, (fxa + fxb) * Dy);
    return;
  }
  if (xb > xa) {
    const int m = exb - exa;
    const long long dxh = xb - xa;
    const long long f0 = 256 - fxa;
    int j = 0;
    int yprev = ya;
    // cells left of the bbox only contribute their total cover
    if (exa < acc.X0) {
      const int jb = min(acc.X0 - exa, m + 1);  // first cell index that is inside (or past the end)
      // y at the left boundary of cell exa+jb (or yb if the hline ends before it)
      int yb0 = (jb > m) ? yb : ya + (int)floordiv64((f0 + 256LL * (jb - 1)) * Dy, dxh);
      acc.add(r, acc.X0 - 1, yb0 - ya, 0);
      yprev = yb0;
      j = jb;
    }
    for (; j <= m; ++j) {
      const int cell = exa + j;
      if (cell > acc.X1) break;
      const int ynext = (j == m) ? yb : ya + (int)floordiv64((f0 + 256LL * j) * Dy, dxh);
      const int d = ynext - yprev;
      const int fin = (j == 0) ? fxa : 0;
      const int fout = (j == m) ? fxb : 256;
      acc.add(r, cell, d, (fin + fout) * d);
      yprev = ynext;
    }
  } else {
    const int m = exa - exb;
    const long long dxh = xa - xb;
    const long long f0 = fxa;
    // walking leftwards: cells right of X1 are skipped, cells left of X0 are lumped
    int j = 0;
    int yprev = ya;
    if (exa > acc.X1) {
      const int jb = min(exa - acc.X1, m + 1);
      yprev = (jb > m) ? yb : ya + (int)floordiv64((f0 + 256LL * (jb - 1)) * Dy, dxh);
      j = jb;
    }
    for (; j <= m; ++j) {
      const int cell = exa - j;
      if (cell < acc.X0) {
        acc.add(r, acc.X0 - 1, yb - yprev, 0);  // everything that remains, in one go
        break;
      }
      const int ynext = (j == m) ? yb : ya + (int)floordiv64((f0 + 256LL * j) * Dy, dxh);
      const int d = ynext - yprev;
      const int fin = (j == 0) ? fxa : 256;
      const int fout = (j == m) ? fxb : 0;
      acc.add(r, cell, d, (fin + fout) * d);
      yprev = ynext;
    }
  }
}

// rasterizer_cells_aa::line restricted to scanline y (closed form of the stepping).
template <class Acc>
__device__ __forceinline__ void edge_scanline(const Acc& acc, int r, int y, int x1, int y1, int x2, int y2) {
  const int ey1 = y1 >> 8, ey2 = y2 >> 8;
  const int fy1 = y1 & 255, fy2 = y2 & 255;
  {
    const int dxi = x2 - x1, dyi = y2 - y1;
    const int adx = dxi < 0 ? -dxi : dxi, ady = dyi < 0 ? -dyi : dyi;
    if (adx < (1 << 21) && ady < (1 << 18)) {  // 32-bit fast path (virtually every edge)
      if (ey1 == ey2) {
        if (y == ey1) hline32(acc, r, x1, fy1, x2, fy2);
        return;
      }
      const bool up = dyi < 0;
      if (y < min(ey1, ey2) || y > max(ey1, ey2)) return;
      const int k = up ? ey1 - y : y - ey1;
      const bool last = (y == ey2);
      const float rcp = __frcp_rn((float)ady);
      // X(k) = x1 + floor((base + 256 (k-1)) dx / |dy|)
      //      = x1 + d1 + (k-1) lift + floor((m1 + (k-1) rem) / |dy|)   (exact identity)
      const int p1 = (up ? fy1 : 256 - fy1) * dxi;
      const int d1 = fdiv_big(p1, ady, rcp), m1 = p1 - d1 * ady;
      const int lift = fdiv_big(256 * dxi, ady, rcp), rem = 256 * dxi - lift * ady;
      const int xa = (k == 0) ? x1 : x1 + d1 + (k - 1) * lift + fdiv_small(m1 + (k - 1) * rem, ady, rcp);
      const int xb = last ? x2 : x1 + d1 + k * lift + fdiv_small(m1 + k * rem, ady, rcp);
      const int ya = (k == 0) ? fy1 : (up ? 256 : 0);
      const int yb = last ? fy2 : (up ? 0 : 256);
      hline32(acc, r, xa, ya, xb, yb);
      return;
    }
  }
  if (ey1 == ey2) {
    if (y == ey1) hline(acc, r, x1, fy1, x2, fy2);
    return;
  }
  const long long dx = x2 - x1;
  if (y2 > y1) {
    if (y < ey1 || y > ey2) return;
    const long long dy = y2 - y1;
    const int k = y - ey1;
    const int xa = (k == 0) ? x1 : x1 + (int)floordiv64(((256 - fy1) + 256LL * (k - 1)) * dx, dy);
    const int ya = (k == 0) ? fy1 : 0;
    const int xb = (y == ey2) ? x2 : x1 + (int)floordiv64(((256 - fy1) + 256LL * k) * dx, dy);
    const int yb = (y == ey2) ? fy2 : 256;
    hline(acc, r, xa, ya, xb, yb);
  } else {
    if (y > ey1 || y < ey2) return;
    const long long dy = y1 - y2;
    const int k = ey1 - y;
    const int xa = (k == 0) ? x1 : x1 + (int)floordiv64((fy1 + 256LL * (k - 1)) * dx, dy);
    const int ya = (k == 0) ? fy1 : 256;
    const int xb = (y == ey2) ? x2 : x1 + (int)floordiv64((fy1 + 256LL * k) * dx, dy);
    const int yb = (y == ey2) ? fy2 : 0;
    hline(acc, r, xa, ya, xb, yb);
  }
}

struct ChunkAcc {
  int* cover;  // [kBandRows][kChunkW]
  int* area;
  int* carry;  // [kBandRows]: cover of all cells left of the chunk
  int X0, X1;
  __device__ __forceinline__ void add(int r, int cell, int dcover, int darea) const {
    if (cell > X1 || dcover == 0) return;  // (darea is a multiple of dcover)
    if (cell < X0) {
      atomicAdd(&carry[r], dcover);
    } else {
      const int i = r * kChunkW + (cell - X0);
      atomicAdd(&cover[i], dcover);
      atomicAdd(&area[i], darea);
    }
  }
};

constexpr int kRasterWaves = 1;
struct ChunkCells {
  int cover[kBandRows][kChunkW];
  int area[kBandRows][kChunkW];
  int carry[kBandRows];
};

// LDS of one rasterising wave
struct RasterWs {
  ChunkCells cells;
  int queue[64 * kBandRows];
};

// One item = one outline x one band of kBandRows scanlines x the padded column range [X0, X1], by ONE wave (no
// workgroup barriers): clear its cells, accumulate every (edge, scanline) pair of the outline that can reach the
// chunk, sweep, store the coverage bytes, mark the block masks.  v: the outline's nv vertices; F*: its pixel box.
// pa, pb: v[lane] and v[lane + 1], requested by the caller while the previous item was being rasterised.
__device__ __forceinline__ void raster_item(RasterWs& ws, const int2* __restrict__ v, int nv, int Fx0, int Fy0, int Fx1, int Fy1, int sf,
                                            int band, int X0, int X1, int W, int H, uint8_t* __restrict__ cov,
                                            unsigned long long* __restrict__ blockmask, int sample, int obj_local, int lane, int2 pa, int2 pb) {
  ChunkCells& tc = ws.cells;
  const int by0 = band * kBandRows;
  const int rows = min(kBandRows, H - by0);
  const bool has_edges = (Fx0 <= Fx1) && (by0 <= Fy1) && (by0 + kBandRows - 1 >= Fy0) && (Fx0 <= X1) && (Fx1 >= X0);
  uint8_t* dst = cov + ((size_t)sf * H + by0) * W + X0;
  const int c2 = 2 * lane;  // this lane's two columns of the chunk
  const bool in_range = (X0 + c2) <= X1;  // X0 is even and X1 odd (tile-aligned, W % 4 == 0)
  if (!has_edges) {
    if (in_range)
      for (int r = 0; r < rows; ++r) *reinterpret_cast<uint16_t*>(dst + (size_t)r * W + c2) = 0;
    return;
  }
  {  // clear
    int4* z = reinterpret_cast<int4*>(&tc);
    const int n16 = (2 * kBandRows * kChunkW) / 4;
    for (int i = lane; i < n16; i += 64) z[i] = make_int4(0, 0, 0, 0);
    if (lane < kBandRows) tc.carry[lane] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  ChunkAcc acc;
  acc.cover = &tc.cover[0][0]; acc.area = &tc.area[0][0]; acc.carry = &tc.carry[0];
  acc.X0 = X0; acc.X1 = X1;
  // Lane = edge: find the scanlines of the band each edge crosses, compact the
  // (edge, scanline) pairs that can reach the chunk into a dense queue, then let
  // all 64 lanes work on real pairs.  The closing edge (last -> first) is edge nv-1.
  int* queue = ws.queue;
  for (int e0 = 0; e0 < nv; e0 += 64) {
    const int e = e0 + lane;
    int n_rows = 0, rlo = 0;
    // this lane's edge (a -> b); the first 64 edges' vertices are in hand already
    int2 a = pa, b = pb;
    if (e0 > 0 && e < nv) { a = v[e]; b = v[(e + 1 == nv) ? 0 : e + 1]; }  // (never past the outline's slot: nv may be kMaxVerts)
    {
      const int v0x = __shfl(pa.x, 0, 64), v0y = __shfl(pa.y, 0, 64);  // the closing edge ends at vertex 0
      if (e + 1 == nv) b = make_int2(v0x, v0y);
    }
    if (e < nv) {
      const int eya = a.y >> 8, eyb = b.y >> 8;
      rlo = max(min(eya, eyb), by0);
      const int rhi = min(max(eya, eyb), by0 + rows - 1);
      if (rhi >= rlo && (min(a.x, b.x) >> 8) <= X1 && a.y != b.y) n_rows = rhi - rlo + 1;
    }
    const int incl = wave_scan_incl(n_rows);
    const int total = __shfl(incl, 63, 64);
    int at = incl - n_rows;
    for (int k = 0; k < n_rows; ++k) queue[at++] = (e << 4) | (rlo + k - by0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (int q0 = 0; q0 < total; q0 += 64) {
      const int q = q0 + lane;
      const int code = (q < total) ? queue[q] : (e0 << 4);
      const int src = (code >> 4) - e0, r = code & 15;   // the pair's edge sits in lane `src` of this block: no second trip to memory
      const int ax = __shfl(a.x, src, 64), ay = __shfl(a.y, src, 64), bx = __shfl(b.x, src, 64), by = __shfl(b.y, src, 64);
      if (q < total) edge_scanline(acc, r, by0 + r, ax, ay, bx, by);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  // sweep: lane l owns columns 2l, 2l+1 of every row
  int nonzero = 0;
  for (int r = 0; r < rows; ++r) {
    const int2 cv = *reinterpret_cast<const int2*>(&tc.cover[r][c2]);
    const int2 ar = *reinterpret_cast<const int2*>(&tc.area[r][c2]);
    const int s2 = cv.x + cv.y;
    const int base = tc.carry[r] + wave_scan_incl(s2) - s2;
    int a0 = ((base + cv.x) << 9) - ar.x, a1 = ((base + s2) << 9) - ar.y;  // calculate_alpha (shift 9)
    a0 >>= 9; a1 >>= 9;
    a0 = a0 < 0 ? -a0 : a0; a1 = a1 < 0 ? -a1 : a1;
    a0 = a0 > 255 ? 255 : a0; a1 = a1 > 255 ? 255 : a1;
    if (in_range) { *reinterpret_cast<uint16_t*>(dst + (size_t)r * W + c2) = (uint16_t)(a0 | (a1 << 8)); nonzero |= a0 | a1; }
  }
  // lanes 0-31 hold the left 64 x 8 block of the chunk, lanes 32-63 the right one
  const unsigned long long nz = __ballot(nonzero != 0);
  if (blockmask) {
    // compose visits an object in a 64 x 8 block only if its outline has coverage there
    const int half = lane >> 5;
    if ((lane & 31) == 0 && ((nz >> (32 * half)) & 0xFFFFFFFFull)) {
      const int nbx = (W + kTileW - 1) / kTileW, nby = (H + kBandRows - 1) / kBandRows;
      atomicOr(blockmask + ((size_t)(sample * nby + band) * nbx + (X0 / kTileW + half)) * 2 + (sf & 1), 1ull << obj_local);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // reads done before the next item's clear
}

// raster_kernel: persistent single-wave workgroups walk geom_kernel's item list.
__global__ __launch_bounds__(64 * kRasterWaves) void raster_kernel(const DevShapeFrame* __restrict__ frames,
                                                     const int4* __restrict__ items,
                                                     const int* __restrict__ item_count,
                                                     const int2* __restrict__ verts, int W, int H,
                                                     uint8_t* __restrict__ cov,
                                                     unsigned long long* __restrict__ blockmask_next, int n_mask_words,
                                                     unsigned long long* __restrict__ blockmask) {
  __shared__ __attribute__((aligned(16))) RasterWs s_ws[kRasterWaves];
  step_kernel_priority();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // clear the block masks the NEXT launch of this slot accumulates into (the other parity)
  for (int gid = blockIdx.x * blockDim.x + threadIdx.x; gid < n_mask_words; gid += gridDim.x * blockDim.x) blockmask_next[gid] = 0ull;
  const int n_items = *item_count;
  const int n_waves = gridDim.x * kRasterWaves;
  // An item is three dependent trips to memory before its first instruction of work (the item, its outline's record, the
  // vertices): the loop is a software pipeline - item k + 2, the record of item k + 1 and its first 64 vertices are requested
  // before item k is rasterised.  The work list and the records are constant while this kernel runs (scalar loads).
  typedef OFDG_CONSTANT const int4 ConstItem;
  typedef OFDG_CONSTANT const DevShapeFrame ConstFrame;
  int it = __builtin_amdgcn_readfirstlane(blockIdx.x * kRasterWaves + wave);
  if (it >= n_items) return;
  // (what is carried around the loop is a per-lane value to the compiler: readfirstlane says "uniform" again)
  struct Item { int sf, band, X0, X1; };
  struct Frame { int nv, x0, y0, x1, y1, sample, obj_local; };
  auto item_at = [&](int i) {
    const int4 q = ((ConstItem*)items)[i];
    return Item{q.x, q.y, q.z, q.w};
  };
  auto frame_of = [&](int sf) {
    const ConstFrame& f = ((ConstFrame*)frames)[sf];
    return Frame{f.n_verts, f.x0, f.y0, f.x1, f.y1, f.pad[0], f.pad[1]};
  };
  auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  int it1 = it + n_waves;
  Item I0 = item_at(it), I1 = item_at(it1 < n_items ? it1 : it);
  Frame F0 = frame_of(I0.sf);
  int2 pa = verts[(size_t)I0.sf * kMaxVerts + lane], pb = verts[(size_t)I0.sf * kMaxVerts + lane + 1];
  for (;;) {
    I0 = Item{uni(I0.sf), uni(I0.band), uni(I0.X0), uni(I0.X1)};
    I1 = Item{uni(I1.sf), uni(I1.band), uni(I1.X0), uni(I1.X1)};
    F0 = Frame{uni(F0.nv), uni(F0.x0), uni(F0.y0), uni(F0.x1), uni(F0.y1), uni(F0.sample), uni(F0.obj_local)};
    it1 = uni(it1);
    const int it2 = it1 + n_waves;
    const Item I2 = item_at(it2 < n_items ? it2 : it1 < n_items ? it1 : it);
    const Frame F1 = frame_of(I1.sf);
    const int2 na = verts[(size_t)I1.sf * kMaxVerts + lane], nb = verts[(size_t)I1.sf * kMaxVerts + lane + 1];
    raster_item(s_ws[wave], verts + (size_t)I0.sf * kMaxVerts, F0.nv, F0.x0, F0.y0, F0.x1, F0.y1, I0.sf, I0.band, I0.X0, I0.X1, W, H, cov, blockmask,
                F0.sample, F0.obj_local, lane, pa, pb);
    if (it1 >= n_items) break;
    I0 = I1; F0 = F1; I1 = I2; pa = na; pb = nb;
    it = it1; it1 = it2;
  }
}

// --------------------------------------------------------------------------
// compose_kernel
// --------------------------------------------------------------------------
// pixfmt_gray8 solid blend of colour 255 onto a cleared buffer (AGG 2.4): the AA mask byte.
__device__ __forceinline__ int aa_byte(int c) { return c == 255 ? 255 : (255 * c) >> 8; }
// floor(a / 255) for 0 <= a <= 65025
__device__ __forceinline__ int div255(int a) { return (a + 1 + (a >> 8)) >> 8; }
// CImg::draw_image(sprite, mask, 1, 255): d = floor((m*s + (255-m)*d) / 255)
__device__ __forceinline__ int blend(int d, int s, int m) { return div255(m * s + (255 - m) * d); }

// The same blend on a packed B | G<<8 | R<<16 pixel: B and R share one multiply-add
// (16-bit lanes: m*s + (255-m)*d <= 65025 never carries), G goes alone.
__device__ __forceinline__ uint32_t blend_px(uint32_t d, uint32_t s, uint32_t m) {
  const uint32_t im = 255u - m;
  uint32_t rb = m * (s & 0x00FF00FFu) + im * (d & 0x00FF00FFu);
  uint32_t g = m * ((s >> 8) & 255u) + im * ((d >> 8) & 255u);
  rb = ((rb + 0x00010001u + ((rb >> 8) & 0x00FF00FFu)) >> 8) & 0x00FF00FFu;  // floor(x / 255) per lane
  g = (g + 1u + (g >> 8)) >> 8;
  return rb | (g << 8);
}

// MovingObjectComposite::renderMasks, strict fp32 (DG:606, 626)
// (float)u / 255.f, correctly rounded, for a byte u: the product with rn(1/255) is wrong for 126 of the 256 bytes, one
// Newton step on it (exact residual through an fma) is right for all of them (tests: the exhaustive byte-formula test
// against the oracle's division; tools/check_q255.py derives it).  3 VALU instructions instead of the ~10 of a division -
// a composite evaluates 32 of these per component and pixel quad.
__device__ __forceinline__ float byte_over_255(int u) {
  const float fu = (float)u, c = 0x1.010102p-8f;  // rn(1 / 255)
  const float q0 = __fmul_rn(fu, c);
  return __fmaf_rn(__fmaf_rn(-q0, 255.f, fu), c, q0);
}
__device__ __forceinline__ int comp_add(int u, int v) {
  const float fu = byte_over_255(u), fv = byte_over_255(v);
  const float t = __fmul_rn(__fsub_rn(1.f, fu), __fsub_rn(1.f, fv));
  return (int)(unsigned char)__fmul_rn(255.f, __fsub_rn(1.f, t));
}
__device__ __forceinline__ int comp_sub(int u, int v) {
  const float fu = byte_over_255(u), fv = byte_over_255(v);
  return (int)(unsigned char)__fmul_rn(255.f, __fmul_rn(fu, __fsub_rn(1.f, fv)));
}

// --------------------------------------------------------------------------
// Load helpers of the texture warps
// --------------------------------------------------------------------------
// a wave-uniform pointer the compiler can see is uniform (SGPR pair): loads take it as scalar base + 32-bit lane offset
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
// ... and the loads through it name the GLOBAL address space (a pointer rebuilt from integers is generic: flat_load with a
// 64-bit address per lane; as global memory it is global_load v, voffset32, s[base])
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const u32x2_t g_uint2;
typedef __attribute__((address_space(1))) const uint32_t g_uint32;
typedef __attribute__((address_space(1))) const char g_char;
__device__ __forceinline__ uint2 gload2(const char* base, uint32_t off) {
  asm("" : "+v"(off));  // (the offset stays a 32-bit VGPR: a select folded into a 64-bit phi would defeat the saddr form)
  const u32x2_t v = *(g_uint2*)((g_char*)base + off);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint32_t gload1(const char* base, uint32_t off) { return *(g_uint32*)((g_char*)base + off); }
// span_interpolator_linear::begin + dda2_line_interpolator for one output row.
struct RowDDA {
  int x1, lx, rx;  // start, lft, rem (rem in [1, n])
  int y1, ly, ry;
};
template <bool kPow2 = false>
__device__ __forceinline__ void dda_setup(int a, int b, int n, int nshift, int& lft, int& rem) {
  const int D = b - a;
  if (kPow2 || nshift >= 0) {  // n is a power of two (wave-uniform): floor quotient / remainder by shifts
    const int q = D >> nshift, r = D & (n - 1);
    lft = r > 0 ? q : q - 1;  // dda2_line_interpolator: rem in [1, n]
    rem = r > 0 ? r : n;
  } else {
    lft = D / n;
    rem = D % n;
    if (rem <= 0) { rem += n; lft--; }
  }
}
template <bool kPow2 = false>
__device__ __forceinline__ RowDDA make_row(const Mat& inv, int row, int len, int nshift) {
  RowDDA R;
  double tx = 0 + 0.5, ty = row + 0.5;
  xform(inv, tx, ty);
  R.x1 = iround_d(tx * 256.0);
  R.y1 = iround_d(ty * 256.0);
  tx = 0 + 0.5 + len; ty = row + 0.5;
  xform(inv, tx, ty);
  const int x2 = iround_d(tx * 256.0), y2 = iround_d(ty * 256.0);
  dda_setup<kPow2>(R.x1, x2, len, nshift, R.lx, R.rx);
  dda_setup<kPow2>(R.y1, y2, len, nshift, R.ly, R.ry);
  return R;
}
// value of the interpolator after i increments
template <bool kPow2 = false>
__device__ __forceinline__ int dda_at(int y1, int lft, int rem, int n, int nshift, int i) {
  const int a = (i + 1) * rem + n - 1;
  const int q = (kPow2 || nshift >= 0) ? (a >> nshift) : (a / n);
  return y1 + i * lft + q - 1;
}
template <bool kPow2 = false>
__device__ __forceinline__ int wrap_reflect(int v, int size, int size2, int mask2, int& raw) {
  int m;
  if (kPow2 || mask2 >= 0) {  // power-of-two period (wave-uniform)
    m = v & mask2;
  } else if ((unsigned)(v + size2) < 3u * (unsigned)size2) {  // within one period of the image: no division
    m = v;
    if (m < 0) m += size2;
    if (m >= size2) m -= size2;
  } else {
    m = v % size2;
    if (m < 0) m += size2;
  }
  raw = m;
  return m >= size ? size2 - 1 - m : m;
}
__device__ __forceinline__ int wrap_next(int raw, int size, int size2) {
  int m = raw + 1;
  if (m >= size2) m = 0;
  return m >= size ? size2 - 1 - m : m;
}

struct WarpGeom {
  int tw, th;        // texture size the warp runs on (fg: W x H, bg: 2W x 2H)
  int tw2, th2;      // 2 * size
  int mx2, my2;      // size2 - 1 if power of two else -1
  int nshift;        // log2(tw) if power of two else -1
  int pitch;         // pool image pitch in texels
};

// span_image_filter_rgb_bilinear with wrap_mode_reflect: returns packed B | G<<8 | R<<16.
template <bool kPow2 = false>
__device__ __forceinline__ uint32_t sample_bilinear(const uint32_t* __restrict__ tex, const WarpGeom& g,
                                                    const RowDDA& R, int i) {
  int x_hr = dda_at<kPow2>(R.x1, R.lx, R.rx, g.tw, g.nshift, i) - 128;
  int y_hr = dda_at<kPow2>(R.y1, R.ly, R.ry, g.tw, g.nshift, i) - 128;
  const int x_lr = x_hr >> 8, y_lr = y_hr >> 8;
  x_hr &= 255; y_hr &= 255;
  int rx, ry;
  const int xa = wrap_reflect<kPow2>(x_lr, g.tw, g.tw2, g.mx2, rx);  // (the width's period is a power of two with it)
  const int ya = wrap_reflect(y_lr, g.th, g.th2, g.my2, ry);
  const int xb = wrap_next(rx, g.tw, g.tw2);
  const int yb = wrap_next(ry, g.th, g.th2);
  const uint32_t ra = (uint32_t)(ya * g.pitch), rb = (uint32_t)(yb * g.pitch);
  const uint32_t p00 = tex[ra + (uint32_t)xa];
  const uint32_t p10 = tex[ra + (uint32_t)xb];
  const uint32_t p01 = tex[rb + (uint32_t)xa];
  const uint32_t p11 = tex[rb + (uint32_t)xb];
  const uint32_t w00 = (256 - x_hr) * (256 - y_hr), w10 = x_hr * (256 - y_hr);
  const uint32_t w01 = (256 - x_hr) * y_hr, w11 = x_hr * y_hr;
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sh = 8 * c;
    uint32_t fg = 32768u + w00 * ((p00 >> sh) & 255u) + w10 * ((p10 >> sh) & 255u) +
                  w01 * ((p01 >> sh) & 255u) + w11 * ((p11 >> sh) & 255u);
    out |= (fg >> 16) << sh;
  }
  return out;
}

// The same filter for kPx consecutive pixels of one row.  kPow2 (the row length is a power of
// two, e.g. 512 or 1024): the interpolators advance by additions, and when every needed lane
// of the wave stays inside the texture (no reflection: checked at the two end pixels, the
// interpolators are monotone) the four taps of a pixel are two 8-byte loads.  Otherwise, and
// for other row lengths, sample_bilinear.
__device__ __forceinline__ uint32_t bilerp_rgb(uint2 t0, uint2 t1, uint32_t xf, uint32_t yf) {
  // sum w_ij * p_ij with w = (256 - xf | xf) * (256 - yf | yf): vertical pass on 16-bit lanes
  // (B and R share a multiply: (256 - yf) * a + yf * b <= 65280), horizontal pass per channel
  const uint32_t ify = 256u - yf, ifx = 256u - xf;
  const uint32_t rb0 = (t0.x & 0x00FF00FFu) * ify + (t1.x & 0x00FF00FFu) * yf;
  const uint32_t rb1 = (t0.y & 0x00FF00FFu) * ify + (t1.y & 0x00FF00FFu) * yf;
  const uint32_t g0 = ((t0.x >> 8) & 255u) * ify + ((t1.x >> 8) & 255u) * yf;
  const uint32_t g1 = ((t0.y >> 8) & 255u) * ify + ((t1.y >> 8) & 255u) * yf;
  const uint32_t b = (rb0 & 0xFFFFu) * ifx + (rb1 & 0xFFFFu) * xf + 32768u;
  const uint32_t r = (rb0 >> 16) * ifx + (rb1 >> 16) * xf + 32768u;
  const uint32_t g = g0 * ifx + g1 * xf + 32768u;
  return (b >> 16) | ((g >> 8) & 0xFF00u) | (r & 0xFF0000u);
}
// CImg<float>::_linear_atXY (Neumann): clamp, nx = dx > 0 ? x+1 : x, fp32 polynomial.
template <class Ptr>
__device__ __forceinline__ float linear_neumann(Ptr img, int w, int h, float fx, float fy) {
  const float nfx = fx <= 0 ? 0 : (fx >= (float)(w - 1) ? (float)(w - 1) : fx);
  const float nfy = fy <= 0 ? 0 : (fy >= (float)(h - 1) ? (float)(h - 1) : fy);
  const unsigned x = (unsigned)nfx, y = (unsigned)nfy;
  const float dx = nfx - (float)x, dy = nfy - (float)y;
  const unsigned nx = dx > 0 ? x + 1 : x, ny = dy > 0 ? y + 1 : y;
  const float Icc = img[(size_t)y * w + x], Inc = img[(size_t)y * w + nx];
  const float Icn = img[(size_t)ny * w + x], Inn = img[(size_t)ny * w + nx];
  return Icc + dx * (Inc - Icc + dy * (Icc + Inn - Icn - Inc)) + dy * (Icn - Icc);
}

// ---- mode 9 helpers: CImg<unsigned char>::linear_atXY(fx, fy, 0, c, 0) (Dirichlet), fp32 ----
struct Taps {
  int x, y;      // integer tap (x, y); the others are x+1 / y+1
  float dx, dy;  // fractions
  bool ok;       // false: NaN displacement -> the result is 0 (SURVEY F-9)
};
__device__ __forceinline__ Taps make_taps(float fx, float fy) {
  Taps t;
  t.ok = (fx == fx) && (fy == fy) && fabsf(fx) < 1e8f && fabsf(fy) < 1e8f;
  if (!t.ok) { fx = 0.f; fy = 0.f; }
  t.x = (int)fx - (fx >= 0 ? 0 : 1);
  t.y = (int)fy - (fy >= 0 ? 0 : 1);
  t.dx = fx - (float)t.x;
  t.dy = fy - (float)t.y;
  return t;
}
__device__ __forceinline__ int lerp_u8(const Taps& t, float Icc, float Inc, float Icn, float Inn) {
  const float v = Icc + t.dx * (Inc - Icc + t.dy * (Icc + Inn - Icn - Inc)) + t.dy * (Icn - Icc);
  return t.ok ? (int)(unsigned char)v : 0;
}

// Warp crops as the kernels read them: two interleaved planes of w*h float pairs - (flow x, flow y), then
// (iflow x, iflow y) - so that a pixel's displacement is ONE 8-byte load.
__device__ __forceinline__ float2 crop_pair(const DevCropRef& C, int plane_pair, int x, int y) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = *(OFDG_GLOBAL const f32x2*)(C.data + ((size_t)plane_pair * C.w * C.h + (size_t)y * C.w + x) * 2);
  return make_float2(v.x, v.y);
}
// CImg<float>::_linear_atXY (Neumann) of both components of the forward field at once (four 8-byte taps)
__device__ __forceinline__ float2 linear_neumann2(const DevCropRef& C, float fx, float fy) {
  const int w = C.w, h = C.h;
  const float nfx = fx <= 0 ? 0 : (fx >= (float)(w - 1) ? (float)(w - 1) : fx);
  const float nfy = fy <= 0 ? 0 : (fy >= (float)(h - 1) ? (float)(h - 1) : fy);
  const unsigned x = (unsigned)nfx, y = (unsigned)nfy;
  const float dx = nfx - (float)x, dy = nfy - (float)y;
  const unsigned nx = dx > 0 ? x + 1 : x, ny = dy > 0 ? y + 1 : y;
  const float2 Icc = crop_pair(C, 0, (int)x, (int)y), Inc = crop_pair(C, 0, (int)nx, (int)y);
  const float2 Icn = crop_pair(C, 0, (int)x, (int)ny), Inn = crop_pair(C, 0, (int)nx, (int)ny);
  return make_float2(Icc.x + dx * (Inc.x - Icc.x + dy * (Icc.x + Inn.x - Icn.x - Inc.x)) + dy * (Icn.x - Icc.x),
                     Icc.y + dx * (Inc.y - Icc.y + dy * (Icc.y + Inn.y - Icn.y - Inc.y)) + dy * (Icn.y - Icc.y));
}
// applyWarpFieldToTexture(getTransformedTexture(tex, motion), iwarp) for one pixel (DG:237-252, 341-345, 670-678): the
// four texels of the TRANSFORMED texture around the displaced position t (each a bilinear filter of the source through
// the inverse motion `inv`; Dirichlet 0 outside the tw x th texture), then CImg's fp32 lerp per channel.  While every
// source tap of every lane that needs one lies inside the texture the eight tap pairs of a pixel are eight 8-byte loads
// issued together; otherwise the general interpolator (reflection), sample by sample.
template <bool kPow2>
__device__ __forceinline__ uint32_t deform_texel(const uint32_t* __restrict__ tex_, const WarpGeom& g, const Mat& inv, const Taps& t, bool need) {
  const uint32_t* __restrict__ tex = uniform_ptr(tex_);
  uint32_t tap[4] = {0, 0, 0, 0};
  int xh[4], yh[4];
  bool valid[4];
  RowDDA R[2];
  bool fast = true;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ty = t.y + j;
    const bool row_ok = need && t.ok && ty >= 0 && ty < g.th;
    R[j] = make_row<kPow2>(inv, row_ok ? ty : 0, g.tw, g.nshift);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tx = t.x + i, k = 2 * j + i;
      valid[k] = row_ok && tx >= 0 && tx < g.tw;
      xh[k] = dda_at<kPow2>(R[j].x1, R[j].lx, R[j].rx, g.tw, g.nshift, valid[k] ? tx : 0) - 128;
      yh[k] = dda_at<kPow2>(R[j].y1, R[j].ly, R[j].ry, g.tw, g.nshift, valid[k] ? tx : 0) - 128;
      const bool in_range = (unsigned)(xh[k] >> 8) <= (unsigned)(g.tw - 2) && (unsigned)(yh[k] >> 8) <= (unsigned)(g.th - 2);
      fast = fast && (!valid[k] || in_range);
    }
  }
  if (__ballot(!fast) == 0ull) {
    const char* base = reinterpret_cast<const char*>(tex);
    const char* base1 = uniform_ptr(base + (size_t)g.pitch * 4u);
    uint2 t0[4], t1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t off = valid[k] ? ((uint32_t)(yh[k] >> 8) * (uint32_t)g.pitch + (uint32_t)(xh[k] >> 8)) * 4u : 0u;
      t0[k] = gload2(base, off);
      t1[k] = gload2(base1, off);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tap[k] = valid[k] ? bilerp_rgb(t0[k], t1[k], (uint32_t)xh[k] & 255u, (uint32_t)yh[k] & 255u) : 0u;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) tap[k] = valid[k] ? sample_bilinear<kPow2>(tex, g, R[k >> 1], t.x + (k & 1)) : 0u;
  }
  uint32_t o = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sh = 8 * c;
    o |= (uint32_t)lerp_u8(t, (float)((tap[0] >> sh) & 255u), (float)((tap[1] >> sh) & 255u),
                           (float)((tap[2] >> sh) & 255u), (float)((tap[3] >> sh) & 255u)) << sh;
  }
  return o;
}

struct Taps4 {
  uint2 t0[kPx], t1[kPx];  // texel pairs of the two rows (mode 2: t0[p].x = the finished pixel)
  uint32_t xf, yf;         // fractions, byte p = pixel p
  int mode;                // 0 paired loads, 1 reflected single loads, 2 general interpolator (nothing issued)
};
// the four frame-1 texels of a lane (span_image_filter_rgb_bilinear along the row's interpolator), first half: addresses + loads
__device__ __forceinline__ Taps4 taps_issue(const uint32_t* __restrict__ tex_, const WarpGeom& g, const RowDDA& R, int i0, bool need) {
  const uint32_t* __restrict__ tex = uniform_ptr(tex_);
  Taps4 T;
  int xh[kPx], yh[kPx];
  {
    int ax = (i0 + 1) * R.rx + g.tw - 1, bx = R.x1 - 129 + i0 * R.lx;  // dda_at(...) - 128
    int ay = (i0 + 1) * R.ry + g.tw - 1, by = R.y1 - 129 + i0 * R.ly;
#pragma unroll
    for (int p = 0; p < kPx; ++p) {
      xh[p] = bx + (ax >> g.nshift); yh[p] = by + (ay >> g.nshift);
      ax += R.rx; bx += R.lx; ay += R.ry; by += R.ly;
    }
  }
  T.xf = ((uint32_t)xh[0] & 255u) | (((uint32_t)xh[1] & 255u) << 8) | (((uint32_t)xh[2] & 255u) << 16) | ((uint32_t)xh[3] << 24);
  T.yf = ((uint32_t)yh[0] & 255u) | (((uint32_t)yh[1] & 255u) << 8) | (((uint32_t)yh[2] & 255u) << 16) | ((uint32_t)yh[3] << 24);
  const bool in_range = (unsigned)(xh[0] >> 8) <= (unsigned)(g.tw - 2) && (unsigned)(xh[kPx - 1] >> 8) <= (unsigned)(g.tw - 2) &&
                        (unsigned)(yh[0] >> 8) <= (unsigned)(g.th - 2) && (unsigned)(yh[kPx - 1] >> 8) <= (unsigned)(g.th - 2);
  if (__ballot(need && !in_range) == 0ull) {
    T.mode = 0;
    const char* base = reinterpret_cast<const char*>(tex);
    const char* base1 = uniform_ptr(base + (size_t)g.pitch * 4u);
#pragma unroll
    for (int p = 0; p < kPx; ++p) {
      const uint32_t off = need ? ((uint32_t)(yh[p] >> 8) * (uint32_t)g.pitch + (uint32_t)(xh[p] >> 8)) * 4u : 0u;
      T.t0[p] = gload2(base, off);
      T.t1[p] = gload2(base1, off);
    }
  } else {
    const int tw2 = g.tw2, th2 = g.th2;
    bool one_period = true;
#pragma unroll
    for (int p = 0; p < kPx; p += kPx - 1)
      one_period = one_period && (unsigned)((xh[p] >> 8) + tw2) < 3u * (unsigned)tw2 - 1u && (unsigned)((yh[p] >> 8) + th2) < 3u * (unsigned)th2 - 1u;
    if (__ballot(need && !one_period) == 0ull) {
      T.mode = 1;
      auto reflect = [](int v, int size, int size2) {  // wrap_mode_reflect for -size2 <= v < 2 * size2
        int m = v < 0 ? v + size2 : v;
        m = m >= size2 ? m - size2 : m;
        return m >= size ? size2 - 1 - m : m;
      };
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        const int xl = need ? xh[p] >> 8 : 0, yl = need ? yh[p] >> 8 : 0;
        const uint32_t xa = (uint32_t)reflect(xl, g.tw, tw2), xb = (uint32_t)reflect(xl + 1, g.tw, tw2);
        const uint32_t ra = (uint32_t)reflect(yl, g.th, th2) * (uint32_t)g.pitch, rb = (uint32_t)reflect(yl + 1, g.th, th2) * (uint32_t)g.pitch;
        const char* base = reinterpret_cast<const char*>(tex);
        T.t0[p] = make_uint2(gload1(base, (ra + xa) * 4u), gload1(base, (ra + xb) * 4u));
        T.t1[p] = make_uint2(gload1(base, (rb + xa) * 4u), gload1(base, (rb + xb) * 4u));
        __builtin_amdgcn_sched_barrier(0);  // one pixel's addresses at a time: this (rare) path must not set the kernel's register count
      }
    } else {
      T.mode = 2;  // absurd motions only: the general interpolator, pixel by pixel, right here
      uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;  // (a rolled loop and selects: this path must not set the kernel's register count)
#pragma unroll 1
      for (int p = 0; p < kPx; ++p) {
        const uint32_t v = sample_bilinear<true>(tex, g, R, i0 + p);
        r0 = p == 0 ? v : r0; r1 = p == 1 ? v : r1; r2 = p == 2 ? v : r2; r3 = p == 3 ? v : r3;
      }
      T.t0[0] = make_uint2(r0, 0); T.t0[1] = make_uint2(r1, 0); T.t0[2] = make_uint2(r2, 0); T.t0[3] = make_uint2(r3, 0);
#pragma unroll
      for (int p = 0; p < kPx; ++p) T.t1[p] = make_uint2(0, 0);
    }
  }
  return T;
}
__device__ __forceinline__ void taps_finish(const Taps4& T, uint32_t out[kPx]) {
  if (T.mode == 2) {  // (wave-uniform)
#pragma unroll
    for (int p = 0; p < kPx; ++p) out[p] = T.t0[p].x;
  } else {
#pragma unroll
    for (int p = 0; p < kPx; ++p) out[p] = bilerp_rgb(T.t0[p], T.t1[p], (T.xf >> (8 * p)) & 255u, (T.yf >> (8 * p)) & 255u);
  }
}

// --------------------------------------------------------------------------
// compose_rigid: the compose kernel of every mode (kDeform compiles the mode-9 paths in), written around the latency of a
// strip: what a wave waits for is fetched in as few dependent round trips as the data allows.  One single-wave workgroup
// renders a 64 x 4 strip, a lane kPx = 4 horizontally adjacent pixels; objects are visited in painter's order (ascending
// ID) through the block's object masks; their coverage comes from the slots raster_kernel filled (valid over every
// touched block, zero outside the outlines).
// Reference: Process_TaskBucket DG:1216-1245, blitObject DG:762-799, computeFlowImage/getPointFlow DG:801-818, 388-407, 692-718.
//   scalar stage 1   sample record (background matrices inline) + the block's object masks   [kernel arguments are
//                    preloaded into SGPRs: leading scalar parameters, -amdgpu-kernarg-preload-count]
//   scalar stage 2   headers (coverage slot, texture origin, kind) of the first kPre objects of the mask
//   vector stage 1   background texels of both frames, coverage of those objects, their full records (lane i reads
//                    dword i; the matrices are moved to SGPRs with v_readlane when the visit needs them)
//   per visit        texture taps of both frames (one round trip), blend, flow
//   stores           8 fp32 planes, 16-byte non-temporal stores
// Mode 9 (kDeform): a deformed object's frame-1 mask, its frame-1 texture and its flow - and the same for a deformed
// background - are re-sampled through the object's warp crop, pixel by pixel, inside the same visit (DG:370-386, 341-345,
// 403-406, 670-678, 714-717); everything rigid in a mode-9 batch takes the rigid path above.  120 VGPRs = four waves per
// SIMD (the former separate mode-9 body: 158 = three); sharing the displacement loads between the mask and the texture
// of a visit costs the fourth wave (149 VGPRs) and more than it saves.
// --------------------------------------------------------------------------
constexpr int kPre = 2;  // objects of a block whose header / coverage / record are fetched ahead of their visit
#ifndef OFDG_DEFORM_FENCE
#define OFDG_DEFORM_FENCE 1
#endif

template <bool kPow2, bool kDeform = false>
__device__ __forceinline__ void compose_rigid(const DevSample* __restrict__ samples, const unsigned long long* __restrict__ blockmask,
                                              const DevObject* __restrict__ objects, const uint8_t* __restrict__ cov,
                                              int n_strips, int tiles_x, int tiles_y, int W, int H, int use_aa, int bg_pitch, int fg_pitch,
                                              const uint32_t* __restrict__ pool, const uint32_t* __restrict__ bgpool,
                                              float* __restrict__ img0, float* __restrict__ img1, float* __restrict__ flow,
                                              const DevShapeFrame* __restrict__ frames, int* __restrict__ item_count,
                                              const DevCropRef* __restrict__ crops = nullptr) {
  static_assert(kPx == 4, "mask bytes are packed four to a word");
  step_kernel_priority();
  if (blockIdx.x == 0 && threadIdx.x == 0) *item_count = 0;  // raster_kernel has consumed the work list
  // XCD-aware strip mapping: blocks b and b + 8 share an XCD (round-robin dispatch).  Every XCD takes every 8th run of 32
  // consecutive strips (= one 64 x 16 tile row of 8 tiles): neighbouring strips share their background rows, coverage and
  // object records in that XCD's L2, and the foreground-heavy samples are spread over all XCDs.  One wave per workgroup: a
  // finished wave's slot is refilled at once, not when the slowest of four sibling waves retires.
  int wg = blockIdx.x;
  if (wg < (n_strips & ~255)) {
    const int xcd = wg & 7, slot = wg >> 3;
    wg = (((slot >> 5) * 8 + xcd) << 5) + (slot & 31);
  }
  const int tile = wg >> 2, sub = wg & 3;
  const int tiles = tiles_x * tiles_y;
  const int s = tile / tiles;
  const int t = tile - s * tiles;
  const int trow = t / tiles_x;
  const int ty0 = trow * kTileH, tx0 = (t - trow * tiles_x) * kTileW;
  const int lane = (int)threadIdx.x;
  const int x0 = tx0 + (lane & 15) * kPx;
  const int y = ty0 + sub * 4 + (lane >> 4);
  const bool inside = (x0 < W) && (y < H);

  // ---- scalar stage 1: sample (+ background) record and block masks, requested in ONE batch ----
  // (the compiler loads a struct field where its first use is; an empty asm statement that names every value right here
  //  makes that one place: one s_waitcnt for the whole record instead of one per use site)
  struct SmpRec { int first_object, first_shape; Mat bg_motion, bg_tex_inv; unsigned long long bg_tex_base; int bg_deform; } smp;
  unsigned long long mask0, mask1;
  {
    const DevSample& R = samples[s];
    smp.first_object = R.first_object; smp.first_shape = R.first_shape;
    smp.bg_motion = R.bg_motion; smp.bg_tex_inv = R.bg_tex_inv; smp.bg_tex_base = R.bg_tex_base;
    smp.bg_deform = kDeform ? R.bg_deform : 0;
    const int nby = (H + kBandRows - 1) / kBandRows;
    const int brow = (ty0 + (sub >> 1) * kBandRows) / kBandRows;
    const ulonglong2 mm = *reinterpret_cast<const ulonglong2*>(blockmask + ((size_t)(s * nby + min(brow, nby - 1)) * tiles_x + tx0 / kTileW) * 2);
    mask0 = mm.x; mask1 = mm.y;
    asm volatile("" : "+s"(smp.first_object), "+s"(smp.first_shape), "+s"(smp.bg_tex_base), "+s"(mask0), "+s"(mask1),
                      "+s"(smp.bg_tex_inv.sx), "+s"(smp.bg_tex_inv.shy), "+s"(smp.bg_tex_inv.shx), "+s"(smp.bg_tex_inv.sy),
                      "+s"(smp.bg_tex_inv.tx), "+s"(smp.bg_tex_inv.ty));
    asm volatile("" : "+s"(smp.bg_motion.sx), "+s"(smp.bg_motion.shy), "+s"(smp.bg_motion.shx), "+s"(smp.bg_motion.sy),
                      "+s"(smp.bg_motion.tx), "+s"(smp.bg_motion.ty));
  }
  unsigned long long omask = mask0 | mask1;
  const DevObject* objs = objects + smp.first_object;
  const uint32_t pix = (uint32_t)(y * W + x0);
  const size_t slot_bytes = (size_t)W * H;
  constexpr int kRecLast = kDeform ? 31 : 25;  // last dword of an object's record a visit needs (31: DevObject.deform)
  static_assert(offsetof(DevObject, deform) == 31 * 4 && offsetof(DevObject, tex_base) == 24 * 4, "record dwords");

  // ---- scalar stage 2: outline slots of the first kPre objects of the mask (sample record: scalar-cache hits) ----
  const uint32_t* shape_tab = reinterpret_cast<const uint32_t*>(samples[s].shape_of);
  auto shape_entry = [&](int oi) -> uint32_t {  // oi >= 1; two 16-bit entries per dword
    const uint32_t w = shape_tab[(oi - 1) >> 1];
    return ((oi - 1) & 1) ? (w >> 16) : (w & 0xFFFFu);
  };
  int pre_oi[kPre];
  uint32_t pre_sh[kPre];
  {
    unsigned long long m = omask;
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      pre_oi[k] = m ? __ffsll((long long)m) : 0;
      m &= m - 1;  // (0 stays 0)
      pre_sh[k] = pre_oi[k] ? shape_entry(pre_oi[k]) : (uint32_t)kShapeComposite;
    }
  }

  // ---- vector stage 1: EVERY load the wave knows how to address goes out before anything is waited for: background
  // taps of frame 1, background texels of frame 0, coverage + records of the first kPre objects.  Lanes outside the frame
  // (W, H not multiples of the strip) read the texel at the origin instead of branching around the loads. ----
  WarpGeom gb;
  gb.tw = 2 * W; gb.th = 2 * H; gb.tw2 = 4 * W; gb.th2 = 4 * H;
  gb.mx2 = ((gb.tw2 & (gb.tw2 - 1)) == 0) ? gb.tw2 - 1 : -1;
  gb.my2 = ((gb.th2 & (gb.th2 - 1)) == 0) ? gb.th2 - 1 : -1;
  gb.nshift = ((gb.tw & (gb.tw - 1)) == 0) ? (31 - __clz(gb.tw)) : -1;
  gb.pitch = bg_pitch;
  const uint32_t* btex = bgpool + smp.bg_tex_base;
  const int yy = y + H / 2, xx = x0 + W / 2;
  uint4 bq = make_uint4(0, 0, 0, 0);
  RowDDA Rb;
  Taps4 Tb;
  Rb = make_row<kPow2>(smp.bg_tex_inv, inside ? yy : H / 2, gb.tw, gb.nshift);
  if constexpr (kPow2) Tb = taps_issue(btex, gb, Rb, xx, inside);
  {
    uint32_t boff = inside ? (uint32_t)(yy * gb.pitch + xx) * 4u : 0u;
    asm("" : "+v"(boff));
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t v = *(__attribute__((address_space(1))) const u32x4_t*)((g_char*)reinterpret_cast<const char*>(btex) + boff);  // frame 0: identity warp == copy (DG:667-668, 680)
    bq = make_uint4(v.x, v.y, v.z, v.w);
  }
  uint32_t pre_c0[kPre], pre_c1[kPre], pre_rec[kPre];
#pragma unroll
  for (int k = 0; k < kPre; ++k) {
    pre_c0[k] = 0; pre_c1[k] = 0; pre_rec[k] = 0;
    if (!(pre_sh[k] & kShapeComposite)) {  // a simple object (wave-uniform)
      const uint8_t* c = cov + (size_t)(smp.first_shape + (int)pre_sh[k]) * 2 * slot_bytes;
      if (inside) {
        if ((mask0 >> (pre_oi[k] - 1)) & 1ull) pre_c0[k] = *reinterpret_cast<const uint32_t*>(c + pix);
        if ((mask1 >> (pre_oi[k] - 1)) & 1ull) pre_c1[k] = *reinterpret_cast<const uint32_t*>(c + slot_bytes + pix);
      }
      pre_rec[k] = reinterpret_cast<const uint32_t*>(&objs[pre_oi[k]])[min(lane, kRecLast)];  // motion, tex_inv, tex_base (mode 9: .. deform)
    }
  }

  // ---- background: frames start as its textures (masks are all 255), flow of every pixel ----
  uint32_t px0[kPx], px1[kPx];
  float fu[kPx], fv[kPx];
  if (inside) {
    if constexpr (kPow2) taps_finish(Tb, px1);
    else {
#pragma unroll
      for (int p = 0; p < kPx; ++p) px1[p] = sample_bilinear(btex, gb, Rb, xx + p);
    }
    const uint32_t tt[4] = {bq.x, bq.y, bq.z, bq.w};
    // MovingObjectBackground::getPointFlow (DG:692-718): T(-W,-H), motion, T(W,H)
    const double by = (double)(y + H / 2) + (double)(-H);
#pragma unroll
    for (int p = 0; p < kPx; ++p) {
      px0[p] = tt[p] & 0x00FFFFFFu;
      double ix = (double)(x0 + p + W / 2), iy = by;
      const float save_x = (float)(x0 + p + W / 2), save_y = (float)(y + H / 2);
      ix = ix + (double)(-W);
      xform(smp.bg_motion, ix, iy);
      ix = ix + (double)W; iy = iy + (double)H;
      fu[p] = (float)(ix - (double)save_x);
      fv[p] = (float)(iy - (double)save_y);
    }
    if constexpr (kDeform) {
      if (smp.bg_deform > 0) {  // background re-sampled through its (2W x 2H, upscaled) warp crop (DG:670-678, 714-717)
        const DevCropRef C = crops[smp.bg_deform - 1];
        const int X0 = x0 + W / 2, Y = y + H / 2;
#pragma unroll 1
        for (int p = 0; p < kPx; ++p) {
          const int X = X0 + p;
          const float2 iw = crop_pair(C, 1, X, Y);
          const Taps t = make_taps((float)X + iw.x, (float)Y + iw.y);
          px1[p] = deform_texel<kPow2>(btex, gb, smp.bg_tex_inv, t, true);
          // flow: + forward field at the destination (detour coordinates), Neumann
          double ix = (double)(x0 + p + W / 2) + (double)(-W), iy = by;
          xform(smp.bg_motion, ix, iy);
          ix = ix + (double)W; iy = iy + (double)H;
          if (ix >= 0 && ix < (double)(2 * W) && iy >= 0 && iy < (double)(2 * H)) {
            const float2 f = linear_neumann2(C, (float)ix, (float)iy);
            fu[p] += f.x;
            fv[p] += f.y;
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < kPx; ++p) { fu[p] = fv[p] = 0.f; px0[p] = px1[p] = 0; }
  }

  // ---- foreground objects in z-order ----
  WarpGeom g_frame;
  g_frame.tw = W; g_frame.th = H; g_frame.tw2 = 2 * W; g_frame.th2 = 2 * H;
  g_frame.mx2 = ((g_frame.tw2 & (g_frame.tw2 - 1)) == 0) ? g_frame.tw2 - 1 : -1;
  g_frame.my2 = ((g_frame.th2 & (g_frame.th2 - 1)) == 0) ? g_frame.th2 - 1 : -1;
  g_frame.nshift = ((W & (W - 1)) == 0) ? (31 - __clz(W)) : -1;
  g_frame.pitch = fg_pitch;
  int vi = 0;  // visit number
  while (omask) {
    const int oi = __ffsll((long long)omask);  // 1-based == index into objs[]
    omask &= omask - 1;
    const bool has0 = (mask0 >> (oi - 1)) & 1ull, has1 = (mask1 >> (oi - 1)) & 1ull;  // wave-uniform
    // (opaque copies: the int -> double conversions of the pixel coordinates are redone per visit instead of being
    //  hoisted out of the loop, where they would hold two dozen registers across every visit)
    int xv = x0, yv = y;
    asm volatile("" : "+v"(xv), "+v"(yv));
    // (mode 9, likewise: what depends only on the frame size or the lane - the doubles of W / H and + 0.5 of the row set-up, record
    //  and coverage addresses - is redone per visit instead of being held across all of them: 111 instead of 120 VGPRs)
    int Wv = W, Hv = H, lanev = lane;
    if constexpr (kDeform && OFDG_DEFORM_FENCE) asm volatile("" : "+s"(Wv), "+s"(Hv), "+v"(lanev));
    WarpGeom g_visit;
    if constexpr (kDeform && OFDG_DEFORM_FENCE) {
      g_visit.tw = Wv; g_visit.th = Hv; g_visit.tw2 = 2 * Wv; g_visit.th2 = 2 * Hv;
      g_visit.mx2 = ((g_visit.tw2 & (g_visit.tw2 - 1)) == 0) ? g_visit.tw2 - 1 : -1;
      g_visit.my2 = ((g_visit.th2 & (g_visit.th2 - 1)) == 0) ? g_visit.th2 - 1 : -1;
      g_visit.nshift = ((Wv & (Wv - 1)) == 0) ? (31 - __clz(Wv)) : -1;
      g_visit.pitch = fg_pitch;
    }
    const WarpGeom& g = (kDeform && OFDG_DEFORM_FENCE) ? g_visit : g_frame;
    const int lane = lanev;
    const int W = Wv, H = Hv;
    uint32_t sh, c0w = 0, c1w = 0, recw = 0;
    if (vi < kPre) {
      sh = vi == 0 ? pre_sh[0] : pre_sh[kPre - 1];
      c0w = vi == 0 ? pre_c0[0] : pre_c0[kPre - 1];
      c1w = vi == 0 ? pre_c1[0] : pre_c1[kPre - 1];
      recw = vi == 0 ? pre_rec[0] : pre_rec[kPre - 1];
    } else {
      sh = shape_entry(oi);
      if (!(sh & kShapeComposite)) {
        const uint8_t* c = cov + (size_t)(smp.first_shape + (int)sh) * 2 * slot_bytes;
        if (inside) {
          if (has0) c0w = *reinterpret_cast<const uint32_t*>(c + pix);
          if (has1) c1w = *reinterpret_cast<const uint32_t*>(c + slot_bytes + pix);
        }
        recw = reinterpret_cast<const uint32_t*>(&objs[oi])[min(lane, kRecLast)];
      }
    }
    ++vi;
    static_assert(kPre == 2 || kPre == 1, "the selects above pick between two prefetched sets");

    // mode 9: frame-1 mask bytes (AA and thresholded) of one outline re-sampled through the inverse field
    // (MovingObjectBase::renderMasks, DG:370-386).  Taps outside the outline's rasterised box read as 0 (the mask is 0
    // there; outside the frame: Dirichlet).
    auto warped_mask1 = [&](int shape, const DevShapeFrame& F, const DevCropRef& C, int p, int& aa, int& na) {
      aa = 0; na = 0;
      if (!inside || !has1) return;
      if (F.x0 > F.x1) return;
      const int x = xv + p;
      const float2 iw = crop_pair(C, 1, x, yv);
      const Taps t = make_taps((float)x + iw.x, (float)yv + iw.y);
      if (!t.ok || t.x + 1 < F.x0 || t.x > F.x1 || t.y + 1 < F.y0 || t.y > F.y1) return;
      const uint8_t* c = cov + ((size_t)shape * 2 + 1) * slot_bytes;
      float va[4], vn[4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int tx = t.x + i, ty = t.y + j;
          int cv = 0;
          if (tx >= F.x0 && tx <= F.x1 && ty >= F.y0 && ty <= F.y1) cv = c[(size_t)ty * W + tx];
          va[2 * j + i] = (float)aa_byte(cv);
          vn[2 * j + i] = cv >= 128 ? 255.f : 0.f;
        }
      aa = lerp_u8(t, va[0], va[1], va[2], va[3]);
      na = lerp_u8(t, vn[0], vn[1], vn[2], vn[3]);
    };
    int odef = 0;  // mode 9: the object's warp slot + 1 (0: rigid)

    uint32_t m0w, m1w, n0w;  // blending masks of the two frames and the thresholded frame-0 mask, byte p = pixel p
    if (!(sh & kShapeComposite)) {
      if constexpr (kDeform) odef = __builtin_amdgcn_readlane((int)recw, 31);
      // the box touches the block but the outline covers none of this strip's pixels: nothing to mask, sample or blend
      // (mode 9 re-samples a deformed object's frame-1 mask from elsewhere: no such shortcut)
      if (odef <= 0 && __ballot((c0w | c1w) != 0u) == 0ull) continue;
      m0w = 0; m1w = 0; n0w = 0;
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        const int c0 = (int)((c0w >> (8 * p)) & 255), c1 = (int)((c1w >> (8 * p)) & 255);
        const int na0 = c0 >= 128 ? 255 : 0;
        n0w |= (uint32_t)na0 << (8 * p);
        m0w |= (uint32_t)(use_aa ? aa_byte(c0) : na0) << (8 * p);
        m1w |= (uint32_t)(use_aa ? aa_byte(c1) : (c1 >= 128 ? 255 : 0)) << (8 * p);
      }
      if constexpr (kDeform) {
        if (odef > 0) {
          const int shape = smp.first_shape + (int)sh;
          const DevShapeFrame F1 = frames[shape * 2 + 1];  // (once per visit, not once per pixel)
          const DevCropRef C = crops[odef - 1];
          m1w = 0;
#pragma unroll 1
          for (int p = 0; p < kPx; ++p) {
            int aa, na;
            warped_mask1(shape, F1, C, p, aa, na);
            m1w |= (uint32_t)(use_aa ? aa : na) << (8 * p);
          }
        }
      }
    } else {
      // composite: sequential fp32 add / subtract over the components (DG:591-646)
      const DevObjectHdr h = *reinterpret_cast<const DevObjectHdr*>(&objs[oi].tex_base);
      if constexpr (kDeform) odef = h.deform;
      int ua0[kPx], ua1[kPx], un1[kPx], na0[kPx];
#pragma unroll
      for (int p = 0; p < kPx; ++p) { ua0[p] = ua1[p] = un1[p] = 0; na0[p] = 0; }
      for (int k = 0; k < h.n_shapes; ++k) {
        const uint8_t* c = cov + (size_t)(h.first_shape + k) * 2 * slot_bytes;
        uint32_t k0w = 0, k1w = 0;
        const DevShapeFrame F0 = frames[(h.first_shape + k) * 2], F1 = frames[(h.first_shape + k) * 2 + 1];
        if (inside) {
          // a component's coverage exists only in the 64 x 8 blocks its own box touches
          const int by0c = ty0 + (sub >> 1) * kBandRows;
          int d1 = 0;
          if constexpr (kDeform) { if (odef > 0) d1 = (int)ceilf(__uint_as_float(*crops[odef - 1].max_bits)) + 2; }
          const bool v0 = F0.x0 <= F0.x1 && F0.x0 <= tx0 + kTileW - 1 && F0.x1 >= tx0 && F0.y0 <= by0c + kBandRows - 1 && F0.y1 >= by0c;
          const bool v1 = F1.x0 <= F1.x1 && F1.x0 - d1 <= tx0 + kTileW - 1 && F1.x1 + d1 >= tx0 && F1.y0 - d1 <= by0c + kBandRows - 1 && F1.y1 + d1 >= by0c;
          if (has0 && v0) k0w = *reinterpret_cast<const uint32_t*>(c + pix);
          if (has1 && v1) k1w = *reinterpret_cast<const uint32_t*>(c + slot_bytes + pix);
        }
        const bool additive = (h.additive >> k) & 1u;
#pragma unroll
        for (int p = 0; p < kPx; ++p) {
          const int c0 = (int)((k0w >> (8 * p)) & 255), c1 = (int)((k1w >> (8 * p)) & 255);
          const int va0 = aa_byte(c0);
          int va1 = aa_byte(c1);
          const int vn0 = c0 >= 128 ? 255 : 0;
          int vn1 = c1 >= 128 ? 255 : 0;
          if constexpr (kDeform) {
            if (odef > 0) warped_mask1(h.first_shape + k, F1, crops[odef - 1], p, va1, vn1);  // components warp individually
          }
          if (additive) {
            ua0[p] = comp_add(ua0[p], va0); ua1[p] = comp_add(ua1[p], va1);
            na0[p] = comp_add(na0[p], vn0); un1[p] = comp_add(un1[p], vn1);
          } else {
            ua0[p] = comp_sub(ua0[p], va0); ua1[p] = comp_sub(ua1[p], va1);
            na0[p] = comp_sub(na0[p], vn0); un1[p] = comp_sub(un1[p], vn1);
          }
        }
      }
      m0w = 0; m1w = 0; n0w = 0;
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        m0w |= (uint32_t)(use_aa ? ua0[p] : na0[p]) << (8 * p);
        m1w |= (uint32_t)(use_aa ? ua1[p] : un1[p]) << (8 * p);
        n0w |= (uint32_t)na0[p] << (8 * p);
      }
      if (__ballot((m0w | m1w | n0w) != 0u) == 0ull) continue;
      recw = reinterpret_cast<const uint32_t*>(&objs[oi])[min(lane, kRecLast)];
    }

    // the object's matrices: dword i of the record sits in lane i
    auto rec_double = [&](int i) { return __hiloint2double(__builtin_amdgcn_readlane((int)recw, 2 * i + 1), __builtin_amdgcn_readlane((int)recw, 2 * i)); };
    // origin of the W x H centre crop: dwords 24, 25 of the record
    const uint32_t* tex = pool + (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)recw, 25) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)recw, 24));
    // frame 0 texture: identity warp == the crop itself (DG:339-340); issued with the frame-1 taps: one round trip
    uint4 q0 = make_uint4(0, 0, 0, 0);
    if (m0w) q0 = *reinterpret_cast<const uint4*>(tex + (uint32_t)(yv * g.pitch + xv));
    if (__ballot(m1w != 0u)) {
      Mat ti;
      ti.sx = rec_double(6); ti.shy = rec_double(7); ti.shx = rec_double(8); ti.sy = rec_double(9); ti.tx = rec_double(10); ti.ty = rec_double(11);
      uint32_t t1[kPx];
      if (kDeform && odef > 0) {  // applyWarpFieldToTexture(getTransformedTexture(tex, motion), iwarp) (DG:341-345)
        const DevCropRef C = crops[odef - 1];
#pragma unroll
        for (int p = 0; p < kPx; ++p) t1[p] = 0;
#pragma unroll 1
        for (int p = 0; p < kPx; ++p) {
          const bool need = ((m1w >> (8 * p)) & 255u) != 0u;
          if (__ballot(need) == 0ull) continue;
          const int x = xv + p;
          const float2 iw = need ? crop_pair(C, 1, x, yv) : make_float2(0.f, 0.f);
          const Taps t = make_taps((float)x + iw.x, (float)yv + iw.y);
          const uint32_t o = deform_texel<kPow2>(tex, g, ti, t, need);
          const uint32_t keep = need ? o : 0u;
          t1[0] = p == 0 ? keep : t1[0]; t1[1] = p == 1 ? keep : t1[1]; t1[2] = p == 2 ? keep : t1[2]; t1[3] = p == 3 ? keep : t1[3];
        }
      } else {
        const RowDDA R = make_row<kPow2>(ti, yv, W, g.nshift);
        if constexpr (kPow2) {
          const Taps4 T = taps_issue(tex, g, R, xv, m1w != 0u);
          taps_finish(T, t1);
        } else {
#pragma unroll
          for (int p = 0; p < kPx; ++p) t1[p] = m1w ? sample_bilinear(tex, g, R, xv + p) : 0u;
        }
      }
#pragma unroll
      for (int p = 0; p < kPx; ++p) px1[p] = blend_px(px1[p], t1[p], (m1w >> (8 * p)) & 255u);  // m == 0 leaves the pixel as is
    }
    if (m0w) {
      const uint32_t tt[4] = {q0.x, q0.y, q0.z, q0.w};
#pragma unroll
      for (int p = 0; p < kPx; ++p) px0[p] = blend_px(px0[p], tt[p], (m0w >> (8 * p)) & 255u);
    }
    if (__ballot(n0w != 0u)) {
      // MovingObjectBase::getPointFlow (DG:388-407) for pixels this object now owns
      Mat mo;
      mo.sx = rec_double(0); mo.shy = rec_double(1); mo.shx = rec_double(2); mo.sy = rec_double(3); mo.tx = rec_double(4); mo.ty = rec_double(5);
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        if (((n0w >> (8 * p)) & 255u) == 255u) {
          double ix = (double)(xv + p), iy = (double)yv;
          const float save_x = (float)(xv + p), save_y = (float)yv;
          xform(mo, ix, iy);
          fu[p] = (float)(ix - (double)save_x);
          fv[p] = (float)(iy - (double)save_y);
          if constexpr (kDeform) {
            if (odef > 0 && ix >= 0 && ix < (double)W && iy >= 0 && iy < (double)H) {  // DG:403-406
              const float2 f = linear_neumann2(crops[odef - 1], (float)ix, (float)iy);
              fu[p] += f.x;
              fv[p] += f.y;
            }
          }
        }
      }
    }
  }

  if (!inside) return;
  // u8 -> float planes (DG:1229-1245); streaming 16-byte stores, never re-read.  Every plane is a wave-uniform base
  // (SGPR pair) + one 32-bit byte offset per lane; the offset is made opaque so that no address arithmetic is
  // hoisted above the object loop (it would hold registers there).
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const size_t plane = (size_t)W * H;
  uint32_t ob = ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 4u;
  asm volatile("" : "+v"(ob));
  char* b0 = reinterpret_cast<char*>(img0 + (size_t)s * 3 * plane);
  char* b1 = reinterpret_cast<char*>(img1 + (size_t)s * 3 * plane);
  char* bf = reinterpret_cast<char*>(flow + (size_t)s * 2 * plane);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x4 a = {(float)((px0[0] >> (8 * c)) & 255u), (float)((px0[1] >> (8 * c)) & 255u),
               (float)((px0[2] >> (8 * c)) & 255u), (float)((px0[3] >> (8 * c)) & 255u)};
    f32x4 b = {(float)((px1[0] >> (8 * c)) & 255u), (float)((px1[1] >> (8 * c)) & 255u),
               (float)((px1[2] >> (8 * c)) & 255u), (float)((px1[3] >> (8 * c)) & 255u)};
    __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(b0 + (size_t)c * plane * 4 + ob));
    __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(b1 + (size_t)c * plane * 4 + ob));
  }
  f32x4 u = {fu[0], fu[1], fu[2], fu[3]};
  f32x4 v = {fv[0], fv[1], fv[2], fv[3]};
  __builtin_nontemporal_store(u, reinterpret_cast<f32x4*>(bf + ob));
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(bf + plane * 4 + ob));
}

// Leading scalar parameters are preloaded into SGPRs (no load, no wait before the first record fetch).
__global__ __launch_bounds__(64) void compose_rigid_kernel(
    const DevSample* __restrict__ samples, const unsigned long long* __restrict__ blockmask, const DevObject* __restrict__ objects,
    const uint8_t* __restrict__ cov, int n_strips, int tiles_x, int tiles_y, int W, int H, int use_aa, int bg_pitch, int fg_pitch,
    const uint32_t* __restrict__ pool, const uint32_t* __restrict__ bgpool, float* __restrict__ img0, float* __restrict__ img1,
    float* __restrict__ flow, const DevShapeFrame* __restrict__ frames, int* __restrict__ item_count) {
  compose_rigid<false>(samples, blockmask, objects, cov, n_strips, tiles_x, tiles_y, W, H, use_aa, bg_pitch, fg_pitch, pool, bgpool, img0, img1,
                       flow, frames, item_count);
}
// W a power of two (512, 1024, ...): shift-only interpolators and paired tap loads.
__global__ __launch_bounds__(64) void compose_rigid_pow2_kernel(
    const DevSample* __restrict__ samples, const unsigned long long* __restrict__ blockmask, const DevObject* __restrict__ objects,
    const uint8_t* __restrict__ cov, int n_strips, int tiles_x, int tiles_y, int W, int H, int use_aa, int bg_pitch, int fg_pitch,
    const uint32_t* __restrict__ pool, const uint32_t* __restrict__ bgpool, float* __restrict__ img0, float* __restrict__ img1,
    float* __restrict__ flow, const DevShapeFrame* __restrict__ frames, int* __restrict__ item_count) {
  compose_rigid<true>(samples, blockmask, objects, cov, n_strips, tiles_x, tiles_y, W, H, use_aa, bg_pitch, fg_pitch, pool, bgpool, img0, img1,
                      flow, frames, item_count);
}

// Mode 9: the same body with the deformation paths compiled in (masks, textures and flow of deformed objects and backgrounds
// re-sampled through their warp crops).
__global__ __launch_bounds__(64) void compose_deform_kernel(
    RenderDims dm, const DevSample* __restrict__ samples, const DevObject* __restrict__ objects,
    const unsigned long long* __restrict__ blockmask, const uint8_t* __restrict__ cov, const uint32_t* __restrict__ pool,
    const uint32_t* __restrict__ bgpool, float* __restrict__ img0, float* __restrict__ img1, float* __restrict__ flow,
    const DevShapeFrame* __restrict__ frames, const DevCropRef* __restrict__ crops, int* __restrict__ item_count) {
  compose_rigid<false, true>(samples, blockmask, objects, cov, (int)gridDim.x, dm.tiles_x, dm.tiles_y, dm.W, dm.H, dm.use_aa, dm.bg_pitch, dm.fg_pitch,
                             pool, bgpool, img0, img1, flow, frames, item_count, crops);
}
// Mode 9, W a power of two.
#ifdef OFDG_DEFORM_WAVES
#define OFDG_DEFORM_OCC __attribute__((amdgpu_waves_per_eu(OFDG_DEFORM_WAVES)))
#else
#define OFDG_DEFORM_OCC
#endif
__global__ __launch_bounds__(64) OFDG_DEFORM_OCC void compose_deform_pow2_kernel(
    RenderDims dm, const DevSample* __restrict__ samples, const DevObject* __restrict__ objects,
    const unsigned long long* __restrict__ blockmask, const uint8_t* __restrict__ cov, const uint32_t* __restrict__ pool,
    const uint32_t* __restrict__ bgpool, float* __restrict__ img0, float* __restrict__ img1, float* __restrict__ flow,
    const DevShapeFrame* __restrict__ frames, const DevCropRef* __restrict__ crops, int* __restrict__ item_count) {
  compose_rigid<true, true>(samples, blockmask, objects, cov, (int)gridDim.x, dm.tiles_x, dm.tiles_y, dm.W, dm.H, dm.use_aa, dm.bg_pitch, dm.fg_pitch,
                            pool, bgpool, img0, img1, flow, frames, item_count, crops);
}

// --------------------------------------------------------------------------
// Mode-9 warp fields (reference: src/caffe/WarpFields.cpp = WF).
// A field is 4 planes of S*S floats: flow x, flow y, iflow x, iflow y.
// --------------------------------------------------------------------------
// DisplacementComposer::flow_at / iflow_at sampled on the S x S grid (WF:296-316, 347-354)
__global__ __launch_bounds__(256) void wf_sample_kernel(const DevDisplacer* __restrict__ disp, int n_disp, int S,
                                                        float* __restrict__ field) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * S) return;
  const float x = (float)(i % S), y = (float)(i / S);
  float fu = 0, fv = 0, iu = 0, iv = 0;
  for (int k = 0; k < n_disp; ++k) {
    const DevDisplacer& D = disp[k];
    float u, v, ju, jv;
    if (D.type == 0) {
      u = D.dx; v = D.dy; ju = -D.dx; jv = -D.dy;
    } else {
      const float dx = x - D.cx, dy = y - D.cy;
      if (D.type == 1) {
        u = (D.cos_nomega * dx - D.sin_nomega * dy) - dx;
        v = (D.sin_nomega * dx + D.cos_nomega * dy) - dy;
        ju = (D.cos_omega * dx - D.sin_omega * dy) - dx;
        jv = (D.sin_omega * dx + D.cos_omega * dy) - dy;
      } else {
        u = D.factor * dx - dx; v = D.factor * dy - dy;
        ju = D.ifactor * dx - dx; jv = D.ifactor * dy - dy;
      }
    }
    // Gaussian2D::at (WF:101-112)
    const float rx = D.a * (x - D.scx) + D.b * (y - D.scy);
    const float ry = (D.c * (x - D.scx) + D.d * (y - D.scy)) * D.ratio_x_y;
    const float dist_sq = rx * rx + ry * ry;
    // (expf = ofdg_det_expf: the fp64 value rounded once, the same on the device and in the oracle)
    const float w = D.normalizer * (D.gauss_prefactor * ofdg_det_expf(-dist_sq / D.two_sigma_sq));
    fu += u * w; fv += v * w;
    iu += ju * w; iv += jv * w;
  }
  const size_t n = (size_t)S * S;
  field[i] = fu; field[n + i] = fv; field[2 * n + i] = iu; field[3 * n + i] = iv;
}

// One self-composition pass f <- f + f o (id + f) for both the forward pair (planes 0,1)
// and the inverse pair (planes 2,3) (WF:366-384, 406-424); flags pixels leaving the field.
__global__ __launch_bounds__(256) void wf_compose_kernel(const float* __restrict__ from, float* __restrict__ to, int S,
                                                         uint8_t* __restrict__ flagged) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int pair = blockIdx.y;  // 0 forward, 1 inverse
  if (i >= S * S) return;
  const size_t n = (size_t)S * S;
  const float* fx_p = from + (size_t)(2 * pair) * n;
  const float* fy_p = fx_p + n;
  const int x = i % S, y = i / S;
  const float fx = fx_p[i], fy = fy_p[i];
  float ox = fx, oy = fy;
  if ((float)x + fx < 0 || (float)x + fx >= (float)S || (float)y + fy < 0 || (float)y + fy >= (float)S) {
    flagged[(size_t)pair * n + i] = 255;
  } else {
    ox = fx + linear_neumann(fx_p, S, S, (float)x + fx, (float)y + fy);
    oy = fy + linear_neumann(fy_p, S, S, (float)x + fx, (float)y + fy);
  }
  to[(size_t)(2 * pair) * n + i] = ox;
  to[(size_t)(2 * pair + 1) * n + i] = oy;
}

// NaN-out flagged pixels, then clamp_near_zeros (WF:389-398, 425-434, 444-455)
__global__ __launch_bounds__(256) void wf_finish_kernel(float* __restrict__ field, int S, const uint8_t* __restrict__ flagged) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int pair = blockIdx.y;
  if (i >= S * S) return;
  const size_t n = (size_t)S * S;
  float* fx_p = field + (size_t)(2 * pair) * n;
  float* fy_p = fx_p + n;
  const int x = i % S, y = i / S;
  float fx = fx_p[i], fy = fy_p[i];
  bool flag = flagged[(size_t)pair * n + i] != 0;
  if ((float)x + fx < 0 || (float)x + fx >= (float)S || (float)y + fy < 0 || (float)y + fy >= (float)S) flag = true;
  if (flag) {
    fx = __int_as_float(0x7fc00000); fy = fx;
  } else {
    if (fabsf(fx) < 1e-3f) fx = 0.f;
    if (fabsf(fy) < 1e-3f) fy = 0.f;
  }
  fx_p[i] = fx; fy_p[i] = fy;
}

// get_crop(x, y, x+W, y+H) (WF:623-624): copy the four planes of one crop out of a field,
// and reduce max |iflow| (NaNs ignored) for the box dilation of deforming objects.
__global__ __launch_bounds__(256) void wf_crop_kernel(const float* __restrict__ field, int S, int x0, int y0, int cw, int ch,
                                                      float* __restrict__ crop, unsigned* __restrict__ max_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  if (i < cw * ch) {
    const int x = i % cw, y = i / cw;
    const size_t n = (size_t)S * S, cn = (size_t)cw * ch;
    for (int f = 0; f < 4; ++f) {
      const float v = field[f * n + (size_t)(y0 + y) * S + (x0 + x)];
      crop[(size_t)(f >> 1) * 2 * cn + 2 * (size_t)i + (f & 1)] = v;  // interleaved pairs: (flow x, flow y), (iflow x, iflow y)
      if (f >= 2 && v == v) m = fmaxf(m, fabsf(v));
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(max_bits, __float_as_uint(m));  // non-negative floats order as uints
}

// Background warp field: CImg resize(2W, 2H, -100, -100, 3) then *= 2 (DG:1197-1200), linear
// interpolation, X pass then Y pass, each pass rounded to float (CImg 2.x linear resize,
// boundary 0, upscaling branch).  off/foff tables come from the host.
__global__ __launch_bounds__(256) void wf_resize2_kernel(const float* __restrict__ crop, int cw, int ch, int sx, int sy,
                                                         const int* __restrict__ xi, const double* __restrict__ xa,
                                                         const int* __restrict__ yi, const double* __restrict__ ya,
                                                         float* __restrict__ out, unsigned* __restrict__ max_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  if (i < sx * sy) {
    const int x = i % sx, y = i / sx;
    const size_t cn = (size_t)cw * ch, on = (size_t)sx * sy;
    const int x1 = xi[x], x2 = min(x1 + 1, cw - 1);
    const int y1 = yi[y], y2 = min(y1 + 1, ch - 1);
    const double ax = xa[x], ay = ya[y];
    for (int f = 0; f < 4; ++f) {
      const float* p = crop + (size_t)(f >> 1) * 2 * cn + (f & 1);  // (interleaved pairs, as wf_crop_kernel writes them)
      const float r1 = (float)((1 - ax) * (double)p[2 * ((size_t)y1 * cw + x1)] + ax * (double)p[2 * ((size_t)y1 * cw + x2)]);
      const float r2 = (float)((1 - ax) * (double)p[2 * ((size_t)y2 * cw + x1)] + ax * (double)p[2 * ((size_t)y2 * cw + x2)]);
      const float v = (float)((1 - ay) * (double)r1 + ay * (double)r2);
      const float v2 = (float)((double)v * 2.);
      out[(size_t)(f >> 1) * 2 * on + 2 * (size_t)i + (f & 1)] = v2;
      if (f >= 2 && v2 == v2) m = fmaxf(m, fabsf(v2));
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(max_bits, __float_as_uint(m));
}

// --------------------------------------------------------------------------
// texture pool kernels (BGRX u32 texels)
// --------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t lattice(uint32_t seed, uint32_t tex, uint32_t c, uint32_t oct, uint32_t ix, uint32_t iy) {
  uint32_t h = mix32(seed ^ 0x9e3779b9u);
  h = mix32(h ^ (tex * 0x632be5abu + 1u));
  h = mix32(h ^ (c * 0x1b873593u + oct * 0xcc9e2d51u + 7u));
  h = mix32(h ^ (ix * 0x27d4eb2fu));
  h = mix32(h ^ (iy * 0x165667b1u));
  return h & 255u;
}
// Synthetic texture = 3 octaves of integer value noise (cell sizes 64, 16, 4; weights
// 4:2:1), i.e. a smooth field with fine detail so that bilinear filtering and LSB
// errors are visible.  Pure integer arithmetic: reproducible anywhere.
// Background texture preparation (ofdg_params.background_prep = 1): one thread per texel of the
// sample's 2W x 2H texture; see DevBgPrep.  Texture::getRandomizedCrop, DG:87-109 (CImg chain
// restated as one resampling; parity unpinned).  Strict fp32, same operation order as the oracle.
__device__ __forceinline__ float cimg_modf(float x, float m) { return (float)((double)x - (double)m * floor((double)x / (double)m)); }
__device__ __forceinline__ int mirror_index(int i, int n) {
  if ((unsigned)i < (unsigned)n) return i;  // (in range: no division)
  int m = i % (2 * n);
  if (m < 0) m += 2 * n;
  return m < n ? m : 2 * n - m - 1;
}
// R(x, y) of CImg's rotate(angle, 1, 3) for xc = x - rw2, yc = y - rh2, on the SHIFTED pool image (strict fp32, the
// oracle's operation order): mirrored float coordinates, _linear_atXY (Neumann), truncation to u8 per channel
__device__ __forceinline__ uint32_t bgprep_rot_sample(const DevBgPrep& p, float xc, float yc) {
  OFDG_GLOBAL const uint32_t* img = (OFDG_GLOBAL const uint32_t*)p.image_addr;
  const int pw = p.pw, ph = p.ph;
  const float ww = 2.0f * pw, hh = 2.0f * ph;
  float mx = __fadd_rn(__fadd_rn(p.w2, __fmul_rn(xc, p.ca)), __fmul_rn(yc, p.sa));
  float my = __fadd_rn(__fsub_rn(p.h2, __fmul_rn(xc, p.sa)), __fmul_rn(yc, p.ca));
  if (!(mx >= 0.f && mx < ww)) mx = cimg_modf(mx, ww);  // (cimg::mod is the identity on [0, m))
  if (!(my >= 0.f && my < hh)) my = cimg_modf(my, hh);
  mx = mx < (float)pw ? mx : __fsub_rn(__fsub_rn(ww, mx), 1.0f);
  my = my < (float)ph ? my : __fsub_rn(__fsub_rn(hh, my), 1.0f);
  // _linear_atXY (Neumann) on the shifted image
  const float nfx = mx <= 0 ? 0.f : (mx >= (float)(pw - 1) ? (float)(pw - 1) : mx);
  const float nfy = my <= 0 ? 0.f : (my >= (float)(ph - 1) ? (float)(ph - 1) : my);
  const int x = (int)nfx, y = (int)nfy;
  const float dx = __fsub_rn(nfx, (float)x), dy = __fsub_rn(nfy, (float)y);
  const int nx = dx > 0 ? x + 1 : x, ny = dy > 0 ? y + 1 : y;
  // texel (i, j) of the shifted image = pool texel (mirror(i - shx), mirror(j - shy)); for
  // 0 <= shift <= size the mirror of a negative index -k is k - 1
  auto shifted = [](int i, int sh, int n) { const int j = i - sh; return (sh >= 0 && sh <= n) ? (j < 0 ? -j - 1 : j) : mirror_index(j, n); };
  const int xa = shifted(x, p.shx, pw), xb = shifted(nx, p.shx, pw);
  const int ya = shifted(y, p.shy, ph), yb = shifted(ny, p.shy, ph);
  const uint32_t ra = (uint32_t)ya * (uint32_t)pw, rb = (uint32_t)yb * (uint32_t)pw;
  const uint32_t tcc = img[ra + xa], tnc = img[ra + xb], tcn = img[rb + xa], tnn = img[rb + xb];
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sh = 8 * c;
    const float Icc = (float)((tcc >> sh) & 255u), Inc = (float)((tnc >> sh) & 255u);
    const float Icn = (float)((tcn >> sh) & 255u), Inn = (float)((tnn >> sh) & 255u);
    // Icc + dx*(Inc - Icc + dy*(Icc + Inn - Icn - Inc)) + dy*(Icn - Icc)
    const float t = __fsub_rn(__fsub_rn(__fadd_rn(Icc, Inn), Icn), Inc);
    const float val = __fadd_rn(__fadd_rn(Icc, __fmul_rn(dx, __fadd_rn(__fsub_rn(Inc, Icc), __fmul_rn(dy, t)))), __fmul_rn(dy, __fsub_rn(Icn, Icc)));
    out |= (uint32_t)(unsigned char)val << sh;
  }
  return out;
}
// background_prep = 2 (fast form): one prepared texel by ONE resampling along the composed coordinate map
__device__ __forceinline__ uint32_t bgprep_texel(const DevBgPrep& p, int u, int v) {
  const float cxf = fminf((float)(p.cw - 1), __fmul_rn((float)u, p.fx)), cyf = fminf((float)(p.ch - 1), __fmul_rn((float)v, p.fy));
  const float xc = __fsub_rn(__fadd_rn((float)p.x0, cxf), p.rw2), yc = __fsub_rn(__fadd_rn((float)p.y0, cyf), p.rh2);
  return bgprep_rot_sample(p, xc, yc);
}

// ---- background_prep = 1: the CImg chain stage by stage (DG:96-103), four u8 images -------------------------------
//   C = crop(rotate(shift(T)))  cw x ch   bgprep_rotcrop_kernel (shift, rotate and crop are per-texel maps: fused, exact)
//   M = resize of C along x     2W x ch   } bgprep_resize_kernel: every texel of B evaluates the one or two (enlarging) or
//   B = resize of M along y     2W x 2H   } few (moving average) texels of M it needs from C, rounded to u8 as CImg stores them
// Only what compose can read of B (DevBgPrep.r*) is produced, and of C what that needs.
// CImg's enlarging tables (source index and weight of every destination pixel: running double sums, curr = min(n - 1,
// curr + f), sequential by definition) depend on the source and destination lengths only: the host tabulates them once
// for every source length below the destination's (ofdg_api.hip ensure_bgprep_tables): entry [n * S + x].
struct DevResizeTabs {
  const uint16_t* at_x;    // [2W][2W]
  const double* alpha_x;
  const uint16_t* at_y;    // [2H][2H]
  const double* alpha_y;
};
// source range [lo, hi] a resize pass reads for destination pixels d0..d1 (n source, s destination pixels)
__device__ __forceinline__ void cimg_resize_range(int n, int s, int d0, int d1, const uint16_t* __restrict__ at, int* lo, int* hi) {
  if (s > n) { *lo = at[(size_t)n * s + d0]; *hi = min((int)at[(size_t)n * s + d1] + 1, n - 1); }
  else if (s == n) { *lo = d0; *hi = d1; }
  else { *lo = (d0 * n) / s; *hi = ((d1 + 1) * n - 1) / s; }
}
// the region of C (columns cx0..cx1, rows cy0..cy1) the sample's visible part of B needs; false: the crop does not fit the workspace
__device__ __forceinline__ bool bgprep_region(const DevBgPrep& p, const DevResizeTabs& T, int TW, int TH, int cap_cw, int cap_ch,
                                              int* cx0, int* cx1, int* cy0, int* cy1) {
  if (!(p.cw >= 1 && p.ch >= 1 && p.cw <= cap_cw && p.ch <= cap_ch)) return false;
  cimg_resize_range(p.ch, TH, p.ry0, p.ry1, T.at_y, cy0, cy1);
  cimg_resize_range(p.cw, TW, p.rx0, p.rx1, T.at_x, cx0, cx1);
  return true;
}
// Two horizontally adjacent texels of C at once: R(x, y) of bgprep_rot_sample for (xc0, yc) and (xc1, yc), the same
// strict-fp32 operations per texel, evaluated on float pairs (v_pk_mul_f32 / v_pk_add_f32: one instruction for both), and
// - when no lane of the wave leaves the image (no mirroring, no clamping: the usual case, the rotation is a few degrees) -
// without the wrap logic.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct RotPair { f32x2 mx, my; };
// mx = (w2 + xc * ca) + yc * sa;  my = (h2 - xc * sa) + yc * ca   (element-wise: exactly the scalar sequence)
__device__ __forceinline__ RotPair bgprep_rot_coords(const DevBgPrep& p, float xc0, float xc1, float yc) {
  const f32x2 xc = {xc0, xc1};
  RotPair r;
  // (w2, h2 as opaque scalars: read straight into the vector initialisers the two loads become vector loads of the record,
  //  which keep a copy of the record's floats in private memory - scratch traffic in the middle of the callers' loops)
  float w2 = p.w2, h2 = p.h2;
  asm("" : "+v"(w2), "+v"(h2));
  r.mx = (f32x2{w2, w2} + xc * f32x2{p.ca, p.ca}) + f32x2{__fmul_rn(yc, p.sa), __fmul_rn(yc, p.sa)};
  r.my = (f32x2{h2, h2} - xc * f32x2{p.sa, p.sa}) + f32x2{__fmul_rn(yc, p.ca), __fmul_rn(yc, p.ca)};
  return r;
}
__device__ __forceinline__ bool bgprep_shift_plain(const DevBgPrep& p) { return p.shx >= 0 && p.shx <= p.pw && p.shy >= 0 && p.shy <= p.ph; }
// Both texels lie inside [0, pw - 1) x [0, ph - 1) of the rotated image's source coordinates (mod, mirror and the Neumann
// clamp are identities, x + 1 and y + 1 exist) and the shift is plain (0 <= shift <= size).
// the strict-fp32 bilinear form of bgprep_rot_sample for two texels at once (taps cc / nc / cn / nn of texel 0 and 1, their
// fractions as float pairs): Icc + dx*(Inc - Icc + dy*(Icc + Inn - Icn - Inc)) + dy*(Icn - Icc), truncated to u8 per channel
__device__ __forceinline__ uint2 rot_blend2(uint32_t cc0, uint32_t nc0, uint32_t cn0, uint32_t nn0, uint32_t cc1, uint32_t nc1, uint32_t cn1, uint32_t nn1,
                                            f32x2 dx, f32x2 dy) {
  uint2 out = make_uint2(0, 0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int b = 8 * c;
    f32x2 Icc = {(float)((cc0 >> b) & 255u), (float)((cc1 >> b) & 255u)}, Inc = {(float)((nc0 >> b) & 255u), (float)((nc1 >> b) & 255u)};
    f32x2 Icn = {(float)((cn0 >> b) & 255u), (float)((cn1 >> b) & 255u)}, Inn = {(float)((nn0 >> b) & 255u), (float)((nn1 >> b) & 255u)};
    // (opaque: differences of converted bytes are exact, and where two taps are visibly halves of ONE load the compiler subtracts
    //  the bytes as integers and converts the differences - 30 more instructions per texel pair, none of them packed)
    asm("" : "+v"(Icc), "+v"(Inc), "+v"(Icn), "+v"(Inn));
    const f32x2 t = ((Icc + Inn) - Icn) - Inc;
    const f32x2 val = (Icc + dx * ((Inc - Icc) + dy * t)) + dy * (Icn - Icc);
    out.x |= (uint32_t)(unsigned char)val.x << b;
    out.y |= (uint32_t)(unsigned char)val.y << b;
  }
  return out;
}
template <class PairFn>
__device__ __forceinline__ uint2 bgprep_rot_inside2_t(const DevBgPrep& p, const RotPair& r, bool second, PairFn pair) {
  const f32x2 mx = r.mx, my = r.my;
  const int x0i = (int)mx.x, y0i = (int)my.x, x1i = (int)mx.y, y1i = (int)my.y;
  const f32x2 dx = mx - f32x2{(float)x0i, (float)x1i}, dy = my - f32x2{(float)y0i, (float)y1i};
  // The neighbours are ALWAYS x + 1 and y + 1 here: where dx (dy) is 0 the scalar form reads the texel itself instead, but
  // the neighbour's weight is then an exact zero and the value the same.  Texel (i, j) of the shifted image = pool texel
  // (mirror(i - shx), mirror(j - shy)), for 0 <= shift <= size -k -> k - 1: x and x + 1 are neighbours in the pool too
  // (ascending, descending left of the mirror line, or twice texel 0 across it), so ONE 8-byte load per row fetches both.
  // pair(row, col): pool texels (col, row) and (col + 1, row).
  auto sh = [](int i, int s_) { const int j = i - s_; return j < 0 ? -j - 1 : j; };
  auto row_pair = [&](int xi, int ya, int yb, uint32_t* cc, uint32_t* nc, uint32_t* cn, uint32_t* nn) {
    const int j = xi - p.shx;
    const int base = j >= 0 ? j : max(-j - 2, 0);
    const uint2 r0 = pair(ya, base), r1 = pair(yb, base);
    const bool rev = j < 0, swap = j < -1;
    *cc = swap ? r0.y : r0.x; *nc = rev ? r0.x : r0.y;
    *cn = swap ? r1.y : r1.x; *nn = rev ? r1.x : r1.y;
  };
  uint32_t cc0, nc0, cn0, nn0, cc1 = 0, nc1 = 0, cn1 = 0, nn1 = 0;
  row_pair(x0i, sh(y0i, p.shy), sh(y0i + 1, p.shy), &cc0, &nc0, &cn0, &nn0);
  if (second) row_pair(x1i, sh(y1i, p.shy), sh(y1i + 1, p.shy), &cc1, &nc1, &cn1, &nn1);
  return rot_blend2(cc0, nc0, cn0, nn0, cc1, nc1, cn1, nn1, dx, dy);
}
// ... with the taps fetched from the pool image in memory
__device__ __forceinline__ uint2 bgprep_rot_inside2(const DevBgPrep& p, const RotPair& r, bool second) {
  const char* imgc = (const char*)p.image_addr;
  const uint32_t upw = (uint32_t)p.pw;
  return bgprep_rot_inside2_t(p, r, second, [&](int row, int col) { return gload2(imgc, (__umul24((uint32_t)row, upw) + (uint32_t)col) * 4u); });
}
__device__ __forceinline__ uint2 bgprep_rot_sample2(const DevBgPrep& p, float xc0, float xc1, float yc, bool second) {
  const int pw = p.pw, ph = p.ph;
  const RotPair r = bgprep_rot_coords(p, xc0, xc1, yc);
  const bool in0 = r.mx.x >= 0.f && r.mx.x < (float)(pw - 1) && r.my.x >= 0.f && r.my.x < (float)(ph - 1);
  const bool in1 = !second || (r.mx.y >= 0.f && r.mx.y < (float)(pw - 1) && r.my.y >= 0.f && r.my.y < (float)(ph - 1));
  if (__ballot(!(in0 && in1)) != 0ull || !bgprep_shift_plain(p))
    return make_uint2(bgprep_rot_sample(p, xc0, yc), second ? bgprep_rot_sample(p, xc1, yc) : 0u);
  return bgprep_rot_inside2(p, r, second);
}
// C(i, j) = R(mirror(x0 + i), mirror(y0 + j)), R = rotate(shift(T)); sample blockIdx.y; a thread takes texel PAIRS
__global__ __launch_bounds__(256) void bgprep_rotcrop_kernel(const DevBgPrep* __restrict__ prep, DevResizeTabs T, int W, int H,
                                                             int cap_cw, int cap_ch, uint32_t* __restrict__ C, uint32_t* __restrict__ err) {
  const int s = blockIdx.y;
  const DevBgPrep p = prep[s];
  int cx0, cx1, cy0, cy1;
  if (!bgprep_region(p, T, 2 * W, 2 * H, cap_cw, cap_ch, &cx0, &cx1, &cy0, &cy1)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err, kErrBgPrepCapacity);
    return;
  }
  const int rw_ = cx1 - cx0 + 1, rh_ = cy1 - cy0 + 1;
  const int pairs = (rw_ + 1) / 2;  // texel pairs per row of the region
  uint32_t* Cs = C + (size_t)s * cap_cw * cap_ch;
  const float inv_pairs = 1.0f / (float)pairs;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < pairs * rh_; k += gridDim.x * blockDim.x) {  // (the region's size is only known on the device)
    int jj = (int)((float)k * inv_pairs);  // k / pairs (k < 2^22: the float quotient is off by one at most)
    int pi = k - jj * pairs;
    if (pi < 0) { --jj; pi += pairs; } else if (pi >= pairs) { ++jj; pi -= pairs; }
    const int j = cy0 + jj, i = cx0 + 2 * pi;
    const bool second = i + 1 <= cx1;
    const int rx0 = mirror_index(p.x0 + i, p.rw), rx1 = mirror_index(p.x0 + i + 1, p.rw), ry = mirror_index(p.y0 + j, p.rh);
    const float xc0 = __fsub_rn((float)rx0, p.rw2), xc1 = __fsub_rn((float)rx1, p.rw2), yc = __fsub_rn((float)ry, p.rh2);
    const uint2 v = bgprep_rot_sample2(p, xc0, xc1, yc, second);
    const uint32_t o = (uint32_t)(j * p.cw + i);
    Cs[o] = v.x;
    if (second) Cs[o + 1] = v.y;
  }
}
// a / d correctly rounded for operands that need none of the IEEE division's range handling (here 0 <= a < 2^31,
// 1 <= d < 2^16): the compiler's own expansion of a / d - v_rcp_f32, two fma on the reciprocal, a product and four fma on
// the quotient - without v_div_scale / v_div_fixup, and with the reciprocal's part computed once per divisor.
struct UniformDivisor { float d, r; };
__device__ __forceinline__ UniformDivisor make_divisor(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  return {d, __fmaf_rn(__fmaf_rn(-d, r0, 1.f), r0, r0)};
}
__device__ __forceinline__ float div_rn(float a, const UniformDivisor& u) {
  float q = __fmul_rn(a, u.r);
  q = __fmaf_rn(__fmaf_rn(-u.d, q, a), u.r, q);
  return __fmaf_rn(__fmaf_rn(-u.d, q, a), u.r, q);
}
// one axis of CImg's linear get_resize on BGRX texels: destination pixel k of a line whose source texels are texel(j),
// j < n; sdim = destination length.  Enlarging: (T)((1 - a) * v1 + a * v2) in double; shrinking: moving average over the
// n * sdim grid in float, / n, truncated; same length: copy.
// (double)u for u < 2^32 without v_cvt_f64_u32 (9 issue cycles against 5 for an addition, tools/microbench/f64_rates.hip):
// the bits of 2^52 + u, minus 2^52 - exact
__device__ __forceinline__ double u32_to_double(uint32_t u) { return __hiloint2double(0x43300000, (int)u) - 4503599627370496.0; }
// `at` / `alpha`: the enlarging tables' rows of THIS source length (the caller adds n * sdim once: uniform per sample)
template <class Texel>
__device__ __forceinline__ uint32_t cimg_resize_texel(int n, int sdim, int k, const uint16_t* __restrict__ at, const double* __restrict__ alpha,
                                                      Texel texel) {
  uint32_t out = 0;
  if (sdim == n) {
    out = texel(k);
  } else if (sdim > n) {
    const int a0 = at[(uint32_t)k];
    const double al = alpha[(uint32_t)k], al1 = 1 - al;
    const uint32_t t1 = texel(a0), t2 = a0 < n - 1 ? texel(a0 + 1) : t1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double v1 = u32_to_double((t1 >> (8 * c)) & 255u), v2 = u32_to_double((t2 >> (8 * c)) & 255u);
      out |= (uint32_t)(unsigned char)(al1 * v1 + al * v2) << (8 * c);
    }
  } else {
    float acc[3] = {0.f, 0.f, 0.f};
    const int lo = k * n, hi = lo + n;  // (< 2^31: both lengths are a few thousand at most)
    for (int j = lo / sdim; j * sdim < hi; ++j) {
      const int a = j * sdim, b = a + sdim;
      const float d = (float)((b < hi ? b : hi) - (a > lo ? a : lo));
      const uint32_t t = texel(j);
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn((float)((t >> (8 * c)) & 255u), d));
    }
    const UniformDivisor dn = make_divisor((float)n);
#pragma unroll
    for (int c = 0; c < 3; ++c) out |= (uint32_t)(unsigned char)div_rn(acc[c], dn) << (8 * c);
  }
  return out;
}
// B(x, y) = Y-resize of M(x, .), M(x, j) = X-resize of C(., j); M never touches memory.  The kernel is bound by its
// arithmetic (≈ 100 VALU instructions per texel, 12 M texels per batch), so a thread renders kResizeRun consecutive rows
// of one column and evaluates every row of M they share ONCE (one more than the run when enlarging, at most 4/3 of it
// + 1 when shrinking), parked in the thread's own LDS column - an indexable scratch: registers cannot be indexed by a
// run-time row.  (Issuing a run's loads together, or four texels per thread in bgprep_rotcrop_kernel, made both
// kernels slower: profiles/r02_ab_background_prep_kernels.txt.)
constexpr int kResizeRun = 4, kResizeMaxM = 8;
__global__ __launch_bounds__(256) void bgprep_resize_kernel(const DevBgPrep* __restrict__ prep, DevResizeTabs T, int W, int H, int cap_cw,
                                                            int cap_ch, const uint32_t* __restrict__ C, uint32_t* __restrict__ B) {
  __shared__ uint32_t s_m[kResizeMaxM][256];
  const int s = blockIdx.y, TW = 2 * W, TH = 2 * H, tid = threadIdx.x;
  const DevBgPrep p = prep[s];
  if (!(p.cw >= 1 && p.ch >= 1 && p.cw <= cap_cw && p.ch <= cap_ch)) return;
  const int rw_ = p.rx1 - p.rx0 + 1, rh_ = p.ry1 - p.ry0 + 1;
  const int runs = (rh_ + kResizeRun - 1) / kResizeRun;
  const uint32_t* Cs = C + (size_t)s * cap_cw * cap_ch;
  uint32_t* Bs = B + (size_t)s * TW * TH;
  // the enlarging tables' rows of this sample's source lengths (entry [n * S + x]: uniform bases, 32-bit indices below)
  const uint16_t* at_x = T.at_x + (size_t)p.cw * TW;
  const double* alpha_x = T.alpha_x + (size_t)p.cw * TW;
  const uint16_t* at_y = T.at_y + (size_t)p.ch * TH;
  const double* alpha_y = T.alpha_y + (size_t)p.ch * TH;
  const float inv_rw = 1.0f / (float)rw_;
  for (int k = blockIdx.x * blockDim.x + tid; k < runs * rw_; k += gridDim.x * blockDim.x) {
    int rr = (int)((float)k * inv_rw);  // k / rw_ (k < 2^22: the float quotient is off by one at most)
    int xo = k - rr * rw_;
    if (xo < 0) { --rr; xo += rw_; } else if (xo >= rw_) { ++rr; xo -= rw_; }
    const int x = p.rx0 + xo, y0 = p.ry0 + rr * kResizeRun, y1 = min(y0 + kResizeRun - 1, p.ry1);
    auto c_row = [&](int j) {  // M(x, j)
      return cimg_resize_texel(p.cw, TW, x, at_x, alpha_x, [&](int i) { return Cs[(uint32_t)(j * p.cw + i)]; });
    };
    int jlo, jhi;
    cimg_resize_range(p.ch, TH, y0, y1, T.at_y, &jlo, &jhi);
    if (jhi - jlo < kResizeMaxM) {
#pragma unroll
      for (int m = 0; m < kResizeMaxM; ++m)
        if (jlo + m <= jhi) s_m[m][tid] = c_row(jlo + m);
#pragma unroll
      for (int r = 0; r < kResizeRun; ++r)
        if (y0 + r <= y1) Bs[(uint32_t)((y0 + r) * TW + x)] = cimg_resize_texel(p.ch, TH, y0 + r, at_y, alpha_y, [&](int j) { return s_m[j - jlo][tid]; });
    } else {  // (a crop beyond 4/3 of the texture is refused before it gets here; kept for safety)
      for (int y = y0; y <= y1; ++y) Bs[(uint32_t)(y * TW + x)] = cimg_resize_texel(p.ch, TH, y, at_y, alpha_y, c_row);
    }
  }
}

// ---- the same chain in ONE launch (pool images at least 2W x 2H: a crop is at most 4/3 of the texture) ------------------
// bgprep_stream_kernel: ONE WAVE renders a 64 x kPrepH tile of B = the sample's 2W x 2H texture by walking the tile's rows of
// C = crop(rotate(shift(T))) once, top to bottom, kPrepG rows at a time:
//   rotation   the group's texels of C are sampled into kPrepG LDS rows (texel pairs, two rounds of gathers in flight);
//   X resize   lane = column x of the tile computes M(x, j) from LDS row j into a REGISTER; the last three rows of M stay
//              (m0, m1, m2);
//   Y resize   source rows are monotone in y, so every row y of B whose last source row is j is complete as soon as M(., j)
//              exists: it is computed from the register window and stored.
// C lives in 8.4 KB of LDS per wave, M never exists outside registers, B goes to memory once; no barrier between waves, one
// pass over the tile, and a tile may be as tall as one likes (its rows of C are streamed; a taller tile recomputes fewer seam
// rows: 2 of kPrepH / zoom + 2).  The tiles of all samples are numbered consecutively (their count per sample is only known
// on the device) and handed out grid-stride to kPrepGrid single-wave workgroups (ofdg_api.hip; a bench batch has fewer tiles
// than that: one tile per wave).
// Round 5 replaced the round-4 form with it (two-wave workgroups, a 64 x 16 tile's whole piece of C in 9 KB of LDS through
// three barrier-separated passes; tools/patches/r05_bgprep_experiment_switches.patch): +3 - 4 % on the headline step in same-box
// A/Bs (profiles/r05_experiments_log.md section 2), no __syncthreads between waves, no cap on a tile's rows.  What decides its
// speed is the LENGTH OF A WAVE'S DEPENDENT CHAIN per tile - rounds of gathers, then LDS, then stores, and a gather behind a
// store waits for that store too (vmcnt counts loads and stores in issue order) - so rows are sampled in groups of 24: most
// tiles are two groups (groups of 4 / 8: -6 % / -3 %; 16 / 20: the same step with compose's launch 10 us longer; 28: -4 %;
// 44 rows = 16 KB of LDS: -14 %), the next round's gathers are requested before this round is blended (three rounds in
// flight: 98 registers, -4 %), and a tile is 32 rows (24: faster alone, slower in the step; 16 / 64: -10 %).
constexpr int kPrepW = 64;                      // columns of B per tile: lane = column
constexpr int kPrepH = 32;                      // rows of B per tile (<= 64: lane r holds row r's resize entry)
constexpr int kPrepG = 24;                      // rows of C sampled per group
constexpr int kPrepRun = 16;                    // consecutive tiles that share an XCD (the kernel's blockIdx -> tile mapping)
constexpr int kPrepCW = 90;                     // columns of C a tile needs at most: 64 * 4/3 + 2, even (texel pairs), + the margin of the crop size
#ifndef OFDG_PREP_DIRS
#define OFDG_PREP_DIRS 1
#endif
#ifndef OFDG_PREP_FAST_RESIZE
#define OFDG_PREP_FAST_RESIZE 1
#endif
constexpr bool kPrepFastResize = OFDG_PREP_FAST_RESIZE != 0;  // the resize passes specialised by a tile's case (both axes enlarge / shrink)
#ifndef OFDG_PREP_ROUND2
#define OFDG_PREP_ROUND2 1
#endif
constexpr bool kPrepDirs = OFDG_PREP_DIRS != 0;  // the rotation specialised by a tile's side of the shift's mirror lines
constexpr int kPrepMaxSamples = 512;            // (the counter sampler's batch limit; ofdg_api.hip falls back to the two-kernel form beyond)
// A tile's placement costs small dependent loads - which sample holds tile t (prefix of the samples' tile counts: LDS),
// that sample's record, the four resize-table entries that bound the tile's piece of C - so a workgroup that has another
// tile to do requests those for the NEXT tile (scalar loads) at the top of this tile's turn.
struct PrepTile {
  int s;                    // sample; n_samples: no tile left
  int bx0, bx1, by0, by1;   // texels of B the tile renders
  int cx0, cx1, cy0, cy1;   // texels of C it needs
  int fits;                 // ... which fit the LDS tile of C
  int inside;               // no mirroring / clamping anywhere in the tile: the per-texel range tests are skipped
  int xdir, ydir;           // ... and where the tile lies relative to the shift's mirror lines: 1 = every tap right of / below the line
                            // (pool index = i - shift), 2 = every tap left of / above it (pool index = shift - 1 - i), 0 = per lane
};
__host__ __device__ __forceinline__ bool prep_sample_fits(const DevBgPrep& q, int cap_cw, int cap_ch) { return q.cw >= 1 && q.ch >= 1 && q.cw <= cap_cw && q.ch <= cap_ch; }  // (caps <= 4/3 of the texture + 2)
__host__ __device__ __forceinline__ int prep_tile_cols(const DevBgPrep& q) { return (q.rx1 - q.rx0 + kPrepW) / kPrepW; }
__host__ __device__ __forceinline__ int prep_tile_rows(const DevBgPrep& q) { return (q.ry1 - q.ry0 + kPrepH) / kPrepH; }
// cimg_resize_range with the two table entries it reads already in hand (e0 = at[n * s + d0], e1 = at[n * s + d1])
__device__ __forceinline__ void prep_range(int n, int s, int d0, int d1, int e0, int e1, int* lo, int* hi) {
  if (s > n) { *lo = e0; *hi = min(e1 + 1, n - 1); }
  else if (s == n) { *lo = d0; *hi = d1; }
  else { *lo = (d0 * n) / s; *hi = ((d1 + 1) * n - 1) / s; }
}
// the tile's piece of C and whether the whole of it maps inside the source image (S3)
__device__ __forceinline__ void prep_finish_tile(const DevBgPrep& p, PrepTile& F, int TW, int TH, int ex0, int ex1, int ey0, int ey1) {
  prep_range(p.cw, TW, F.bx0, F.bx1, ex0, ex1, &F.cx0, &F.cx1);
  prep_range(p.ch, TH, F.by0, F.by1, ey0, ey1, &F.cy0, &F.cy1);
  const int cx0 = F.cx0, cx1 = F.cx1, cy0 = F.cy0, cy1 = F.cy1;
  F.fits = cx1 - cx0 + 1 <= kPrepCW ? 1 : 0;  // (rows are streamed: any number of them)
  // The usual tile: its crop coordinates need no mirroring and its four corners - so, the map being affine, all its texels
  // (a margin of one texel covers the rounding of the per-texel evaluation) - lie inside the source image: no per-texel tests.
  bool inside = F.fits && bgprep_shift_plain(p) && p.x0 + cx0 >= 0 && p.x0 + cx1 + 1 < p.rw && p.y0 + cy0 >= 0 && p.y0 + cy1 < p.rh;
  int xdir = 0, ydir = 0;
  if (inside) {
    const float xa = __fsub_rn((float)(p.x0 + cx0), p.rw2), xb = __fsub_rn((float)(p.x0 + cx1 + 1), p.rw2);
    const float ya = __fsub_rn((float)(p.y0 + cy0), p.rh2), yb = __fsub_rn((float)(p.y0 + cy1), p.rh2);
    const RotPair ra = bgprep_rot_coords(p, xa, xb, ya), rb = bgprep_rot_coords(p, xa, xb, yb);
    const float lo_x = fminf(fminf(ra.mx.x, ra.mx.y), fminf(rb.mx.x, rb.mx.y)), hi_x = fmaxf(fmaxf(ra.mx.x, ra.mx.y), fmaxf(rb.mx.x, rb.mx.y));
    const float lo_y = fminf(fminf(ra.my.x, ra.my.y), fminf(rb.my.x, rb.my.y)), hi_y = fmaxf(fmaxf(ra.my.x, ra.my.y), fmaxf(rb.my.x, rb.my.y));
    inside = lo_x >= 1.f && hi_x < (float)(p.pw - 2) && lo_y >= 1.f && hi_y < (float)(p.ph - 2);
    // a tap's column is floor(mx) (and + 1), mx within [lo_x, hi_x] up to the rounding the margin of one texel covers
    xdir = lo_x - 1.f >= (float)p.shx ? 1 : (hi_x + 3.f <= (float)p.shx ? 2 : 0);
    ydir = lo_y - 1.f >= (float)p.shy ? 1 : (hi_y + 3.f <= (float)p.shy ? 2 : 0);
  }
  F.inside = __builtin_amdgcn_readfirstlane(inside ? 1 : 0);  // (wave-uniform by construction)
  F.xdir = __builtin_amdgcn_readfirstlane(inside ? xdir : 0);
  F.ydir = __builtin_amdgcn_readfirstlane(inside ? ydir : 0);
}
// ---- the resize passes in exact INTEGER arithmetic --------------------------------------------------------------------------
// Enlarging: CImg evaluates (T)((1 - a) v1 + a v2) in double.  Where the weight a has at most 45 fractional bits - every
// table entry whose source position is >= 128: the weight is `curr - floor(curr)` of a double `curr`, which then has at most
// 52 - 7 fractional bits - 1 - a is exact, both products (8-bit integer x 45 fractional bits: 53 bits) are exact and so is
// their sum (below 256): the double result IS the real number v1 + a (v2 - v1), and its truncation is
//     v1 + floor(d A / 2^45),  d = v2 - v1,  A = a 2^45 = A1 2^23 + A0:
//     d A = (d A1 + floor(d A0 / 2^23)) 2^23 + r,  0 <= r < 2^23   =>   floor(d A / 2^45) = (d A1 + (d A0 >> 23)) >> 22
// (arithmetic shifts; |d A0| < 2^31, |d A1| < 2^30): two 24-bit multiplies, two shifts, two additions per channel instead of
// two conversions, two fp64 products, an fp64 sum and a conversion.  Entries that are not exact keep the double form.
// Shrinking: CImg's float moving average sum(v_j w_j) / n, truncated.  The sum is an integer below 2^24 (weights are overlap
// lengths, sum n <= 4/3 of 1024..2048): exact in float; the correctly rounded quotient of an integer by n < 2^13 cannot round
// up to the next integer (it is at least 1 / n below it, half an ulp at 255 is 2^-17): the byte is floor(sum / n), in integers
// sum via v_mad_u32_u24 and the division by a multiply-high with M = ceil(2^32 / n) (exact for sum < 2^19 n / ... : the
// error sum e / 2^32 < 2^-13 < 1 / n).
struct FixWeight { int a1, a0; bool exact; };
__device__ __forceinline__ FixWeight fix_weight(double al) {
  const double t = ldexp(al, 45);           // (exact: a power of two)
  const double u = floor(ldexp(t, -23));    // floor(a 2^22)
  FixWeight w;
  w.exact = al >= 0.0 && al < 1.0 && t == floor(t);
  w.a1 = (int)u;
  w.a0 = (int)(t - ldexp(u, 23));           // (exact when w.exact; unused otherwise)
  return w;
}
__device__ __forceinline__ uint32_t enlarge_texel_fix(uint32_t t1, uint32_t t2, int a1, int a0) {
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int v1 = (int)((t1 >> (8 * c)) & 255u), d = (int)((t2 >> (8 * c)) & 255u) - v1;
    const int m = (__mul24(d, a1) + (__mul24(d, a0) >> 23)) >> 22;
    out |= (uint32_t)(v1 + m) << (8 * c);
  }
  return out;
}
// Shrinking by less than half (n <= 2 sdim: what a tile of the one-launch form holds, prep_sample_fits): destination texel k averages the source
// interval [k n, (k + 1) n) in units of 1 / sdim, which touches at most THREE source texels j0, j0 + 1, j0 + 2 with overlap
// lengths d0 > 0, d1, d2 >= 0 (sum n).  The taps depend on (n, sdim, k) only: once per column of a tile in the X pass, once
// per row in the Y pass (scalar) - not per texel.
struct ShrinkTaps { int j0; uint32_t d0, d1, d2; };
__device__ __forceinline__ ShrinkTaps shrink_taps(int n, int sdim, int k) {
  const int lo = k * n, hi = lo + n;  // (< 2^31: both lengths are a few thousand at most)
  ShrinkTaps t;
  t.j0 = lo / sdim;
  const int a = t.j0 * sdim;
  t.d0 = (uint32_t)(min(a + sdim, hi) - lo);
  t.d1 = (uint32_t)min(max(hi - (a + sdim), 0), sdim);
  t.d2 = (uint32_t)min(max(hi - (a + 2 * sdim), 0), sdim);
  return t;
}
// floor(acc / n) for acc <= 255 n as a multiply-high by mdiv = ceil(2^32 / n) = 2^32 / n + e, 0 <= e < 1: the product is
// 2^32 (acc / n + acc e / 2^32), and the excess acc e / 2^32 < 255 n / 2^32 stays below the 1 / n that separates acc / n from the
// next integer as long as 255 n^2 < 2^32: n <= 4103 (crops of frames up to 1536 wide).  With n > 256 both factors are also
// below 2^24: the full-rate 24-bit multiply-high instead of the quarter-rate 32-bit one.  Any other n: 32-bit multiply-high
// (one too large at most) and a correction.
__device__ __forceinline__ bool shrink_div24(int n) { return n > 256 && n <= 4103; }
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// bgprep_rot_inside2 in two halves - the four 8-byte gathers of a texel pair requested, and the pair blended - so that the
// next pair's gathers can be in flight while this one is blended
// kXD / kYD: the tile's side of the shift's mirror lines when it is the same for every tap (PrepTile.xdir / ydir; 0: per lane).
// Right of the line a tap's pool column is i - shx and its neighbour's the next one: the 8-byte load IS (cc, nc); left of it the
// columns descend (shx - 1 - i): the load at the neighbour's column is (nc, cc).  Rows likewise (y + 1 is the row below or above).
struct RotTaps { uint2 a0, a1, b0, b1; f32x2 dx, dy; int ja, jb; };
template <int kXD = 0, int kYD = 0>
__device__ __forceinline__ RotTaps rot_issue(const DevBgPrep& p, const RotPair& r) {
  const char* imgc = (const char*)p.image_addr;
  const uint32_t upw = (uint32_t)p.pw;
  const f32x2 mx = r.mx, my = r.my;
  const int x0i = (int)mx.x, y0i = (int)my.x, x1i = (int)mx.y, y1i = (int)my.y;
  RotTaps t;
  t.dx = mx - f32x2{(float)x0i, (float)x1i};
  t.dy = my - f32x2{(float)y0i, (float)y1i};
  auto pair = [&](int row, int col) { return gload2(imgc, (__umul24((uint32_t)row, upw) + (uint32_t)col) * 4u); };
  t.ja = x0i - p.shx; t.jb = x1i - p.shx;
  int basea, baseb;
  if (kXD == 1) { basea = t.ja; baseb = t.jb; }
  else if (kXD == 2) { basea = -t.ja - 2; baseb = -t.jb - 2; }
  else { basea = t.ja >= 0 ? t.ja : max(-t.ja - 2, 0); baseb = t.jb >= 0 ? t.jb : max(-t.jb - 2, 0); }
  int ra0, ra1, rb0, rb1;  // pool rows of texel a's and b's upper and lower taps
  if (kYD == 1) { ra0 = y0i - p.shy; ra1 = ra0 + 1; rb0 = y1i - p.shy; rb1 = rb0 + 1; }
  else if (kYD == 2) { ra0 = p.shy - 1 - y0i; ra1 = ra0 - 1; rb0 = p.shy - 1 - y1i; rb1 = rb0 - 1; }
  else {
    auto sh = [](int i, int s_) { const int j = i - s_; return j < 0 ? -j - 1 : j; };
    ra0 = sh(y0i, p.shy); ra1 = sh(y0i + 1, p.shy); rb0 = sh(y1i, p.shy); rb1 = sh(y1i + 1, p.shy);
  }
  t.a0 = pair(ra0, basea); t.a1 = pair(ra1, basea);
  t.b0 = pair(rb0, baseb); t.b1 = pair(rb1, baseb);
  return t;
}
template <int kXD = 0>
__device__ __forceinline__ uint2 rot_finish(const RotTaps& t) {
  if (kXD == 1) return rot_blend2(t.a0.x, t.a0.y, t.a1.x, t.a1.y, t.b0.x, t.b0.y, t.b1.x, t.b1.y, t.dx, t.dy);
  if (kXD == 2) return rot_blend2(t.a0.y, t.a0.x, t.a1.y, t.a1.x, t.b0.y, t.b0.x, t.b1.y, t.b1.x, t.dx, t.dy);
  const bool reva = t.ja < 0, swapa = t.ja < -1, revb = t.jb < 0, swapb = t.jb < -1;
  const uint32_t cc0 = swapa ? t.a0.y : t.a0.x, nc0 = reva ? t.a0.x : t.a0.y, cn0 = swapa ? t.a1.y : t.a1.x, nn0 = reva ? t.a1.x : t.a1.y;
  const uint32_t cc1 = swapb ? t.b0.y : t.b0.x, nc1 = revb ? t.b0.x : t.b0.y, cn1 = swapb ? t.b1.y : t.b1.x, nn1 = revb ? t.b1.x : t.b1.y;
  return rot_blend2(cc0, nc0, cn0, nn0, cc1, nc1, cn1, nn1, t.dx, t.dy);
}
// entry [idx] of a resize table of 16-bit entries, for a uniform idx, as ONE scalar load (the tables are constant while the
// kernel runs; there is no scalar load of 16 bits: the dword that holds the entry, hipMalloc'ed base)
__device__ __forceinline__ int tab_entry_uniform(const uint16_t* at, uint32_t idx) {
  const uint32_t w = ((OFDG_CONSTANT const uint32_t*)at)[idx >> 1];
  return (int)((w >> ((idx & 1u) * 16u)) & 0xffffu);
}
// One destination pixel's share of a resize axis (n source, sdim destination pixels) in three registers - the kernel keeps one
// per lane for its column and one per lane for "its" row all through a tile, so they are packed:
//   enlarging   u0 = source index a0, u1 = A1, u2 = A0 | exact << 31   (fix_weight of the table's alpha; not exact: u1 = u2 = 0)
//   same size   u0 = k
//   shrinking   u0 = first source index j0, u1 = d0, u2 = d1           (shrink_taps; d2 = n - d0 - d1)
struct AxisEntry { int u0; uint32_t u1, u2; };
__device__ __forceinline__ AxisEntry axis_entry(int n, int sdim, int k, const uint16_t* __restrict__ at, const double* __restrict__ alpha) {
  AxisEntry e{k, 0u, 0u};
  if (sdim > n) {
    const uint32_t i = (uint32_t)(n * sdim + k);
    e.u0 = at[i];
    const FixWeight w = fix_weight(alpha[i]);
    if (w.exact) { e.u1 = (uint32_t)w.a1; e.u2 = (uint32_t)w.a0 | 0x80000000u; }
  } else if (sdim < n) {
    const ShrinkTaps t = shrink_taps(n, sdim, k);
    e.u0 = t.j0; e.u1 = t.d0; e.u2 = t.d1;
  }
  return e;
}
// the last source index the pixel needs
__device__ __forceinline__ int axis_last(int n, int sdim, int u0, uint32_t u1, uint32_t u2) {
  if (sdim > n) return min(u0 + 1, n - 1);
  if (sdim == n) return u0;
  return u0 + ((uint32_t)n - u1 - u2 > 0u ? 2 : (u2 > 0u ? 1 : 0));
}
// ... and its value from the source texels t0, t1, t2 at u0, u0 + 1, u0 + 2 (clamped by the caller to what exists; enlarging
// reads t0, t1 - t1 = t0 in the last source pixel -, same size t0).  `exact`: u1 / u2 hold the weight (uniform); otherwise
// CImg's double form with the table's alpha[k] loaded here (entries at source positions below 128: rare).
__device__ __forceinline__ uint32_t axis_texel(int n, int sdim, int k, uint32_t u1, uint32_t u2, bool exact, uint32_t t0, uint32_t t1, uint32_t t2,
                                               uint32_t mdiv, bool div24, const double* __restrict__ alpha) {
  if (sdim > n) {
    if (exact) return enlarge_texel_fix(t0, t1, (int)u1, (int)(u2 & 0x7fffffffu));
    const double al = alpha[(uint32_t)(n * sdim + k)], al1 = 1 - al;
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double v1 = u32_to_double((t0 >> (8 * c)) & 255u), v2 = u32_to_double((t1 >> (8 * c)) & 255u);
      out |= (uint32_t)(unsigned char)(al1 * v1 + al * v2) << (8 * c);
    }
    return out;
  }
  if (sdim == n) return t0;
  const uint32_t d0 = u1, d1 = u2, d2 = (uint32_t)n - u1 - u2;
  uint32_t acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    acc[c] = __umul24((t2 >> (8 * c)) & 255u, d2) + (__umul24((t1 >> (8 * c)) & 255u, d1) + __umul24((t0 >> (8 * c)) & 255u, d0));
  if (div24) return mulhi_u24(acc[0], mdiv) | (mulhi_u24(acc[1], mdiv) << 8) | (mulhi_u24(acc[2], mdiv) << 16);
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint32_t q = __umulhi(acc[c], mdiv);
    q -= (q * (uint32_t)n > acc[c]) ? 1u : 0u;
    out |= q << (8 * c);
  }
  return out;
}
// paths (diagnostics, nullptr in production): 9 counters of tiles by the form that rendered them, [rotation + 3 * resize]: rotation
// 0 = general (off the `inside` fast path), 1 = inside, side of the mirror lines decided per lane, 2 = inside and specialised by
// side; resize 0 = decided per row, 1 = both axes enlarge, 2 = both shrink
__global__ __launch_bounds__(64) void bgprep_stream_kernel(const DevBgPrep* __restrict__ prep, DevResizeTabs T, int W, int H, int n_samples,
                                                           int cap_cw, int cap_ch, uint32_t* __restrict__ B, uint32_t* __restrict__ err,
                                                           uint32_t* __restrict__ paths) {
  __shared__ uint32_t s_c[kPrepG][kPrepCW];
  extern __shared__ int s_first[];  // [n_samples + 1]: tiles of the samples before sample i
  const int TW = 2 * W, TH = 2 * H, lane = threadIdx.x;
  {  // the tiles of all samples are numbered consecutively: prefix of their counts, 64 samples at a time
    int running = 0;
    if (lane == 0) s_first[0] = 0;
    for (int c = 0; c < n_samples; c += 64) {
      const int i = c + lane;
      int nt = 0;
      if (i < n_samples) {
        const DevBgPrep& q = prep[i];
        const bool fits = prep_sample_fits(q, cap_cw, cap_ch);
        if (!fits && blockIdx.x == 0) atomicOr(err, kErrBgPrepCapacity);
        nt = fits ? prep_tile_cols(q) * prep_tile_rows(q) : 0;
      }
      const int incl = wave_scan_incl(nt);
      if (i < n_samples) s_first[i + 1] = running + incl;
      running += __shfl(incl, 63, 64);
    }
  }
  __syncthreads();
  const int total = __builtin_amdgcn_readfirstlane(s_first[n_samples]);
  // the sample that holds tile t: the number of samples whose tiles end at or before t (64 samples per ballot)
  auto sample_of = [&](int t) {
    int n = 0;
    for (int c = 0; c < n_samples; c += 64) {
      const int i = c + lane;
      n += __popcll(__ballot(i < n_samples && t >= s_first[i + 1]));
    }
    return __builtin_amdgcn_readfirstlane(n);
  };
  auto tab_index = [](int n, int dim, int k) { return (uint32_t)(n < dim ? n * dim + k : 0); };
  struct PrepBox { int cw, ch, rx0, ry0, rx1, ry1; };
  typedef OFDG_CONSTANT const DevBgPrep ConstPrep;  // (records are constant while this kernel runs: scalar loads)
  auto box_fields = [&](int s) { const ConstPrep& q = ((ConstPrep*)prep)[s]; return PrepBox{q.cw, q.ch, q.rx0, q.ry0, q.rx1, q.ry1}; };
  auto box_of = [&](const PrepBox& q, int t, int s, PrepTile& F) {  // the tile's texels of B
    const int tcols = (q.rx1 - q.rx0 + kPrepW) / kPrepW;
    const int ti = t - __builtin_amdgcn_readfirstlane(s_first[s]), ty = ti / tcols, tx = ti - ty * tcols;
    F.s = s;
    F.bx0 = q.rx0 + tx * kPrepW; F.bx1 = min(F.bx0 + kPrepW - 1, q.rx1);
    F.by0 = q.ry0 + ty * kPrepH; F.by1 = min(F.by0 + kPrepH - 1, q.ry1);
  };
  // Workgroups are dealt to the 8 XCDs round-robin.  Runs of kPrepRun consecutive tiles - neighbours in a sample's tile row and
  // the row below: they share the source lines under their seams - go to ONE XCD, so that what one tile fetched the next finds
  // in that XCD's L2: the kernel's memory-side reads fall from 45.7 to 28.7 MB per batch (FETCH_SIZE; 64-tile runs: 24.8 MB but a
  // worse balance), the step gains 0.6 % (profiles/r05_ab_prep_xcd_runs*.txt).  (The tail of a grid that is no multiple of
  // 8 runs keeps its order.)
  int t = blockIdx.x;
  if (t < (int)(gridDim.x / (8 * kPrepRun)) * (8 * kPrepRun)) {
    const int xcd = t & 7, slot = t >> 3;
    t = ((slot / kPrepRun) * 8 + xcd) * kPrepRun + (slot % kPrepRun);
  }
  if (t >= total) return;
  PrepTile cur;
  int ex0, ex1, ey0, ey1;  // the four table entries that bound the tile's piece of C (requested one tile ahead)
  {
    const int s0 = sample_of(t);
    const PrepBox q = box_fields(s0);
    box_of(q, t, s0, cur);
    ex0 = tab_entry_uniform(T.at_x, tab_index(q.cw, TW, cur.bx0)); ex1 = tab_entry_uniform(T.at_x, tab_index(q.cw, TW, cur.bx1));
    ey0 = tab_entry_uniform(T.at_y, tab_index(q.ch, TH, cur.by0)); ey1 = tab_entry_uniform(T.at_y, tab_index(q.ch, TH, cur.by1));
  }
  for (;;) {
    // ---- this tile's sample record, the placement of the next tile and the entries that bound ITS piece of C: scalar loads,
    // requested here, used at the top of the next turn ----
    const int tn = t + (int)gridDim.x;
    const bool more = tn < total;
    const int sc = __builtin_amdgcn_readfirstlane(cur.s);
    const int sn = more ? sample_of(tn) : sc;
    const DevBgPrep p = ((ConstPrep*)prep)[sc];
    const PrepBox qn = box_fields(sn);
    prep_finish_tile(p, cur, TW, TH, ex0, ex1, ey0, ey1);
    const int bx0 = cur.bx0, bx1 = cur.bx1, by0 = cur.by0, by1 = cur.by1;
    const int cx0 = cur.cx0, cx1 = cur.cx1, cy0 = cur.cy0, cy1 = cur.cy1;
    const int ncw = cx1 - cx0 + 1, nch = cy1 - cy0 + 1;
    // this tile's resize entries: column bx0 + lane (the X pass), row by0 + lane (the Y pass: lane r holds row r's)
    const int x = bx0 + lane;
    const bool have_x = x <= bx1;
    const AxisEntry ex = axis_entry(p.cw, TW, min(x, bx1), T.at_x, T.alpha_x);
    const AxisEntry ey = axis_entry(p.ch, TH, min(by0 + lane, by1), T.at_y, T.alpha_y);
    PrepTile nxt = cur;
    int nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;
    if (more) {
      box_of(qn, tn, sn, nxt);
      nx0 = tab_entry_uniform(T.at_x, tab_index(qn.cw, TW, nxt.bx0)); nx1 = tab_entry_uniform(T.at_x, tab_index(qn.cw, TW, nxt.bx1));
      ny0 = tab_entry_uniform(T.at_y, tab_index(qn.ch, TH, nxt.by0)); ny1 = tab_entry_uniform(T.at_y, tab_index(qn.ch, TH, nxt.by1));
    }
    if (!cur.fits) {  // (a crop beyond 4/3 of the texture: the host launches the two-kernel form for such pools)
      if (lane == 0) atomicOr(err, kErrBgPrepCapacity);
    } else {
      const bool x_exact = __ballot(have_x && TW > p.cw && !(ex.u2 >> 31)) == 0ull;  // (the wave's columns all have exact weights: uniform)
      const uint32_t xdiv = 0xFFFFFFFFu / (uint32_t)p.cw + 1u, ydiv = 0xFFFFFFFFu / (uint32_t)p.ch + 1u;
      const bool xdiv24 = shrink_div24(p.cw), ydiv24 = shrink_div24(p.ch);
      // LDS columns of the X pass's three taps (clamped to what the tile holds; a clamped tap has weight 0 or is not read)
      const int xi0 = ex.u0 - cx0;
      const int xi1 = (TW > p.cw ? min(ex.u0 + 1, p.cw - 1) : min(ex.u0 + 1, cx1)) - cx0, xi2 = min(ex.u0 + 2, cx1) - cx0;
      const int ylastv = axis_last(p.ch, TH, ey.u0, ey.u1, ey.u2);
      uint32_t* Bs = B + (size_t)cur.s * TW * TH;
      // the tile's resize case (uniform): both axes enlarge with exact weights in every column and row of the tile / both shrink
      // with 24-bit quotients
      const bool y_exact = __ballot(lane <= by1 - by0 && TH > p.ch && !(ey.u2 >> 31)) == 0ull;
      const bool fast_enlarge = TW > p.cw && TH > p.ch && x_exact && y_exact;
      const bool fast_shrink = TW < p.cw && TH < p.ch && xdiv24 && ydiv24;
      if (paths && lane == 0)
        atomicAdd(&paths[(cur.inside ? ((kPrepDirs && cur.xdir && cur.ydir) ? 2 : 1) : 0) + 3 * (kPrepFastResize ? (fast_enlarge ? 1 : fast_shrink ? 2 : 0) : 0)], 1u);
      uint32_t m0 = 0, m1 = 0, m2 = 0;  // M(x, j - 2), M(x, j - 1), M(x, j)
      int y = by0;                      // the next row of B to emit ...
      int ylast = __builtin_amdgcn_readlane(ylastv, 0);  // ... and the last row of M it needs
      const int pairs = (ncw + 1) / 2;
      const uint32_t inv_pairs = (1u << 20) / (uint32_t)pairs + 1u;  // k / pairs = (k * inv) >> 20, exact for k <= 45 * 48
      for (int g0 = 0; g0 < nch; g0 += kPrepG) {
        const int rows = min(kPrepG, nch - g0);
        const int items = pairs * rows;
        // ---- rotation: C(cx0 + 2 pi .., cy0 + g0 + jj), texel pairs ----
        if (cur.inside) {
          // two rounds in flight: the next round's four gathers are requested before this round's texels are blended
          auto rotate = [&](auto xd, auto yd) {
            constexpr int kXD = decltype(xd)::value, kYD = decltype(yd)::value;
            auto issue = [&](int k) {
              const int jj = (int)(__umul24((uint32_t)k, inv_pairs) >> 20);
              const int pi = k - jj * pairs;
              const int xi = p.x0 + cx0 + 2 * pi;
              return rot_issue<kXD, kYD>(p, bgprep_rot_coords(p, __fsub_rn((float)xi, p.rw2), __fsub_rn((float)(xi + 1), p.rw2), __fsub_rn((float)(p.y0 + cy0 + g0 + jj), p.rh2)));
            };
#if OFDG_PREP_ROUND2
            // a round = TWO pairs per lane (k and k + 64), all eight gathers requested before the first wait: one exposed round trip
            // per four texels, uniform control flow (the last round's lanes beyond the end repeat the last pair), no set of taps copied
            for (int k0 = 0; k0 < items; k0 += 128) {
              const int ka = min(k0 + lane, items - 1), kb = min(k0 + 64 + lane, items - 1);
              const RotTaps ta = issue(ka);
              const RotTaps tb = issue(kb);
              const int ja = (int)(__umul24((uint32_t)ka, inv_pairs) >> 20), jb = (int)(__umul24((uint32_t)kb, inv_pairs) >> 20);
              *reinterpret_cast<uint2*>(&s_c[ja][2 * (ka - ja * pairs)]) = rot_finish<kXD>(ta);
              *reinterpret_cast<uint2*>(&s_c[jb][2 * (kb - jb * pairs)]) = rot_finish<kXD>(tb);
            }
          };
#else
            RotTaps tcur = issue(min(lane, items - 1));
            for (int k = lane; k < items; k += 64) {
              const int kn = k + 64;
              RotTaps tnext = tcur;
              if (kn < items) tnext = issue(kn);
              const int jj = (int)(__umul24((uint32_t)k, inv_pairs) >> 20);
              const int pi = k - jj * pairs;
              *reinterpret_cast<uint2*>(&s_c[jj][2 * pi]) = rot_finish<kXD>(tcur);  // (the odd texel beyond an odd-width region is inside too; nobody reads it)
              tcur = tnext;
            }
          };
#endif
          // (a tile lies on ONE side of the shift's mirror lines unless a line crosses it: the per-lane selects and mirrored
          //  indices of the general form drop out - 140 instead of 168 vector instructions per texel pair)
          using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
          if (kPrepDirs && cur.xdir == 1 && cur.ydir == 1) rotate(I1{}, I1{});
          else if (kPrepDirs && cur.xdir == 2 && cur.ydir == 1) rotate(I2{}, I1{});
          else if (kPrepDirs && cur.xdir == 1 && cur.ydir == 2) rotate(I1{}, I2{});
          else if (kPrepDirs && cur.xdir == 2 && cur.ydir == 2) rotate(I2{}, I2{});
          else rotate(I0{}, I0{});
        } else {
          // (the rare general form - mirrored crop coordinates, modulo, clamps: an opaque copy of the record's sizes, so that nothing it
          //  derives from them - floats, doubles of the image size - is computed in front of the tile and held through the fast forms)
          DevBgPrep q = p;
          asm volatile("" : "+s"(q.pw), "+s"(q.ph), "+s"(q.rw), "+s"(q.rh), "+s"(q.shx), "+s"(q.shy));
          for (int k = lane; k < items; k += 64) {
            const int jj = (int)(__umul24((uint32_t)k, inv_pairs) >> 20);
            const int pi = k - jj * pairs;
            const int i = cx0 + 2 * pi, j = cy0 + g0 + jj;
            const bool second = i + 1 <= cx1;
            const int rx0 = mirror_index(q.x0 + i, q.rw), rx1 = mirror_index(q.x0 + i + 1, q.rw), ry = mirror_index(q.y0 + j, q.rh);
            const float xc0 = __fsub_rn((float)rx0, q.rw2), xc1 = __fsub_rn((float)rx1, q.rw2), yc = __fsub_rn((float)ry, q.rh2);
            *reinterpret_cast<uint2*>(&s_c[jj][2 * pi]) = bgprep_rot_sample2(q, xc0, xc1, yc, second);
          }
        }
        __syncthreads();  // (one wave: the rows are complete before its lanes read their neighbours' texels)
        // ---- X resize into the register window, and every row of B that is complete with it ----
        // The general loop decides per row what its axis does (enlarge with an exact weight / in double, copy, shrink with a 24-bit
        // or a 32-bit quotient): a chain of uniform branches per row of C and per row of B.  A sample's zoom makes BOTH axes
        // enlarge (zoom > 1) or BOTH shrink (zoom < 1), and all weights of a tile compose can see are exact: the two loops
        // below are those two cases with nothing to decide inside (round 6; the resize passes are 40 % of the kernel's time).
        auto emit_rows = [&](int j, auto value) {  // every row y of B whose last source row is j
          while (y <= by1 && ylast <= j) {
            const int rr = y - by0;
            const int v0 = __builtin_amdgcn_readlane(ey.u0, rr);
            const uint32_t v1 = (uint32_t)__builtin_amdgcn_readlane((int)ey.u1, rr), v2 = (uint32_t)__builtin_amdgcn_readlane((int)ey.u2, rr);
            const uint32_t v = value(j - v0, v1, v2);
            // (non-temporal like compose's planes: written once here, read once by the next kernel - +3.5 % on the step over plain
            //  stores, profiles/r05_ab_nt_intermediate_stores.txt; raster's coverage bytes: no difference)
            if (have_x) __builtin_nontemporal_store(v, &Bs[(uint32_t)(y * TW + x)]);
            ++y;
            ylast = y <= by1 ? __builtin_amdgcn_readlane(ylastv, y - by0) : 0x7fffffff;
          }
        };
        if (kPrepFastResize && fast_enlarge) {
          const int xa1 = (int)ex.u1, xa0 = (int)(ex.u2 & 0x7fffffffu);
          for (int r = 0; r < rows; ++r) {
            const uint32_t t0 = s_c[r][xi0], t1 = s_c[r][xi1];
            m1 = m2;
            m2 = enlarge_texel_fix(t0, t1, xa1, xa0);
            // rows v0, v0 + 1 of M: the row's last source row j is v0 + 1, or v0 in the last row of M
            emit_rows(cy0 + g0 + r, [&](int dj, uint32_t v1, uint32_t v2) { return enlarge_texel_fix(dj >= 1 ? m1 : m2, m2, (int)v1, (int)(v2 & 0x7fffffffu)); });
          }
        } else if (kPrepFastResize && fast_shrink) {
          const uint32_t xd0 = ex.u1, xd1 = ex.u2, xd2 = (uint32_t)p.cw - ex.u1 - ex.u2;
          auto shrink24 = [](uint32_t t0, uint32_t t1, uint32_t t2, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t mdiv) {
            uint32_t acc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
              acc[c] = __umul24((t2 >> (8 * c)) & 255u, d2) + (__umul24((t1 >> (8 * c)) & 255u, d1) + __umul24((t0 >> (8 * c)) & 255u, d0));
            return mulhi_u24(acc[0], mdiv) | (mulhi_u24(acc[1], mdiv) << 8) | (mulhi_u24(acc[2], mdiv) << 16);
          };
          for (int r = 0; r < rows; ++r) {
            const uint32_t t0 = s_c[r][xi0], t1 = s_c[r][xi1], t2 = s_c[r][xi2];
            m0 = m1; m1 = m2;
            m2 = shrink24(t0, t1, t2, xd0, xd1, xd2, xdiv);
            // rows v0 .. v0 + 2 of M, clamped to j: j is v0 + 2, or v0 + 1 where the third tap has no weight
            emit_rows(cy0 + g0 + r, [&](int dj, uint32_t v1, uint32_t v2) {
              const uint32_t t0y = dj >= 2 ? m0 : (dj == 1 ? m1 : m2), t1y = dj >= 2 ? m1 : m2;
              return shrink24(t0y, t1y, m2, v1, v2, (uint32_t)p.ch - v1 - v2, ydiv);
            });
          }
        } else {
        int xg = x, cwg = p.cw, chg = p.ch;  // (opaque: see the general rotation above)
        asm volatile("" : "+v"(xg), "+s"(cwg), "+s"(chg));
        for (int r = 0; r < rows; ++r) {
          const int j = cy0 + g0 + r;
          m0 = m1; m1 = m2;
          m2 = have_x ? axis_texel(cwg, TW, xg, ex.u1, ex.u2, x_exact, s_c[r][xi0], s_c[r][xi1], s_c[r][xi2], xdiv, xdiv24, T.alpha_x) : 0u;
          emit_rows(j, [&](int dj, uint32_t v1, uint32_t v2) {
            // rows v0, v0 + 1, v0 + 2 of M (uniform), clamped to j = the row's last source row: the window holds j - 2 .. j
            // (enlarging: j is v0 + 1, or v0 in the last row of M; shrinking: v0 + 2, or v0 + 1 where the third tap has no weight)
            uint32_t t0 = m2, t1 = m2;
            if (dj >= 2) { t0 = m0; t1 = m1; } else if (dj == 1) { t0 = m1; }
            return axis_texel(chg, TH, y, v1, v2, (v2 >> 31) != 0u, t0, t1, m2, ydiv, ydiv24, T.alpha_y);
          });
        }
        }
        __syncthreads();  // (the next group overwrites the rows)
      }
    }
    if (!more) break;
    ex0 = nx0; ex1 = nx1; ey0 = ny0; ey1 = ny1;
    cur = nxt;
    t = tn;
  }
}

// one thread = 4 consecutive texels of a row of sample blockIdx.y's texture; texels outside the
// sample's read region (DevBgPrep.r*) are skipped - compose never looks at them
__global__ __launch_bounds__(256) void bgprep_kernel(const DevBgPrep* __restrict__ prep, int W, int H, uint32_t* __restrict__ bgtex) {
  const int TW = 2 * W, TH = 2 * H, quads = TW / 4;
  const int s = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= quads * TH) return;
  const int v = i / quads, u0 = (i - v * quads) * 4;
  const DevBgPrep p = prep[s];
  if (v < p.ry0 || v > p.ry1 || u0 + 3 < p.rx0 || u0 > p.rx1) return;
  uint4 o;
  o.x = bgprep_texel(p, u0, v);
  o.y = bgprep_texel(p, u0 + 1, v);
  o.z = bgprep_texel(p, u0 + 2, v);
  o.w = bgprep_texel(p, u0 + 3, v);
  *reinterpret_cast<uint4*>(bgtex + (size_t)s * TW * TH + (size_t)v * TW + u0) = o;
}

__global__ __launch_bounds__(256) void pool_synth_kernel(uint32_t* __restrict__ pool, int n, int w, int h, uint32_t seed) {
  const size_t total = (size_t)n * w * h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t tex = (uint32_t)(i / ((size_t)w * h));
    const uint32_t rem = (uint32_t)(i - (size_t)tex * w * h);
    const uint32_t y = rem / w, x = rem - y * w;
    uint32_t px = 0;
    for (uint32_t c = 0; c < 3; ++c) {
      uint32_t acc = 0;
      for (uint32_t oct = 0; oct < 3; ++oct) {
        const uint32_t sh = 6 - 2 * oct, cs = 1u << sh;
        const uint32_t ix = x >> sh, iy = y >> sh, fx = x & (cs - 1), fy = y & (cs - 1);
        const uint32_t l00 = lattice(seed, tex, c, oct, ix, iy), l10 = lattice(seed, tex, c, oct, ix + 1, iy);
        const uint32_t l01 = lattice(seed, tex, c, oct, ix, iy + 1), l11 = lattice(seed, tex, c, oct, ix + 1, iy + 1);
        const uint32_t v = (l00 * (cs - fx) * (cs - fy) + l10 * fx * (cs - fy) + l01 * (cs - fx) * fy + l11 * fx * fy) >> (2 * sh);
        acc += v << (2 - oct);
      }
      px |= ((acc + 3) / 7) << (8 * c);
    }
    pool[i] = px;
  }
}
__global__ __launch_bounds__(256) void pool_pack_kernel(const uint8_t* __restrict__ planar, uint32_t* __restrict__ dst, int w, int h) {
  const size_t n = (size_t)w * h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (uint32_t)planar[i] | ((uint32_t)planar[n + i] << 8) | ((uint32_t)planar[2 * n + i] << 16);
}
__global__ __launch_bounds__(256) void pool_unpack_kernel(const uint32_t* __restrict__ src, uint8_t* __restrict__ planar, int w, int h) {
  const size_t n = (size_t)w * h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t p = src[i];
    planar[i] = p & 255; planar[n + i] = (p >> 8) & 255; planar[2 * n + i] = (p >> 16) & 255;
  }
}

// CImg<unsigned char>::get_resize(.., 3) along ONE axis of every image of a BGRX pool (the
// reference resizes pool images that are smaller than the texture it needs, DG:102-106).
// Enlarging: linear, table `at` = source index, `alpha` = weight (CImg's running double sums, built
// on the host); value (T)((1 - a) * v1 + a * v2).  Shrinking (at == nullptr): moving average over
// the w*s grid in float, / w, truncated.  dst images are ow x oh, src w x h.
__global__ __launch_bounds__(256) void pool_resize_axis_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n_images,
                                                               int w, int h, int s, int along_x, const int* __restrict__ at,
                                                               const double* __restrict__ alpha) {
  const int ow = along_x ? s : w, oh = along_x ? h : s;
  const size_t per = (size_t)ow * oh, total = per * n_images;
  const int n = along_x ? w : h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t img = i / per, r = i - img * per;
    const int x = (int)(r % ow), y = (int)(r / ow);
    const int k = along_x ? x : y, line = along_x ? y : x;
    const uint32_t* sp = src + img * (size_t)w * h;
    auto texel = [&](int j) { return along_x ? sp[(size_t)line * w + j] : sp[(size_t)j * w + line]; };
    uint32_t out = 0;
    if (at) {
      const int a0 = at[k];
      const double al = alpha[k];
      const uint32_t t1 = texel(a0), t2 = a0 < n - 1 ? texel(a0 + 1) : t1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double v1 = (double)((t1 >> (8 * c)) & 255u), v2 = (double)((t2 >> (8 * c)) & 255u);
        out |= (uint32_t)(unsigned char)((1 - al) * v1 + al * v2) << (8 * c);
      }
    } else {
      // destination k integrates [k*n, (k+1)*n) of the n*s grid; source j covers [j*s, (j+1)*s)
      float acc[3] = {0.f, 0.f, 0.f};
      const long long lo = (long long)k * n, hi = lo + n;
      for (int j = (int)(lo / s); (long long)j * s < hi; ++j) {
        const long long a = (long long)j * s, b = a + s;
        const float d = (float)((b < hi ? b : hi) - (a > lo ? a : lo));
        const uint32_t t = texel(j);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn((float)((t >> (8 * c)) & 255u), d));
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) out |= (uint32_t)(unsigned char)__fdiv_rn(acc[c], (float)n) << (8 * c);
    }
    dst[i] = out;
  }
}

// include/ofdg_detmath.h evaluated on the device (tests: device == host bit for bit)
// Debug entries of the tests: the device's curve flattening and span interpolator on caller-given doubles.
// (path: one wave, the same path_verts the outlines go through; writes the vertices + their number)
__global__ __launch_bounds__(64) void debug_path_kernel(const double* __restrict__ xy, const int* __restrict__ types, int n_seg,
                                                        int2* __restrict__ verts, int* __restrict__ n_out, uint32_t* __restrict__ err) {
  __shared__ double s_stack[kCurveSlots][kCurveMaxDepth][5];
  __shared__ int2 s_stage[kCurveSlots][kCurveMaxPts];
  const int lane = (int)threadIdx.x;
  int minx = 0x7FFFFFFF, miny = 0x7FFFFFFF, maxx = (int)0x80000000, maxy = (int)0x80000000;
  const int n = path_verts(n_seg, [&](int i) { return types[i]; }, [&](int i, double& x, double& y) { x = xy[2 * i]; y = xy[2 * i + 1]; },
                           verts, s_stack, s_stage, err, lane, minx, miny, maxx, maxy);
  if (lane == 0) *n_out = n;
}
// (span interpolator: make_row + dda_at of `rows` output rows of length `len` under the inverse affine `inv`)
__global__ void debug_dda_kernel(Mat inv, int rows, int len, int2* __restrict__ out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= rows * len) return;
  const int y = i / len, x = i - y * len;
  const int nshift = ((len & (len - 1)) == 0) ? (31 - __clz(len)) : -1;
  const RowDDA R = make_row(inv, y, len, nshift);
  out[i] = make_int2(dda_at(R.x1, R.lx, R.rx, len, nshift, x), dda_at(R.y1, R.ly, R.ry, len, nshift, x));
}

// ofdg_poll_errors / ofdg_poll_errors_of: read and clear n device error words in one step (one wave), *out = their OR
__global__ void err_exchange_kernel(uint32_t* __restrict__ d_err, int n, uint32_t* __restrict__ out) {
  uint32_t e = 0;
  for (int i = (int)threadIdx.x; i < n; i += 64) e |= atomicExch(d_err + i, 0u);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) e |= (uint32_t)__shfl_xor((int)e, d, 64);
  if (threadIdx.x == 0) *out = e;
}

__global__ void detmath_kernel(const double* __restrict__ a, int n, double* __restrict__ s, double* __restrict__ c,
                               const float* __restrict__ x, int m, float* __restrict__ e) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ofdg_det_sincos(a[i], s + i, c + i);
  if (i < m) e[i] = ofdg_det_expf(x[i]);
}

// exhaustive probe of the per-byte device formulas (tests): tables of 65536 entries
__global__ void tables_kernel(uint8_t* add_tbl, uint8_t* sub_tbl, uint8_t* aa_tbl, uint8_t* blend_tbl /*256*256 for s=200? no: d,m with s fixed*/, int s_fixed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 65536) return;
  const int u = i >> 8, v = i & 255;
  add_tbl[i] = (uint8_t)comp_add(u, v);
  sub_tbl[i] = (uint8_t)comp_sub(u, v);
  blend_tbl[i] = (uint8_t)((blend_px((uint32_t)u * 0x00010101u, (uint32_t)s_fixed * 0x00010101u, (uint32_t)v) >> 16) & 255u);  // d = u, m = v (R lane)
  if (((blend_px((uint32_t)u * 0x00010101u, (uint32_t)s_fixed * 0x00010101u, (uint32_t)v)) & 255u) != (uint32_t)blend(u, s_fixed, v) || ((blend_px((uint32_t)u * 0x00010101u, (uint32_t)s_fixed * 0x00010101u, (uint32_t)v) >> 8) & 255u) != (uint32_t)blend(u, s_fixed, v)) blend_tbl[i] = 0xEE;  // all three lanes must agree with the scalar form
  if (i < 256) aa_tbl[i] = (uint8_t)aa_byte(i);
}

}  // namespace ofdg
