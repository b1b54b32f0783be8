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

#include "ofdg_device.h"

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
// geom_kernel: one workgroup per (shape, frame).
// Reference: RealizeObjectBlueprint geometry (DG:1073-1117), conv_transform +
// conv_curve feeding rasterizer_scanline_aa::add_path (DG:465-479, 520-534),
// agg::ellipse (100 steps), agg::curve3_div, ras_conv_int::upscale = iround(v*256).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(128) void geom_kernel(const DevShape* __restrict__ shapes, int n_shapes,
                                                   const double* __restrict__ cs_tab, int W, int H,
                                                   DevShapeFrame* __restrict__ frames, int2* __restrict__ verts,
                                                   uint32_t* __restrict__ err) {
  __shared__ int s_cnt[kMaxSegments + 1];
  __shared__ int s_bbox[4];
  __shared__ int2 s_stage[kMaxSegments][kCurveMaxPts];
  __shared__ double s_stack[kMaxSegments][kCurveMaxDepth][7];

  const int sf = blockIdx.x;
  if (sf >= n_shapes * 2) return;
  const int tid = threadIdx.x;
  const DevShape& S = shapes[sf >> 1];
  const Mat M = S.m[sf & 1];
  int2* out = verts + (size_t)sf * kMaxVerts;

  if (tid == 0) {
    s_bbox[0] = 0x7FFFFFFF; s_bbox[1] = 0x7FFFFFFF;
    s_bbox[2] = (int)0x80000000; s_bbox[3] = (int)0x80000000;
  }
  if (tid <= kMaxSegments) s_cnt[tid] = 0;
  __syncthreads();

  int n_verts = 0;
  if (S.type == 1) {
    // agg::ellipse::vertex: x = cx + cos(angle)*rx with angle = step/100 * 2*pi; the
    // cos/sin table comes from the host's libm so the doubles match the CPU's.
    n_verts = 100;
    if (tid < 100) {
      double x = 0.0 + cs_tab[2 * tid] * (double)S.rx;
      double y = 0.0 + cs_tab[2 * tid + 1] * (double)S.ry;
      xform(M, x, y);
      int2 v = make_int2(iround_d(x * 256.0), iround_d(y * 256.0));
      out[tid] = v;
      atomicMin(&s_bbox[0], v.x); atomicMin(&s_bbox[1], v.y);
      atomicMax(&s_bbox[2], v.x); atomicMax(&s_bbox[3], v.y);
    }
  } else {
    const int n_seg = S.n_seg;
    // pass 1: every segment flattens into its staging row
    if (tid < n_seg) {
      const int t = (tid == 0) ? 1 : S.seg_type[tid];  // segment 0 is the move_to vertex
      int cnt = 0;
      if (t == 1) {
        double x = (double)S.seg_x[tid], y = (double)S.seg_y[tid];
        xform(M, x, y);
        s_stage[tid][0] = make_int2(iround_d(x * 256.0), iround_d(y * 256.0));
        cnt = 1;
      } else if (t == 3) {
        // conv_curve: curve3(ctrl = seg[i], to = seg[i+1]) from the current point seg[i-1]
        double x1 = (double)S.seg_x[tid - 1], y1 = (double)S.seg_y[tid - 1];
        double x2 = (double)S.seg_x[tid], y2 = (double)S.seg_y[tid];
        const int ie = (tid + 1 < n_seg) ? tid + 1 : tid;
        double x3 = (double)S.seg_x[ie], y3 = (double)S.seg_y[ie];
        xform(M, x1, y1); xform(M, x2, y2); xform(M, x3, y3);
        const double ex = x3, ey = y3;
        // curve3_div::recursive_bezier as an explicit depth-first walk
        const double tol_sq = 0.25;  // (0.5 / approximation_scale)^2
        int sp = 0;
        int level = 0;
        bool have = true, bad = false;
        while (have) {
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
              if (cnt < kCurveMaxPts - 1) s_stage[tid][cnt++] = make_int2(iround_d(px * 256.0), iround_d(py * 256.0));
              else bad = true;
            }
            if (subdivide) {
              if (sp < kCurveMaxDepth) {
                double* st = s_stack[tid][sp++];
                st[0] = x123; st[1] = y123; st[2] = x23; st[3] = y23; st[4] = x3; st[5] = y3; st[6] = (double)(level + 1);
                // descend into the left half
                x3 = x123; y3 = y123; x2 = x12; y2 = y12;
                level = level + 1;
                continue;
              }
              bad = true;
            }
          }
          if (sp > 0) {
            const double* st = s_stack[tid][--sp];
            x1 = st[0]; y1 = st[1]; x2 = st[2]; y2 = st[3]; x3 = st[4]; y3 = st[5]; level = (int)st[6];
          } else {
            have = false;
          }
        }
        s_stage[tid][cnt++] = make_int2(iround_d(ex * 256.0), iround_d(ey * 256.0));
        if (bad) atomicOr(err, kErrCurveCapacity);
      }
      s_cnt[tid] = cnt;
    }
    __syncthreads();
    if (tid == 0) {  // exclusive scan over <= 20 segments
      int acc = 0;
      for (int i = 0; i < n_seg; ++i) { int c = s_cnt[i]; s_cnt[i] = acc; acc += c; }
      s_cnt[kMaxSegments] = acc;
    }
    __syncthreads();
    n_verts = s_cnt[kMaxSegments];
    if (n_verts > kMaxVerts) {
      if (tid == 0) atomicOr(err, kErrVertCapacity);
      n_verts = 0;
    } else if (tid < n_seg) {
      const int off = s_cnt[tid];
      const int end = (tid + 1 < n_seg) ? s_cnt[tid + 1] : n_verts;
      for (int k = 0; k < end - off; ++k) {
        int2 v = s_stage[tid][k];
        out[off + k] = v;
        atomicMin(&s_bbox[0], v.x); atomicMin(&s_bbox[1], v.y);
        atomicMax(&s_bbox[2], v.x); atomicMax(&s_bbox[3], v.y);
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    DevShapeFrame f;
    f.n_verts = n_verts;
    f.pad[0] = f.pad[1] = f.pad[2] = 0;
    int x0 = s_bbox[0] >> 8, y0 = s_bbox[1] >> 8, x1 = s_bbox[2] >> 8, y1 = s_bbox[3] >> 8;
    // AGG dx_limit: an edge spanning >= 16384 px takes a different code path in
    // rasterizer_cells_aa::line; blueprints never get close, flag it if one does.
    if (n_verts > 0 && ((long long)s_bbox[2] - (long long)s_bbox[0] >= (16384LL << 8))) {
      atomicOr(err, kErrDxLimit);
      n_verts = 0;
    }
    if (n_verts < 2 || x1 < 0 || y1 < 0 || x0 > W - 1 || y0 > H - 1) {
      x0 = 1; x1 = 0; y0 = 1; y1 = 0;  // nothing on screen
    } else {
      x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, W - 1); y1 = min(y1, H - 1);
    }
    f.x0 = x0; f.y0 = y0; f.x1 = x1; f.y1 = y1;
    frames[sf] = f;
  }
}

// --------------------------------------------------------------------------
// raster_kernel: one workgroup per (shape-frame, band of kBandRows scanlines).
//
// AGG's rasterizer_cells_aa walks each edge scanline by scanline and cell by cell
// with incremental lift/rem/mod stepping.  Its per-cell sums of (cover, area) are
// order independent, and the stepping has a closed form (floor of the cumulative
// rational), so every (edge, scanline) pair is processed by its own thread and
// accumulated into LDS with integer atomics.  Column 0 of a row collects the cover
// of all cells left of the shape's on-screen bounding box.
// Then rasterizer_scanline_aa::sweep_scanline: running cover prefix sum per row
// (wavefront scan), alpha = min(|((C << 9) - area) >> 9|, 255) (non-zero rule,
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

// rasterizer_cells_aa::render_hline(ey, xa, ya, xb, yb) in closed form.
__device__ __forceinline__ void hline(const CellAcc& acc, int r, int xa, int ya, int xb, int yb) {
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
__device__ __forceinline__ void edge_scanline(const CellAcc& acc, int r, int y, int x1, int y1, int x2, int y2) {
  const int ey1 = y1 >> 8, ey2 = y2 >> 8;
  const int fy1 = y1 & 255, fy2 = y2 & 255;
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

__global__ __launch_bounds__(256) void raster_kernel(const DevShapeFrame* __restrict__ frames, int n_sf,
                                                     const int2* __restrict__ verts, int W, int H,
                                                     uint8_t* __restrict__ cov) {
  extern __shared__ int s_cells[];  // cover[kBandRows][pitch], area[kBandRows][pitch]
  __shared__ int2 s_verts[kMaxVerts];
  const int sf = blockIdx.x;
  if (sf >= n_sf) return;
  const DevShapeFrame F = frames[sf];
  if (F.x0 > F.x1) return;
  const int by0 = blockIdx.y * kBandRows;
  const int ylo = max(by0, F.y0), yhi = min(by0 + kBandRows - 1, F.y1);
  if (ylo > yhi) return;

  const int tid = threadIdx.x;
  const int ncols = F.x1 - F.x0 + 1;
  const int pitch = ncols + 1;
  CellAcc acc;
  acc.cover = s_cells;
  acc.area = s_cells + kBandRows * pitch;
  acc.pitch = pitch; acc.X0 = F.x0; acc.X1 = F.x1;

  for (int i = tid; i < 2 * kBandRows * pitch; i += 256) s_cells[i] = 0;
  const int nv = F.n_verts;
  const int2* vsrc = verts + (size_t)sf * kMaxVerts;
  for (int i = tid; i < nv; i += 256) s_verts[i] = vsrc[i];
  __syncthreads();

  // (edge, scanline) work items; the closing edge (last -> first) is edge nv-1
  const int n_items = nv * kBandRows;
  for (int it = tid; it < n_items; it += 256) {
    const int r = it & (kBandRows - 1);
    const int e = it / kBandRows;
    const int y = by0 + r;
    if (y < ylo || y > yhi) continue;
    const int2 a = s_verts[e];
    const int2 b = s_verts[(e + 1 == nv) ? 0 : e + 1];
    edge_scanline(acc, r, y, a.x, a.y, b.x, b.y);
  }
  __syncthreads();

  // sweep: one wave per row, 64 columns per step
  const int wave = tid >> 6, lane = tid & 63;
  for (int r = wave; r < kBandRows; r += 4) {
    const int y = by0 + r;
    if (y < ylo || y > yhi) continue;  // wave-uniform
    int carry = acc.cover[r * pitch];
    uint8_t* dst = cov + ((size_t)sf * H + y) * W + F.x0;
    for (int c0 = 0; c0 < ncols; c0 += 64) {
      const int c = c0 + lane;
      int cv = 0, ar = 0;
      if (c < ncols) { cv = acc.cover[r * pitch + c + 1]; ar = acc.area[r * pitch + c + 1]; }
      // inclusive wave prefix sum of cover
      int s = cv;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(s, d, 64);
        if (lane >= d) s += t;
      }
      const int C = carry + s;
      int a = ((C << 9) - ar) >> 9;  // calculate_alpha: poly_subpixel_shift*2 + 1 - aa_shift = 9
      a = a < 0 ? -a : a;
      a = a > 255 ? 255 : a;
      if (c < ncols) dst[c] = (uint8_t)a;
      carry += __shfl(s, 63, 64);
    }
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

// MovingObjectComposite::renderMasks, strict fp32 (DG:606, 626)
__device__ __forceinline__ int comp_add(int u, int v) {
  const float fu = __fdiv_rn((float)u, 255.f), fv = __fdiv_rn((float)v, 255.f);
  const float t = __fmul_rn(__fsub_rn(1.f, fu), __fsub_rn(1.f, fv));
  return (int)(unsigned char)__fmul_rn(255.f, __fsub_rn(1.f, t));
}
__device__ __forceinline__ int comp_sub(int u, int v) {
  const float fu = __fdiv_rn((float)u, 255.f), fv = __fdiv_rn((float)v, 255.f);
  return (int)(unsigned char)__fmul_rn(255.f, __fmul_rn(fu, __fsub_rn(1.f, fv)));
}

// span_interpolator_linear::begin + dda2_line_interpolator for one output row.
struct RowDDA {
  int x1, lx, rx;  // start, lft, rem (rem in [1, n])
  int y1, ly, ry;
};
__device__ __forceinline__ void dda_setup(int a, int b, int n, int& lft, int& rem) {
  const int D = b - a;
  lft = D / n;
  rem = D % n;
  if (rem <= 0) { rem += n; lft--; }
}
__device__ __forceinline__ RowDDA make_row(const Mat& inv, int row, int len) {
  RowDDA R;
  double tx = 0 + 0.5, ty = row + 0.5;
  xform(inv, tx, ty);
  R.x1 = iround_d(tx * 256.0);
  R.y1 = iround_d(ty * 256.0);
  tx = 0 + 0.5 + len; ty = row + 0.5;
  xform(inv, tx, ty);
  const int x2 = iround_d(tx * 256.0), y2 = iround_d(ty * 256.0);
  dda_setup(R.x1, x2, len, R.lx, R.rx);
  dda_setup(R.y1, y2, len, R.ly, R.ry);
  return R;
}
// value of the interpolator after i increments
__device__ __forceinline__ int dda_at(int y1, int lft, int rem, int n, int nshift, int i) {
  const int a = (i + 1) * rem + n - 1;
  const int q = (nshift >= 0) ? (a >> nshift) : (a / n);
  return y1 + i * lft + q - 1;
}
__device__ __forceinline__ int wrap_reflect(int v, int size, int size2, int mask2, int& raw) {
  int m = (mask2 >= 0) ? (v & mask2) : (v % size2);
  if (m < 0) m += size2;
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
__device__ __forceinline__ uint32_t sample_bilinear(const uint32_t* __restrict__ tex, const WarpGeom& g,
                                                    const RowDDA& R, int i) {
  int x_hr = dda_at(R.x1, R.lx, R.rx, g.tw, g.nshift, i) - 128;
  int y_hr = dda_at(R.y1, R.ly, R.ry, g.tw, g.nshift, i) - 128;
  const int x_lr = x_hr >> 8, y_lr = y_hr >> 8;
  x_hr &= 255; y_hr &= 255;
  int rx, ry;
  const int xa = wrap_reflect(x_lr, g.tw, g.tw2, g.mx2, rx);
  const int ya = wrap_reflect(y_lr, g.th, g.th2, g.my2, ry);
  const int xb = wrap_next(rx, g.tw, g.tw2);
  const int yb = wrap_next(ry, g.th, g.th2);
  const uint32_t p00 = tex[(size_t)ya * g.pitch + xa];
  const uint32_t p10 = tex[(size_t)ya * g.pitch + xb];
  const uint32_t p01 = tex[(size_t)yb * g.pitch + xa];
  const uint32_t p11 = tex[(size_t)yb * g.pitch + xb];
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

constexpr int kTileW = 64, kTileH = 16, kPx = 4;

// One thread renders kPx horizontally adjacent pixels; a 256-thread workgroup a
// 64 x 16 tile.  Objects are visited in painter's order (ascending ID); objects
// whose masks cannot touch the tile are skipped with scalar tests.
// Reference: Process_TaskBucket DG:1216-1245, blitObject DG:762-799,
// computeFlowImage/getPointFlow DG:801-818, 388-407, 692-718.
__global__ __launch_bounds__(256) void compose_kernel(RenderDims dm, const DevSample* __restrict__ samples,
                                                      const DevObject* __restrict__ objects,
                                                      const DevShapeFrame* __restrict__ frames,
                                                      const uint8_t* __restrict__ cov,
                                                      const uint32_t* __restrict__ pool,
                                                      float* __restrict__ img0, float* __restrict__ img1,
                                                      float* __restrict__ flow) {
  // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch), so give
  // every XCD a contiguous run of tiles (whole samples): their background rows,
  // coverage slots and object records then stay in that XCD's L2.
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, rm = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + slot;
  }
  const int tiles = dm.tiles_x * dm.tiles_y;
  const int s = bid / tiles;
  if (s >= dm.n_samples) return;
  const int t = bid - s * tiles;
  const int ty0 = (t / dm.tiles_x) * kTileH, tx0 = (t % dm.tiles_x) * kTileW;
  const int W = dm.W, H = dm.H;
  const int x0 = tx0 + (threadIdx.x & 15) * kPx;
  const int y = ty0 + (threadIdx.x >> 4);
  const bool inside = (x0 < W) && (y < H);  // W % 4 == 0 is required by the host

  const DevSample smp = samples[s];
  const DevObject* objs = objects + smp.first_object;

  int f0[3][kPx], f1[3][kPx];
  float fu[kPx], fv[kPx];

  // ---- background (object 0): masks are all 255, so frames start as its textures ----
  {
    const DevObject& B = objs[0];
    const uint32_t* tex = pool + B.tex_base;  // origin of the 2W x 2H centre crop
    WarpGeom g;
    g.tw = 2 * W; g.th = 2 * H; g.tw2 = 4 * W; g.th2 = 4 * H;
    g.mx2 = ((g.tw2 & (g.tw2 - 1)) == 0) ? g.tw2 - 1 : -1;
    g.my2 = ((g.th2 & (g.th2 - 1)) == 0) ? g.th2 - 1 : -1;
    g.nshift = ((g.tw & (g.tw - 1)) == 0) ? (31 - __clz(g.tw)) : -1;
    g.pitch = dm.pool_w;
    const int yy = y + H / 2, xx = x0 + W / 2;
    if (inside) {
      // frame 0: identity warp == copy, then the crop at (W/2, H/2)  (DG:667-668, 680)
      const uint4 t0 = *reinterpret_cast<const uint4*>(tex + (size_t)yy * g.pitch + xx);
      const uint32_t tt[4] = {t0.x, t0.y, t0.z, t0.w};
      const RowDDA R = make_row(B.tex_inv, yy, g.tw);
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        const uint32_t t1 = sample_bilinear(tex, g, R, xx + p);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          f0[c][p] = (tt[p] >> (8 * c)) & 255;
          f1[c][p] = (t1 >> (8 * c)) & 255;
        }
        // MovingObjectBackground::getPointFlow (DG:692-718)
        double ix = (double)(x0 + p + W / 2), iy = (double)(y + H / 2);
        const float save_x = (float)ix, save_y = (float)iy;
        ix = ix * 1.0 + iy * 0.0 + (double)(-W); iy = iy + (double)(-H);  // intrinsic_inv = T(-W,-H)
        xform(B.motion, ix, iy);
        ix = ix + (double)W; iy = iy + (double)H;                        // intrinsic = T(W,H)
        fu[p] = (float)(ix - (double)save_x);
        fv[p] = (float)(iy - (double)save_y);
      }
    } else {
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        fu[p] = fv[p] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) f0[c][p] = f1[c][p] = 0;
      }
    }
  }

  // ---- foreground objects in z-order ----
  WarpGeom g;
  g.tw = W; g.th = H; g.tw2 = 2 * W; g.th2 = 2 * H;
  g.mx2 = ((g.tw2 & (g.tw2 - 1)) == 0) ? g.tw2 - 1 : -1;
  g.my2 = ((g.th2 & (g.th2 - 1)) == 0) ? g.th2 - 1 : -1;
  g.nshift = ((W & (W - 1)) == 0) ? (31 - __clz(W)) : -1;
  g.pitch = dm.pool_w;

  for (int oi = 1; oi < smp.n_objects; ++oi) {
    const DevObject& O = objs[oi];
    // scalar cull: does any component mask touch this tile (either frame)?
    bool touch = false;
    for (int k = 0; k < O.n_shapes; ++k) {
#pragma unroll
      for (int fr = 0; fr < 2; ++fr) {
        const DevShapeFrame& F = frames[(O.first_shape + k) * 2 + fr];
        touch |= (F.x0 <= tx0 + kTileW - 1) && (F.x1 >= tx0) && (F.y0 <= ty0 + kTileH - 1) && (F.y1 >= ty0);
      }
    }
    if (!touch) continue;

    int m0[kPx], m1[kPx];   // blending masks for the two frames
    int na0[kPx];           // thresholded frame-0 mask (index image)
    if (O.kind == 1) {
      const int sf0 = O.first_shape * 2;
      const DevShapeFrame& F0 = frames[sf0];
      const DevShapeFrame& F1 = frames[sf0 + 1];
      const bool r0 = inside && y >= F0.y0 && y <= F0.y1;
      const bool r1 = inside && y >= F1.y0 && y <= F1.y1;
      uint32_t c0w = 0, c1w = 0;
      if (r0) c0w = *reinterpret_cast<const uint32_t*>(cov + ((size_t)sf0 * H + y) * W + x0);
      if (r1) c1w = *reinterpret_cast<const uint32_t*>(cov + ((size_t)(sf0 + 1) * H + y) * W + x0);
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        const int x = x0 + p;
        const int c0 = (r0 && x >= F0.x0 && x <= F0.x1) ? (int)((c0w >> (8 * p)) & 255) : 0;
        const int c1 = (r1 && x >= F1.x0 && x <= F1.x1) ? (int)((c1w >> (8 * p)) & 255) : 0;
        na0[p] = c0 >= 128 ? 255 : 0;
        m0[p] = dm.use_aa ? aa_byte(c0) : na0[p];
        m1[p] = dm.use_aa ? aa_byte(c1) : (c1 >= 128 ? 255 : 0);
      }
    } else {
      // composite: sequential fp32 add / subtract over the components (DG:591-646)
      int ua0[kPx], ua1[kPx], un1[kPx];
#pragma unroll
      for (int p = 0; p < kPx; ++p) { ua0[p] = ua1[p] = un1[p] = 0; na0[p] = 0; }
      for (int k = 0; k < O.n_shapes; ++k) {
        const int sf0 = (O.first_shape + k) * 2;
        const DevShapeFrame& F0 = frames[sf0];
        const DevShapeFrame& F1 = frames[sf0 + 1];
        const bool r0 = inside && y >= F0.y0 && y <= F0.y1;
        const bool r1 = inside && y >= F1.y0 && y <= F1.y1;
        uint32_t c0w = 0, c1w = 0;
        if (r0) c0w = *reinterpret_cast<const uint32_t*>(cov + ((size_t)sf0 * H + y) * W + x0);
        if (r1) c1w = *reinterpret_cast<const uint32_t*>(cov + ((size_t)(sf0 + 1) * H + y) * W + x0);
        const bool additive = (O.additive >> k) & 1u;
#pragma unroll
        for (int p = 0; p < kPx; ++p) {
          const int x = x0 + p;
          const int c0 = (r0 && x >= F0.x0 && x <= F0.x1) ? (int)((c0w >> (8 * p)) & 255) : 0;
          const int c1 = (r1 && x >= F1.x0 && x <= F1.x1) ? (int)((c1w >> (8 * p)) & 255) : 0;
          const int va0 = aa_byte(c0), va1 = aa_byte(c1);
          const int vn0 = c0 >= 128 ? 255 : 0, vn1 = c1 >= 128 ? 255 : 0;
          if (additive) {
            ua0[p] = comp_add(ua0[p], va0); ua1[p] = comp_add(ua1[p], va1);
            na0[p] = comp_add(na0[p], vn0); un1[p] = comp_add(un1[p], vn1);
          } else {
            ua0[p] = comp_sub(ua0[p], va0); ua1[p] = comp_sub(ua1[p], va1);
            na0[p] = comp_sub(na0[p], vn0); un1[p] = comp_sub(un1[p], vn1);
          }
        }
      }
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        m0[p] = dm.use_aa ? ua0[p] : na0[p];
        m1[p] = dm.use_aa ? ua1[p] : un1[p];
      }
    }

    const int any0 = m0[0] | m0[1] | m0[2] | m0[3];
    const int any1 = m1[0] | m1[1] | m1[2] | m1[3];
    const int anyn = na0[0] | na0[1] | na0[2] | na0[3];
    const uint32_t* tex = pool + O.tex_base;  // origin of the W x H centre crop
    if (any0) {
      // frame 0 texture: identity warp == the crop itself (DG:339-340)
      const uint4 t0 = *reinterpret_cast<const uint4*>(tex + (size_t)y * g.pitch + x0);
      const uint32_t tt[4] = {t0.x, t0.y, t0.z, t0.w};
#pragma unroll
      for (int p = 0; p < kPx; ++p)
#pragma unroll
        for (int c = 0; c < 3; ++c) f0[c][p] = blend(f0[c][p], (tt[p] >> (8 * c)) & 255, m0[p]);
    }
    if (any1) {
      const RowDDA R = make_row(O.tex_inv, y, W);
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        if (m1[p]) {
          const uint32_t t1 = sample_bilinear(tex, g, R, x0 + p);
#pragma unroll
          for (int c = 0; c < 3; ++c) f1[c][p] = blend(f1[c][p], (t1 >> (8 * c)) & 255, m1[p]);
        }
      }
    }
    if (anyn) {
      // MovingObjectBase::getPointFlow (DG:388-407) for pixels this object now owns
#pragma unroll
      for (int p = 0; p < kPx; ++p) {
        if (na0[p] == 255) {
          double ix = (double)(x0 + p), iy = (double)y;
          const float save_x = (float)ix, save_y = (float)iy;
          xform(O.motion, ix, iy);
          fu[p] = (float)(ix - (double)save_x);
          fv[p] = (float)(iy - (double)save_y);
        }
      }
    }
  }

  if (!inside) return;
  // u8 -> float planes (DG:1229-1245); streaming 16-byte stores, never re-read
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const size_t plane = (size_t)W * H;
  const size_t o = (size_t)y * W + x0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x4 a = {(float)f0[c][0], (float)f0[c][1], (float)f0[c][2], (float)f0[c][3]};
    f32x4 b = {(float)f1[c][0], (float)f1[c][1], (float)f1[c][2], (float)f1[c][3]};
    __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(img0 + ((size_t)s * 3 + c) * plane + o));
    __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(img1 + ((size_t)s * 3 + c) * plane + o));
  }
  f32x4 u = {fu[0], fu[1], fu[2], fu[3]};
  f32x4 v = {fv[0], fv[1], fv[2], fv[3]};
  __builtin_nontemporal_store(u, reinterpret_cast<f32x4*>(flow + ((size_t)s * 2 + 0) * plane + o));
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(flow + ((size_t)s * 2 + 1) * plane + o));
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

// exhaustive probe of the per-byte device formulas (tests): tables of 65536 entries
__global__ void tables_kernel(uint8_t* add_tbl, uint8_t* sub_tbl, uint8_t* aa_tbl, uint8_t* blend_tbl /*256*256 for s=200? no: d,m with s fixed*/, int s_fixed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 65536) return;
  const int u = i >> 8, v = i & 255;
  add_tbl[i] = (uint8_t)comp_add(u, v);
  sub_tbl[i] = (uint8_t)comp_sub(u, v);
  blend_tbl[i] = (uint8_t)blend(u, s_fixed, v);  // d = u, m = v
  if (i < 256) aa_tbl[i] = (uint8_t)aa_byte(i);
}

}  // namespace ofdg
