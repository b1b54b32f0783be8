// Device-side records of the render pipeline (host realize -> geom -> raster ->
// compose).  Plain PODs shared by host code and HIP kernels.
#pragma once
#include <stdint.h>

namespace ofdg {

constexpr int kMaxSegments = 20;    // OFDG_MAX_SEGMENTS
constexpr int kMaxComponents = 8;   // OFDG_MAX_COMPONENTS
constexpr int kMaxVerts = 1024;     // flattened outline capacity per (shape, frame)
constexpr int kCurveMaxPts = 96;    // points one curve3 may flatten to
constexpr int kCurveMaxDepth = 16;  // DFS stack depth for curve3 subdivision
constexpr int kBandRows = 8;        // scanlines one raster workgroup accumulates in LDS
constexpr int kMaxFgObjects = 64;   // foreground objects per sample (bits of a tile mask)
constexpr int kRasterGrid = 512;    // x4 persistent single-wave raster workgroups: enough to finish in time, few enough not to crowd compose

// error bits raised by kernels (device word, read by ofdg_synchronize)
constexpr uint32_t kErrVertCapacity = 1u;   // outline has more than kMaxVerts vertices
constexpr uint32_t kErrCurveCapacity = 2u;  // a curve exceeded kCurveMaxPts / kCurveMaxDepth
constexpr uint32_t kErrDxLimit = 4u;        // an edge spans >= 16384 px (AGG dx_limit)
constexpr uint32_t kErrBgPrepCapacity = 8u; // background_prep: a crop of the rotated image exceeds the workspace (zoom < 0.75)

// 2x3 affine in AGG's member order (sx, shy, shx, sy, tx, ty), fp64.
struct Mat {
  double sx, shy, shx, sy, tx, ty;
};

// One rasterised outline: an ellipse or polygon blueprint (top-level object or a
// component of a composite) with its two outline transforms.
// Reference: MovingObjectEllipse/Polygon::renderMasks, DataGenerator.cpp:465-479, 520-534.
struct DevShape {
  Mat m[2];        // frame 0: intrinsic; frame 1: intrinsic * motion
  float seg_x[kMaxSegments];
  float seg_y[kMaxSegments];
  int32_t seg_type[kMaxSegments];
  float rx, ry;    // ellipse radii
  int32_t type;    // 1 ellipse, 2 polygon
  int32_t n_seg;
  int32_t sample;  // batch slot
  int32_t deform;  // mode 9: frame-1 mask is re-sampled through warp slot `deform-1`
  int32_t object;  // index of the owning DevObject in the batch
  int32_t obj_local;  // index of the owner among its sample's foreground objects (bit of the block masks)
};

// Produced by the geom kernel for each (shape, frame).
struct DevShapeFrame {
  int32_t n_verts;
  int32_t x0, y0, x1, y1;  // pixel bbox clipped to the screen, inclusive; empty: x0 > x1
  int32_t pad[3];
};

// One blitted object (background or top-level foreground object), in z-order.
// Reference: RenderCore::blitObject / getPointFlow, DataGenerator.cpp:762-799, 388-407, 692-718.
struct DevObject {
  Mat motion;         // m_motion (fg: incl. background motion)
  Mat tex_inv;        // inverse of the texture warp transform (getTransformedTexture, :203-205)
  uint64_t tex_base;  // texel offset of the pool image's crop origin
  int32_t first_shape;
  int32_t n_shapes;   // 0 background, 1 simple shape, >=1 composite
  uint32_t additive;  // bit k: component k is additive (composite only)
  int32_t kind;       // 0 background, 1 simple, 2 composite
  int32_t id;
  int32_t deform;     // mode 9: warp slot + 1, 0 = rigid
  int32_t pad[2];
};
static_assert(sizeof(DevObject) == 136, "record layout is read dword-wise by compose");

// The sample record is self-contained for the background: compose reads it with ONE scalar load batch (the
// background's DevObject at objects[first_object] holds the same matrices; the mode-9 kernels use that one).
struct DevSample {
  int32_t first_object;  // background first, then foreground objects by ascending ID
  int32_t n_objects;
  int32_t first_shape;
  int32_t n_shapes;
  Mat bg_motion;         // = objects[first_object].motion
  Mat bg_tex_inv;        // = objects[first_object].tex_inv
  uint64_t bg_tex_base;  // = objects[first_object].tex_base
  int32_t bg_deform;
  int32_t pad;
  // foreground object k (bit k of the block masks): first outline slot relative to first_shape; bit 15 = composite.
  // In the same record as the matrices above: by the time a wave knows its block's objects these lines are in the
  // scalar cache, and the coverage of a simple object can be fetched without first reading the object's own record.
  uint16_t shape_of[kMaxFgObjects];
};
static_assert(sizeof(DevSample) == 256, "compose reads the sample record as 128 + 128 bytes");
constexpr uint16_t kShapeComposite = 0x8000u;

// The tail of a DevObject as compose reads it ahead of a visit (one 32-byte scalar load at offset 96).
struct DevObjectHdr {
  uint64_t tex_base;
  int32_t first_shape, n_shapes;
  uint32_t additive;
  int32_t kind, id, deform;
};

// Background texture preparation of one sample (ofdg_params.background_prep = 1):
// Texture::getRandomizedCrop(2W, 2H, rot, zoom, shift), DG:87-109 - the CImg chain
// get_shift -> rotate -> crop -> resize as ONE resampling along its composed coordinate map.
struct DevBgPrep {
  float ca, sa;            // cos / sin of the rotation (CImg: the angle counts as degrees)
  float w2, h2, rw2, rh2;  // centres of the pool image and of the (grown) rotated image
  float fx, fy;            // resize step: crop columns / rows per prepared texel
  int32_t x0, y0;          // crop origin in the rotated image
  int32_t cw, ch;          // crop size
  int32_t shx, shy;        // get_shift offsets
  uint64_t image_addr;     // device address of the pool image (BGRX texels, pw x ph)
  int32_t rx0, ry0, rx1, ry1;  // texels of the 2W x 2H texture compose can read (inclusive); the rest is not rendered
  int32_t rw, rh;              // size of the rotated image (crop coordinates are mirrored into it)
  int32_t pw, ph;              // size of the pool image (texture lists may hold images of different sizes)
};
// Where a whole pool image lives (background preparation reads the original image): the table of a pool with images of
// different sizes; a uniform pool needs none (image i = base + i * w * h).
struct DevTexEntry {
  uint64_t addr;
  int32_t w, h;
};

// Where an object's texture lives relative to a pool pointer: image i starts at i * stride, its
// W x H (foreground) or 2W x 2H (background) window at + origin, rows are `pitch` texels apart.
// Pool images at least as large as the window: the centre crop of the image itself
// (getRandomizedCrop, DG:96-101); smaller images: a pool of resized copies (DG:102-106).
struct TexSource {
  uint64_t stride, origin;
  int32_t pitch, pad;
};

// Pointers the kernels read out of records are typed as GLOBAL memory in device code (a pointer loaded from memory is
// generic otherwise: flat_load, which also counts on lgkmcnt and so serialises with every scalar wait).
// OFDG_CONSTANT: memory no kernel writes while it runs; a uniform load from it is a scalar load even behind the kernel's own
// stores (which the compiler must otherwise assume could alias it).
#if defined(__HIP_DEVICE_COMPILE__)
#define OFDG_GLOBAL __attribute__((address_space(1)))
#define OFDG_CONSTANT __attribute__((address_space(4)))
#else
#define OFDG_GLOBAL
#define OFDG_CONSTANT
#endif
// One served warp crop as the kernels see it (mode 9).
struct DevCropRef {
  OFDG_GLOBAL const float* data;         // two planes of w*h float PAIRS: (flow x, flow y), then (iflow x, iflow y)
  OFDG_GLOBAL const unsigned* max_bits;  // float bits of max |iflow| over the crop (NaNs ignored)
  int32_t w, h;
};
inline DevCropRef make_crop_ref(const float* data, const unsigned* max_bits, int w, int h) {
  DevCropRef r;
  r.data = (OFDG_GLOBAL const float*)data; r.max_bits = (OFDG_GLOBAL const unsigned*)max_bits; r.w = w; r.h = h;
  return r;
}

struct RenderDims {
  int32_t W, H;            // output size
  int32_t pool_w, pool_h;  // pool image size (texels are BGRX u32)
  int32_t use_aa;
  int32_t n_samples;
  int32_t n_shapes;        // total rasterised shapes in the batch
  int32_t tiles_x, tiles_y;
  int32_t bg_pitch;        // row pitch (texels) of the background textures: pool_w, or 2W when they are prepared per sample / resized
  int32_t fg_pitch;        // row pitch of the foreground textures: pool_w, or W when the pool images are smaller than W x H
};

}  // namespace ofdg
