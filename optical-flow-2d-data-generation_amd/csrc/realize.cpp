#include "realize.h"

namespace ofdg {
namespace {

struct Motion {
  Mat intrinsic, motion;
};

// setIntrinsicTransform + setMotion + addBackgroundMotion (DataGenerator.cpp:302-335)
Motion object_motion(const ofdg_blueprint& p, const Mat& bg_motion, int W, int H) {
  Motion r;
  r.intrinsic = mat_identity();
  r.intrinsic = mat_mul(r.intrinsic, mat_rotation(p.init_rot));
  r.intrinsic = mat_mul(r.intrinsic, mat_translation(p.init_trans_x, p.init_trans_y));
  r.motion = mat_identity();
  r.motion = mat_mul(r.motion, mat_rotation(p.rot));
  r.motion = mat_mul(r.motion, mat_scaling(p.scale));
  r.motion = mat_mul(r.motion, mat_translation(p.trans_x, p.trans_y));
  Mat bg_n = mat_translation(-W / 2., -H / 2.);
  bg_n = mat_mul(bg_n, bg_motion);
  bg_n = mat_mul(bg_n, mat_translation(W / 2., H / 2.));
  r.motion = mat_mul(r.motion, bg_n);
  return r;
}

}  // namespace

// CImg 2.x: rotate() takes degrees and grows the image to round(1 + |(w-1)cos| + |(h-1)sin|);
// crop(x0, y0, x1, y1) with float -> int truncation of x1 = x0 + 2W/zoom - 1; linear
// get_resize with boundary 0 steps (w - 1)/(sx - 1) when enlarging, w/sx otherwise.
DevBgPrep make_bg_prep(int pw, int ph, int W, int H, float angle, float zoom, int shift_x, int shift_y, uint64_t image_addr) {
  DevBgPrep p;
  const int TW = 2 * W, TH = 2 * H;
  const float nangle = (float)(angle - 360.0f * std::floor((double)angle / 360.0f));
  const float rad = (float)(nangle * 3.14159265358979323846 / 180.0);
  p.ca = std::cos(rad);
  p.sa = std::sin(rad);
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
  p.fx = TW > p.cw ? (float)((p.cw - 1.0) / (TW - 1.0)) : (float)((double)p.cw / TW);
  p.fy = TH > p.ch ? (float)((p.ch - 1.0) / (TH - 1.0)) : (float)((double)p.ch / TH);
  p.shx = shift_x; p.shy = shift_y;
  p.image_addr = image_addr;
  p.pw = pw; p.ph = ph;
  p.rx0 = 0; p.ry0 = 0; p.rx1 = TW - 1; p.ry1 = TH - 1;
  return p;
}

namespace {

int push_shape(const RealizeConfig& cfg, const ofdg_blueprint& p, const Mat& bg_motion, int sample, int object, int obj_local, int deform,
               std::vector<DevShape>* shapes, std::string* msg) {
  if (p.obj_type != OFDG_OBJ_ELLIPSE && p.obj_type != OFDG_OBJ_POLYGON) {
    *msg = "(RealizeObjectBlueprint) Bad object type, or not intended in this mode";  // DataGenerator.cpp:1143
    return OFDG_EOBJTYPE;
  }
  DevShape s = DevShape();
  const Motion m = object_motion(p, bg_motion, cfg.W, cfg.H);
  s.m[0] = m.intrinsic;
  s.m[1] = mat_mul(m.intrinsic, m.motion);  // "save = intrinsic; save *= motion" (:469-470, 522-523)
  s.type = p.obj_type;
  s.rx = p.ellipse_scale_x;
  s.ry = p.ellipse_scale_y;
  s.sample = sample;
  s.object = object;
  s.obj_local = obj_local;
  s.deform = deform;
  if (p.obj_type == OFDG_OBJ_POLYGON) {
    if (p.n_segments < 1 || p.n_segments > kMaxSegments) {
      *msg = "polygon blueprint with a segment count outside [1, 20]";
      return OFDG_ECAPACITY;
    }
    s.n_seg = p.n_segments;
    for (int i = 0; i < p.n_segments; ++i) {
      s.seg_x[i] = p.segment_x[i];
      s.seg_y[i] = p.segment_y[i];
      s.seg_type[i] = p.segment_type[i];
      if (i > 0 && p.segment_type[i] == OFDG_SEG_DUMMY &&
          !(p.segment_type[i - 1] == OFDG_SEG_CURVE3)) {
        *msg = "PolySegmentType_t::Dummy found, this should have been skipped!";  // DataGenerator.cpp:1096
        return OFDG_EOBJTYPE;
      }
      if (p.segment_type[i] == OFDG_SEG_CURVE3 && i + 1 >= p.n_segments) {
        *msg = "Curve3 segment without an end point";
        return OFDG_EOBJTYPE;
      }
    }
  }
  shapes->push_back(s);
  return OFDG_OK;
}

}  // namespace

int realize_batch(const RealizeConfig& cfg, const ofdg_task* tasks, int n_tasks, const ofdg_blueprint* bps,
                  int n_bps, RealizedBatch* out, std::string* msg, CropServer* crops) {
  out->shapes.clear();
  out->objects.clear();
  out->samples.clear();
  out->crops.clear();
  out->bgprep.clear();
  const bool mode9 = (cfg.mode == 9);
  auto serve = [&](bool background) -> int {  // returns deform = table index + 1
    out->crops.push_back(CropUse{crops->get(), background ? 1 : 0});
    return (int)out->crops.size();
  };
  if (mode9 && (!crops || crops->n_crops <= 0)) {
    for (int t = 0; t < n_tasks; ++t) {
      bool need = bps[tasks[t].background].do_warpfield_deformation != 0;
      for (int i = 0; i < tasks[t].n_objects && !need; ++i) need = bps[tasks[t].first_object + i].do_warpfield_deformation != 0;
      if (need) { *msg = "mode 9 needs warp fields: call ofdg_warp_generate or ofdg_warp_upload first"; return OFDG_EINVAL; }
    }
  }
  const int W = cfg.W, H = cfg.H;
  // fg: Texture::getRandomizedCrop() with defaults == exact centre W x H crop
  // (DataGenerator.cpp:87-109 via :1149-1150); bg: 2W x 2H centre crop (parity boundary:
  // tex_rot / tex_scale / tex_shift preparation is not applied).
  const uint64_t img_texels = (uint64_t)cfg.pool_w * cfg.pool_h;
  const uint64_t fg_stride = cfg.fg_stride ? cfg.fg_stride : img_texels, bg_stride = cfg.bg_stride ? cfg.bg_stride : img_texels;
  const uint64_t fg_origin = cfg.fg_stride ? cfg.fg_origin : (uint64_t)(cfg.pool_h / 2 - H / 2) * cfg.pool_w + (cfg.pool_w / 2 - W / 2);
  const uint64_t bg_origin = cfg.bg_stride ? cfg.bg_origin : (uint64_t)(cfg.pool_h / 2 - H) * cfg.pool_w + (cfg.pool_w / 2 - W);
  for (int t = 0; t < n_tasks; ++t) {
    const ofdg_task& task = tasks[t];
    if (task.background < 0 || task.background >= n_bps || task.n_objects < 0 ||
        task.first_object < 0 || task.first_object + task.n_objects > n_bps) {
      *msg = "task refers to blueprints outside the array";
      return OFDG_EINVAL;
    }
    if (task.n_objects > kMaxFgObjects) {
      *msg = "more than 64 foreground objects in one sample";
      return OFDG_ECAPACITY;
    }
    DevSample smp = DevSample();
    smp.first_object = (int32_t)out->objects.size();
    smp.first_shape = (int32_t)out->shapes.size();
    // background (DataGenerator.cpp:1183-1205, 654-663)
    const ofdg_blueprint& pb = bps[task.background];
    Mat bg_motion = mat_identity();
    bg_motion = mat_mul(bg_motion, mat_rotation(pb.rot));
    bg_motion = mat_mul(bg_motion, mat_scaling(pb.scale));
    bg_motion = mat_mul(bg_motion, mat_translation(pb.trans_x, pb.trans_y));
    {
      DevObject o = DevObject();
      o.kind = 0;
      o.id = pb.obj_id;
      o.motion = bg_motion;
      // setIntrinsicTransform(0.f, W, H): rotation(0) * translation(W, H)
      Mat intrinsic = mat_mul(mat_mul(mat_identity(), mat_rotation(0.f)), mat_translation(W, H));
      Mat intrinsic_inv = mat_invert(intrinsic);
      // m_intrinsic_transform_inv * m_motion * m_intrinsic_transform (:673, 677), inverted (:203-205)
      Mat warp = mat_mul(mat_mul(intrinsic_inv, bg_motion), intrinsic);
      o.tex_inv = mat_invert(warp);
      if (cfg.background_prep) {
        // the sample's own prepared 2W x 2H texture (bgprep_kernel), in the slot's buffer
        const int ti = pb.tex_id % cfg.pool_n;
        DevBgPrep bp = cfg.tex_table
            ? make_bg_prep(cfg.tex_table[ti].w, cfg.tex_table[ti].h, W, H, pb.tex_rot, pb.tex_scale, pb.tex_shift_x, pb.tex_shift_y, cfg.tex_table[ti].addr)
            : make_bg_prep(cfg.pool_w, cfg.pool_h, W, H, pb.tex_rot, pb.tex_scale, pb.tex_shift_x, pb.tex_shift_y,
                           cfg.pool_addr + (uint64_t)ti * img_texels * sizeof(uint32_t));
        // (mode 9 re-samples the background through a warp field: anything may be read)
        if (!(mode9 && pb.do_warpfield_deformation)) bg_prep_region(o.tex_inv, W, H, &bp.rx0, &bp.ry0, &bp.rx1, &bp.ry1);
        out->bgprep.push_back(bp);
        o.tex_base = (uint64_t)t * 4ull * (uint64_t)W * (uint64_t)H;
      } else {
        o.tex_base = (uint64_t)(pb.tex_id % cfg.pool_n) * bg_stride + bg_origin;
      }
      o.first_shape = 0;
      o.n_shapes = 0;
      if (mode9 && pb.do_warpfield_deformation) o.deform = serve(true);  // DataGenerator.cpp:1194-1202
      smp.bg_motion = o.motion; smp.bg_tex_inv = o.tex_inv; smp.bg_tex_base = o.tex_base; smp.bg_deform = o.deform; smp.pad = 0;
      out->objects.push_back(o);
    }
    // foreground objects; std::map order == ascending obj_id (DataGenerator.cpp:1216-1223)
    std::vector<int> order(task.n_objects);
    for (int i = 0; i < task.n_objects; ++i) order[i] = task.first_object + i;
    for (size_t i = 1; i < order.size(); ++i)  // insertion sort by id (already sorted when sampled)
      for (size_t j = i; j > 0 && bps[order[j]].obj_id < bps[order[j - 1]].obj_id; --j) std::swap(order[j], order[j - 1]);
    for (int bi : order) {
      const ofdg_blueprint& p = bps[bi];
      if (p.obj_id <= OFDG_BACKGROUND_ID) {
        *msg = "foreground object id must be > 1 (the background owns id 1)";
        return OFDG_EINVAL;
      }
      DevObject o = DevObject();
      o.id = p.obj_id;
      const Motion m = object_motion(p, bg_motion, W, H);
      o.motion = m.motion;
      o.tex_inv = mat_invert(m.motion);
      o.tex_base = (uint64_t)(p.tex_id % cfg.pool_n) * fg_stride + fg_origin;
      o.first_shape = (int32_t)out->shapes.size();
      if (p.obj_type == OFDG_OBJ_COMPOSITE) {
        if (p.n_components < 1 || p.n_components > kMaxComponents || p.first_component < 0 ||
            p.first_component + p.n_components > n_bps) {
          *msg = "composite blueprint with a component range outside [1, 8] / the array";
          return OFDG_ECAPACITY;
        }
        o.kind = 2;
        o.n_shapes = p.n_components;
        // the composite takes its crop before its components, which copy it (DataGenerator.cpp:1120-1128, 1158-1163)
        if (mode9 && p.do_warpfield_deformation) o.deform = serve(false);
        for (int k = 0; k < p.n_components; ++k) {
          const ofdg_blueprint& c = bps[p.first_component + k];
          const int cdef = (mode9 && c.do_warpfield_deformation) ? o.deform : 0;
          int rc = push_shape(cfg, c, bg_motion, t, (int)out->objects.size(), (int)out->objects.size() - smp.first_object - 1, cdef, &out->shapes, msg);
          if (rc != OFDG_OK) return rc;
          if (c.is_additive_component) o.additive |= (1u << k);
        }
      } else {
        o.kind = 1;
        o.n_shapes = 1;
        if (mode9 && p.do_warpfield_deformation) o.deform = serve(false);  // DataGenerator.cpp:1164-1168
        int rc = push_shape(cfg, p, bg_motion, t, (int)out->objects.size(), (int)out->objects.size() - smp.first_object - 1, o.deform, &out->shapes, msg);
        if (rc != OFDG_OK) return rc;
      }
      {
        const int local = (int)out->objects.size() - smp.first_object - 1;  // bit of the block masks
        smp.shape_of[local] = (uint16_t)((o.first_shape - smp.first_shape) | (o.kind == 2 ? kShapeComposite : 0));
      }
      out->objects.push_back(o);
    }
    smp.n_objects = (int32_t)out->objects.size() - smp.first_object;
    smp.n_shapes = (int32_t)out->shapes.size() - smp.first_shape;
    out->samples.push_back(smp);
  }
  return OFDG_OK;
}

}  // namespace ofdg
