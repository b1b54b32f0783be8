// C-ABI of the MI355X-native optical-flow data generator (include/ofdg.h).
// Owns the HIP context state (texture pool, workspaces) and launches the kernels
// in kernels.hip.  There is no CPU fallback: without a HIP device every entry
// point that would render fails with OFDG_EHIP.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ofdg.h"
#include "kernels.hip"
#include "sampler_counter.hip"
#include "ofdg_device.h"
#include "realize.h"
#include "sampler_ref.h"
#include "warpfields.h"

using namespace ofdg;

static thread_local std::string g_create_error;
// GPU_MAX_HW_QUEUES as the process environment had it when this library was loaded: HIP reads the variable once, when the
// runtime starts, so what counts is the environment the process was started with
static const int g_hw_queues_env = [] { const char* q = std::getenv("GPU_MAX_HW_QUEUES"); return q ? std::atoi(q) : 0; }();

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  bool view = false;  // p points into another allocation (the record arena of an uploaded batch): not ours to free
  hipError_t reserve(size_t n) {
    if (n <= cap && !view) return hipSuccess;
    if (p && !view) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; }
    p = nullptr; cap = 0; view = false;
    size_t want = n + n / 4 + 16;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  // look at n elements of somebody else's memory (what we own is given up first)
  hipError_t alias(T* q, size_t n) {
    if (p && !view) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; }
    p = q; cap = n; view = true;
    return hipSuccess;
  }
  void release() { if (p && !view) (void)hipFree(p); p = nullptr; cap = 0; view = false; }
};

struct ofdg_ctx {
  ofdg_params prm;
  std::string err;
  std::string info;
  uint32_t* d_prep_paths = nullptr;  // diagnostics: tiles of bgprep_stream_kernel by form (ofdg_debug_bgprep_paths switches it on)
  // texture pool
  uint32_t* pool = nullptr;
  int pool_n = 0, pool_w = 0, pool_h = 0;
  // where foreground (W x H) and background (2W x 2H) textures are read from: the pool images
  // themselves (centre crops) or pools of resized copies when the images are smaller (DG:96-106)
  uint32_t* pool_fg = nullptr;   // [n][H][W] if the images are smaller than W x H
  uint32_t* pool_bg = nullptr;   // [n][2H][2W] if the images are smaller than 2W x 2H
  TexSource fg_src{}, bg_src{};
  bool pool_final = false;       // derived pools match the current pool contents
  bool pool_mixed = false;       // ofdg_pool_alloc_mixed: only the derived pools exist (images of different sizes)
  std::vector<std::pair<int, int>> mixed_sizes;  // source image sizes of a mixed pool (index table)
  // background_prep on a mixed pool: the whole images stay resident (getRandomizedCrop works on the original image)
  std::vector<uint32_t*> mixed_images;
  std::vector<DevTexEntry> tex_table;   // host copy of ...
  DevTexEntry* d_tex_table = nullptr;   // ... the per-image table the device sampler reads
  int pool_kind = OFDG_POOL_UNIFORM;
  uint32_t pool_seed = 0;
  // sampler
  std::unique_ptr<RefSampler> sampler;
  long long step = 0;
  // host staging (pinned) + device records.  A "slot" is one realised batch resident
  // in HBM (shapes, objects, samples, outlines); slot 0 is what ofdg_render uses, the
  // others let a caller keep several batches in flight (data_param.prefetch).
  struct Slot {
    RealizedBatch batch;
    DevBuf<DevShape> d_shapes;
    DevBuf<DevShapeFrame> d_frames;
    DevBuf<int2> d_verts;
    DevBuf<DevObject> d_objects;
    DevBuf<DevSample> d_samples;
    DevBuf<char> d_rec;  // uploaded batches: shapes | objects | samples in ONE allocation (one copy per batch; the three above look into it)
    DevBuf<int4> d_items;
    DevBuf<DevBgPrep> d_bgprep;   // background_prep: one record ...
    DevBuf<uint32_t> d_bgtex;     // ... and one prepared 2W x 2H BGRX texture per sample
    // background_prep = 1 (the CImg chain stage by stage): crop of the rotated image, X-resized image, resize tables, plan
    DevBuf<uint32_t> d_bgC;
    DevBuf<unsigned long long> d_blockmask;  // [2 parities][samples][64 x 8 blocks][2 frames]
    int res_objects = 0;
    bool bgprep_pending = false;  // background_prep: the records are in d_bgprep, the textures are rendered by the next launch_prepare
    int box_parity = 0;
    size_t box_stride = 0;   // mask words per parity
    int mask_used[2] = {0, 0};  // words the last launch on each parity marked (what the next clear must cover)
    hipEvent_t ev_uploaded = nullptr;
    bool upload_pending = false;
    hipStream_t upload_stream = nullptr;   // where the records were (last) written
    DevBuf<DevCropRef> d_croptab;      // mode 9: crops of this batch's deforming objects
    DevBuf<float> d_bgwarp;            // mode 9: upscaled (2W x 2H) background crops
    DevBuf<unsigned> d_bgwarp_max;
    hipEvent_t compose_event = nullptr;  // last tracked compose that read this slot's records (alias of a chain's ev_done) ...
    bool compose_pending = false;
    hipStream_t compose_stream = nullptr;  // ... and the stream it ran on
    int* d_item_count = nullptr;
    int res_samples = 0, res_shapes = 0;
  };
  static constexpr int kUserSlots = 16;              // ofdg_upload_slot / ofdg_render_slot
  Slot slots[kUserSlots];
  // pinned staging of one batch's records on their way to the device
  struct Stage {
    void* h = nullptr;
    size_t bytes = 0;
    hipEvent_t free_ev = nullptr;
    bool pending = false;
  };
  Stage user_stage;  // ofdg_upload_slot
  // The pipeline: independent IN-ORDER chains.  One call = one chain (round robin); the chain's
  // stream runs [record upload | counter sampler] -> geom -> raster -> compose back to back, with no
  // cross-stream hand-over between them (an event wait in front of a kernel exposes ~15 us of
  // dispatch latency per step; kernels of one stream follow each other without a gap).  The
  // latency-bound preparation kernels of one chain overlap the compose kernels of the others.
  // Each chain owns its coverage workspace, a private record slot (ofdg_render / ofdg_forward*)
  // and its staging buffer, so chains share nothing that is written per call.
  struct Chain {
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> cov;
    Slot slot;
    Stage stage;
    hipEvent_t ev_prep = nullptr;  // coverage ready (hand-over to a caller's stream)
    hipEvent_t ev_done = nullptr;  // the chain's last tracked compose ...
    bool done_pending = false;
    hipStream_t done_stream = nullptr;  // ... and the stream it ran on (the chain's own, or a caller's)
    // A batch whose preparation kernels have been enqueued on this chain and whose compose has not: what launch_prepare
    // hands to launch_compose (and, for the counter sampler, which samples it holds: the look-ahead of ofdg_forward_counter)
    struct Prepared {
      bool valid = false;
      Slot* slot = nullptr;
      long long first_index = -1;  // counter sampler: global index of the batch's first sample (-1: host blueprints)
      int n = 0;
      unsigned long long* box_cur = nullptr;
      const DevCropRef* croptab = nullptr;
      hipEvent_t* ev = nullptr;    // profiled launch: its event set
      hipStream_t stream = nullptr;  // where the preparation was enqueued
      long long ticket = -1;       // the batch's number (its error word)
      bool ahead = false;          // prepared by an EARLIER call (look-ahead): the end of its preparation says nothing about when compose could start
    } prep;
  };
  static constexpr int kMaxChains = 8;
  Chain chains[kMaxChains];
  int lookahead = 0;      // counter sampler: batches prepared ahead of the call that composes them (0: none)
  long long last_first = -1;  // first index of the previous ofdg_forward_counter call (the stride of the caller's sequence)
  int n_chains = 3;       // one hardware queue each: HIP maps streams onto GPU_MAX_HW_QUEUES (4 by default) queues
  unsigned next_chain = 0;
  int last_chain = 0;     // the chain and the slot the last launch used (ofdg_render_resident, debug read-back)
  Slot* last_slot = nullptr;
  hipStream_t last_user_st = nullptr;  // serial mode: the caller's stream of the last launch
  bool have_last_user_st = false;
  // device counter sampler (OFDG_SAMPLER_COUNTER)
  CsMode cs_mode;
  DevBuf<ofdg_blueprint> d_cs_bps;
  DevBuf<int> d_cs_nobj;
  long long next_index = 0;  // next global sample index of this rank's stream
  long long last_ticket = -1;  // ticket of the batch the last render / forward call composed
  // mode 9: served warp crops, each 4 planes of (W+1)*(H+1) floats, contiguous
  float* d_warp = nullptr;         // [n_crops][2 pairs][(H+1)][(W+1)][2]: (flow x, flow y), (iflow x, iflow y) interleaved
  unsigned* d_warp_max = nullptr;  // [n_crops] float bits of max |iflow|
  // counter sampler, mode 9: every crop as the kernels see it, [k] the crop itself (foreground), [n_crops + k] its
  // 2W x 2H upscaled copy (backgrounds); built once per set of crops
  DevCropRef* d_cs_croptab = nullptr;
  float* d_cs_bgwarp = nullptr;
  unsigned* d_cs_bgwarp_max = nullptr;
  CropServer crop_server;
  int rs_w = 0, rs_h = 0;          // CImg resize tables for the background crops
  int *d_rs_xi = nullptr, *d_rs_yi = nullptr;
  double *d_rs_xa = nullptr, *d_rs_ya = nullptr;
  bool overlap = true;
  double* d_cs_tab = nullptr;
  // Device error flags, ONE WORD PER CALL: call number q ("ticket") raises its flags in word q mod kErrWords, so that a prefetch
  // ring can ask at a batch's hand-over whether THAT batch was truncated (ofdg_poll_errors_of) - the reference drops a bad
  // sample silently (DG:1285-1292).  A word is cleared when it is read; one that was never asked for is read by
  // ofdg_synchronize / ofdg_poll_errors.  For a caller that asks by ticket (a prefetch ring) a word must not carry what an
  // EARLIER owner left in it - a look-ahead preparation that was discarded, a batch the caller never asked about: once
  // anybody has asked by ticket, a word that was not read since its last use is cleared on the chain's stream in front of
  // the first kernel of the call that takes it over (`word_clean`, launch_prepare).  A caller that only ever uses the
  // device-wide forms (bench.py) pays nothing.  Word kErrWords belongs to the debug entry points.
  static constexpr int kErrWords = 256;
  uint32_t* d_err = nullptr;     // [kErrWords + 1]
  bool asked_by_ticket = false;  // ofdg_poll_errors_of has been called on this context
  bool word_clean[kErrWords];    // the word holds nothing of a call before the one that owns it now (all true at creation)
  long long word_reserved = -1;  // the ticket whose word an upload's background preparation already writes into (ofdg_upload_slot)
  long long ticket = 0;          // calls made so far = the ticket of the next call
  // background_prep = 1: CImg's enlarging tables for every source length below 2W (x) / 2H (y), tabulated once
  DevBuf<uint16_t> d_bg_at_x, d_bg_at_y;
  DevBuf<double> d_bg_alpha_x, d_bg_alpha_y;
  int bg_tab_w = 0, bg_tab_h = 0;
  int bg_cap_cw = 0, bg_cap_ch = 0, bg_cap_n = -1;  // bgprep_caps of the current pool (reset when the pool changes)
  bool bg_fusable = false;
  uint32_t* h_err = nullptr;        // pinned copy for ofdg_poll_errors, on its own stream
  hipStream_t err_stream = nullptr;
  // profiling: ring of event sets, 6 events per launch: start/stop of geom, raster and compose,
  // attached to the kernels' own dispatch packets
  int profiling = 0;  // 0 off, 1 compose kernel only, 2 all three kernels
  std::vector<hipEvent_t> ev;
  int ev_sets = 0, ev_stride = 1;
  long long ev_count = 0, ev_alloc = 0, launch_count = 0;  // event sets: composed / handed out (a prepared batch holds one)
  std::vector<char> ev_composed;  // per set: its compose was enqueued (a prepared batch that is discarded leaves its set incomplete)
  std::vector<char> ev_bgprep;    // per set: the batch's background preparation ran behind raster and is timed (ev[3] .. ev[2] or ev[4])
  std::vector<ofdg_task> fw_tasks;
  std::vector<ofdg_blueprint> fw_bps;
};

#define HIP_OK(ctx, call)                                                                     \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
      return OFDG_EHIP;                                                                       \
    }                                                                                         \
  } while (0)

// the error word of the call being made (its ticket is taken - c->ticket advanced - when the call's first kernel is enqueued)
static uint32_t* err_word(ofdg_ctx* c, long long ticket) { return c->d_err + (size_t)(ticket % ofdg_ctx::kErrWords); }
// The call with this ticket is about to enqueue its first kernel on `s`: what an earlier owner left in its word goes first
// (only for callers that ask by ticket; see ofdg_ctx::word_clean).
static hipError_t take_err_word(ofdg_ctx* c, long long ticket, hipStream_t s) {
  const size_t i = (size_t)(ticket % ofdg_ctx::kErrWords);
  hipError_t e = hipSuccess;
  if (c->word_reserved == ticket) c->word_reserved = -1;  // (an upload for this very call already raised its flags here: they are this call's)
  else if (c->asked_by_ticket && !c->word_clean[i]) e = hipMemsetAsync(c->d_err + i, 0, sizeof(uint32_t), s);
  c->word_clean[i] = false;
  return e;
}
static std::string err_text(uint32_t e) {
  std::string t = "device capacity exceeded:";
  if (e & kErrVertCapacity) t += " outline vertices > 1024;";
  if (e & kErrCurveCapacity) t += " curve3 subdivision points/depth;";
  if (e & kErrDxLimit) t += " edge spans >= 16384 px;";
  if (e & kErrBgPrepCapacity) t += " background_prep: crop of the rotated image exceeds the workspace (zoom < 0.75);";
  return t;
}

static void drop_counter_croptab(ofdg_ctx* c);
static int discard_all_prepared(ofdg_ctx* c);
static int texture_of_image(ofdg_ctx* c, const uint32_t* image, int w, int h, int tw, int th, uint32_t* out);

extern "C" {

void ofdg_default_params(ofdg_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->width = 512;               // DGEN_WIDTH
  p->height = 384;              // DGEN_HEIGHT
  p->mode = 1;                  // caffe.proto:7
  p->use_antialiasing = 1;      // caffe.proto:11
  p->batch_size = 1;
  p->prefetch = 1;
  p->first_level_threads = 16;  // caffe.proto:9
  p->second_level_threads = 1;  // caffe.proto:10
  p->sampler = OFDG_SAMPLER_REF;
  p->world_size = 1;
}

const char* ofdg_last_error(const ofdg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int ofdg_create(const ofdg_params* params, ofdg_ctx** out) {
  if (!params || !out) { g_create_error = "null argument"; return OFDG_EINVAL; }
  *out = nullptr;
  if (params->mode < 1 || params->mode > 13) { g_create_error = "BAD MODE"; return OFDG_EBADMODE; }
  if (params->width < 8 || params->height < 2 || (params->width % 8) != 0 || (params->height % 2) != 0) {
    g_create_error = "width must be a multiple of 8 and height even";
    return OFDG_EINVAL;
  }
  {  // a sample holds at most 64 foreground objects (bits of a block mask); the device sampler generates at most 32
    const int cap = params->sampler == OFDG_SAMPLER_COUNTER ? kCsMaxObjects : kMaxFgObjects;
    if (params->num_objects < 0 || params->num_objects > cap) {
      g_create_error = "num_objects must be 0 (reference: 16..23) or 1.." + std::to_string(cap) + " for this sampler";
      return OFDG_EINVAL;
    }
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device available (the HIP kernels are the only render path)";
    return OFDG_EHIP;
  }
  e = hipSetDevice(params->device);
  if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return OFDG_EHIP; }
  std::unique_ptr<ofdg_ctx> c(new ofdg_ctx());
  std::fill(c->word_clean, c->word_clean + ofdg_ctx::kErrWords, true);
  c->prm = *params;
  if (c->prm.world_size < 1) c->prm.world_size = 1;
  c->sampler.reset(new RefSampler(params->mode, params->width, params->height, params->num_objects));
  if (!c->sampler->ok()) { g_create_error = "BAD MODE"; return OFDG_EBADMODE; }
  {  // constants of the device counter sampler: the 13 mode tables as deltas to mode 7 (DG:1363-2001)
    CsMode& M = c->cs_mode;
    const double pi = 3.14159265358979323846;
    struct Mag { double bg_rot, bg_trans, bg_s0, bg_s1, obj_trans, obj_rot, obj_s0, obj_s1, t_bgr, t_bgs, t_or, t_os; };
    Mag g{10, 40, 0.93, 1.07, 120, 30, 0.8, 1.2, 0.3, 0.6, 0.7, 0.7};
    const int mode = params->mode;
    if (mode == 10) g = Mag{5, 20, 0.965, 1.035, 60, 15, 0.9, 1.1, 0.176, 0.429, 0.539, 0.539};
    if (mode == 11) g = Mag{20, 80, 0.86, 1.14, 240, 60, 0.6, 1.4, 0.462, 0.75, 0.824, 0.824};
    if (mode == 12) g = Mag{3.3, 13.3, 0.976, 1.023, 40, 10, 0.933, 1.066, 0.125, 0.333, 0.437, 0.437};
    if (mode == 13) g = Mag{30, 120, 0.79, 1.21, 360, 90, 0.4, 1.6, 0.563, 0.818, 0.875, 0.875};
    const bool tonly = (mode == 1 || mode == 2 || mode == 3 || mode == 8);
    const bool rot = !tonly, scl = !tonly && mode != 4;
    M.bg_rot_a = rot ? (float)(-g.bg_rot * pi / 180.) : 0.f; M.bg_rot_b = rot ? (float)(g.bg_rot * pi / 180.) : 0.f;
    M.bg_trans = (float)g.bg_trans;
    M.bg_scale_a = scl ? (float)g.bg_s0 : 1.f; M.bg_scale_b = scl ? (float)g.bg_s1 : 1.f;
    M.t_bg_rot = rot ? (float)g.t_bgr : -1.f; M.t_bg_scale = scl ? (float)g.t_bgs : -1.f;
    M.t_obj_rot = rot ? (float)g.t_or : -1.f; M.t_obj_scale = scl ? (float)g.t_os : -1.f;
    M.obj_trans = (float)g.obj_trans;
    M.obj_rot_a = rot ? (float)(-g.obj_rot * pi / 180.) : 0.f; M.obj_rot_b = rot ? (float)(g.obj_rot * pi / 180.) : 0.f;
    M.obj_scale_a = scl ? (float)g.obj_s0 : 1.f; M.obj_scale_b = scl ? (float)g.obj_s1 : 1.f;
    M.init_rot_a = mode == 1 ? 0.f : (float)-pi; M.init_rot_b = mode == 1 ? 0.f : (float)pi;
    M.deform_thr = mode == 9 ? 0.2f : 0.f;
    M.type_mask = (mode == 1 || mode == 2) ? 2 : mode == 3 ? 1 : (mode == 4 || mode == 5 || mode == 8) ? 3 : 7;
    M.n_types = 0;
    if (M.type_mask & 1) M.types[M.n_types++] = OFDG_OBJ_ELLIPSE;
    if (M.type_mask & 2) M.types[M.n_types++] = OFDG_OBJ_POLYGON;
    if (M.type_mask & 4) M.types[M.n_types++] = OFDG_OBJ_COMPOSITE;
    M.mode = mode; M.W = params->width; M.H = params->height; M.num_objects = params->num_objects;
    M.seed = (uint32_t)params->seed;
  }
  // cos/sin of agg::ellipse's 100 step angles, from the host libm
  double tab[200];
  const double pi = 3.14159265358979323846;
  for (int step = 0; step < 100; ++step) {
    const double angle = double(step) / double(100) * 2.0 * pi;
    tab[2 * step] = std::cos(angle);
    tab[2 * step + 1] = std::sin(angle);
  }
  if ((e = hipMalloc((void**)&c->d_cs_tab, sizeof(tab))) != hipSuccess ||
      (e = hipMemcpy(c->d_cs_tab, tab, sizeof(tab), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMalloc((void**)&c->d_err, (ofdg_ctx::kErrWords + 1) * sizeof(uint32_t))) != hipSuccess ||
      (e = hipMemset(c->d_err, 0, (ofdg_ctx::kErrWords + 1) * sizeof(uint32_t))) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->user_stage.free_ev, hipEventDisableTiming)) != hipSuccess) {
    g_create_error = std::string("HIP initialisation: ") + hipGetErrorString(e);
    return OFDG_EHIP;
  }
  c->overlap = params->serial == 0;  // serial: everything on the caller's stream
  c->lookahead = std::max(params->lookahead, 0);
  // One hardware queue per chain: HIP maps its streams onto GPU_MAX_HW_QUEUES (default 4) queues.  A process started
  // with GPU_MAX_HW_QUEUES >= 8 gets four chains (+5-8 % throughput, profiles/r02_chains_vs_hw_queues.txt); four chains
  // on four queues share a queue with the caller's streams and are slower than three.
  c->n_chains = params->chains > 0 ? std::min(params->chains, (int)ofdg_ctx::kMaxChains) : (g_hw_queues_env >= 8 ? 4 : 3);
  c->info = "chains=" + std::to_string(c->n_chains) +
            (params->chains > 0 ? " (ofdg_params.chains)"
             : g_hw_queues_env >= 8 ? " (GPU_MAX_HW_QUEUES=" + std::to_string(g_hw_queues_env) + ": one hardware queue per chain)"
             : g_hw_queues_env > 0 ? " (GPU_MAX_HW_QUEUES=" + std::to_string(g_hw_queues_env) + " < 8: start the process with GPU_MAX_HW_QUEUES=8 for four chains, +5-8 %)"
                                   : " (GPU_MAX_HW_QUEUES was not set when the library was loaded: HIP's default of 4 hardware queues; start the process with GPU_MAX_HW_QUEUES=8 for four chains, +5-8 %)") +
            " lookahead=" + std::to_string(c->lookahead) + (c->overlap ? "" : " serial");
  for (int i = 0; i < c->n_chains; ++i) {
    ofdg_ctx::Chain& ch = c->chains[i];
    if ((e = hipStreamCreateWithFlags(&ch.stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ch.stage.free_ev, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ch.ev_prep, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ch.ev_done, hipEventDisableTiming)) != hipSuccess) {
      g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
      return OFDG_EHIP;
    }
  }
  // fail early (and loudly) if the gfx950 code object is not usable on this device
  {
    hipFuncAttributes fa;
    e = hipFuncGetAttributes(&fa, (const void*)compose_rigid_kernel);
    if (e != hipSuccess) {
      g_create_error = std::string("compose_rigid_kernel is not loadable (is the gfx950 code object present?): ") + hipGetErrorString(e);
      return OFDG_EHIP;
    }
  }
  *out = c.release();
  return OFDG_OK;
}

const ofdg_params* ofdg_ctx_params(const ofdg_ctx* c) { return c ? &c->prm : nullptr; }
const char* ofdg_ctx_info(const ofdg_ctx* c) { return c ? c->info.c_str() : ""; }

void ofdg_destroy(ofdg_ctx* c) {
  if (!c) return;
  (void)hipDeviceSynchronize();
  if (c->pool) (void)hipFree(c->pool);
  if (c->d_prep_paths) (void)hipFree(c->d_prep_paths);
  if (c->pool_fg) (void)hipFree(c->pool_fg);
  if (c->pool_bg) (void)hipFree(c->pool_bg);
  for (uint32_t* im : c->mixed_images) if (im) (void)hipFree(im);
  if (c->d_tex_table) (void)hipFree(c->d_tex_table);
  auto drop_slot = [](ofdg_ctx::Slot& sl) {
    sl.d_shapes.release(); sl.d_frames.release(); sl.d_verts.release(); sl.d_objects.release(); sl.d_samples.release(); sl.d_rec.release();
    sl.d_items.release(); sl.d_blockmask.release(); sl.d_bgprep.release(); sl.d_bgtex.release();
    sl.d_bgC.release();
    sl.d_croptab.release(); sl.d_bgwarp.release(); sl.d_bgwarp_max.release();
    if (sl.d_item_count) (void)hipFree(sl.d_item_count);
    if (sl.ev_uploaded) (void)hipEventDestroy(sl.ev_uploaded);
  };
  auto drop_stage = [](ofdg_ctx::Stage& g) {
    if (g.h) (void)hipHostFree(g.h);
    if (g.free_ev) (void)hipEventDestroy(g.free_ev);
  };
  for (auto& sl : c->slots) drop_slot(sl);
  drop_stage(c->user_stage);
  for (auto& ch : c->chains) {
    drop_slot(ch.slot);
    drop_stage(ch.stage);
    ch.cov.release();
    if (ch.ev_prep) (void)hipEventDestroy(ch.ev_prep);
    if (ch.ev_done) (void)hipEventDestroy(ch.ev_done);
    if (ch.stream) (void)hipStreamDestroy(ch.stream);
  }
  if (c->d_rs_xi) { (void)hipFree(c->d_rs_xi); (void)hipFree(c->d_rs_xa); (void)hipFree(c->d_rs_yi); (void)hipFree(c->d_rs_ya); }
  drop_counter_croptab(c);
  if (c->d_warp) (void)hipFree(c->d_warp);
  if (c->d_warp_max) (void)hipFree(c->d_warp_max);
  c->d_cs_bps.release(); c->d_cs_nobj.release();
  if (c->d_cs_tab) (void)hipFree(c->d_cs_tab);
  if (c->d_err) (void)hipFree(c->d_err);
  c->d_bg_at_x.release(); c->d_bg_at_y.release(); c->d_bg_alpha_x.release(); c->d_bg_alpha_y.release();
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->err_stream) (void)hipStreamDestroy(c->err_stream);
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  delete c;
}

// ---- texture pool -------------------------------------------------------------------
static int pool_check_dims(ofdg_ctx* c, int n, int w, int h) {
  const int W = c->prm.width, H = c->prm.height;
  if (n < 1) { c->err = "texture pool needs at least one image"; return OFDG_ETEXTURES; }
  (void)W; (void)H;
  if (w < 2 || h < 2) { c->err = "pool images must be at least 2 x 2"; return OFDG_ETEXTURES; }
  return OFDG_OK;
}

int ofdg_pool_alloc(ofdg_ctx* c, int n, int w, int h) {
  if (!c) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }  // (batches prepared ahead read the old pool / crops)
  int rc = pool_check_dims(c, n, w, h);
  if (rc != OFDG_OK) return rc;
  HIP_OK(c, hipDeviceSynchronize());
  if (c->pool) { HIP_OK(c, hipFree(c->pool)); c->pool = nullptr; }
  HIP_OK(c, hipMalloc((void**)&c->pool, (size_t)n * w * h * sizeof(uint32_t)));
  HIP_OK(c, hipMemset(c->pool, 0, (size_t)n * w * h * sizeof(uint32_t)));
  c->pool_n = n; c->pool_w = w; c->pool_h = h;
  c->pool_final = false;
  c->pool_mixed = false;
  c->pool_kind = OFDG_POOL_UNIFORM;
  return OFDG_OK;
}

// A pool of n images of DIFFERENT sizes (real texture lists): only what the path reads is kept - every
// image's W x H foreground texture and 2W x 2H background texture (centre crop, or the resized image if
// it is smaller; DG:96-106), in two uniform arrays.  background_prep needs the originals: not available here.
int ofdg_pool_alloc_mixed(ofdg_ctx* c, int n) {
  if (!c) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }  // (batches prepared ahead read the old pool / crops)
  if (n < 1) { c->err = "texture pool needs at least one image"; return OFDG_ETEXTURES; }
  const int W = c->prm.width, H = c->prm.height;
  HIP_OK(c, hipDeviceSynchronize());
  for (uint32_t* im : c->mixed_images) if (im) (void)hipFree(im);
  c->mixed_images.assign((size_t)n, nullptr);
  if (c->d_tex_table) { (void)hipFree(c->d_tex_table); c->d_tex_table = nullptr; }
  c->tex_table.clear();
  if (c->pool) { HIP_OK(c, hipFree(c->pool)); c->pool = nullptr; }
  if (c->pool_fg) { HIP_OK(c, hipFree(c->pool_fg)); c->pool_fg = nullptr; }
  if (c->pool_bg) { HIP_OK(c, hipFree(c->pool_bg)); c->pool_bg = nullptr; }
  HIP_OK(c, hipMalloc((void**)&c->pool_fg, (size_t)n * W * H * sizeof(uint32_t)));
  HIP_OK(c, hipMalloc((void**)&c->pool_bg, (size_t)n * 4 * W * H * sizeof(uint32_t)));
  HIP_OK(c, hipMemset(c->pool_fg, 0, (size_t)n * W * H * sizeof(uint32_t)));
  HIP_OK(c, hipMemset(c->pool_bg, 0, (size_t)n * 4 * W * H * sizeof(uint32_t)));
  HIP_OK(c, hipDeviceSynchronize());
  c->pool_n = n; c->pool_w = 0; c->pool_h = 0;
  c->fg_src = TexSource{(uint64_t)W * H, 0, W, 0};
  c->bg_src = TexSource{(uint64_t)4 * W * H, 0, 2 * W, 0};
  c->pool_mixed = true;
  c->pool_final = true;
  c->pool_kind = OFDG_POOL_MIXED;
  c->mixed_sizes.assign((size_t)n, std::make_pair(0, 0));
  return OFDG_OK;
}

// image `index` of a mixed pool: planar B,G,R u8 of any size >= 2 x 2
int ofdg_pool_upload_mixed(ofdg_ctx* c, int index, const uint8_t* bgr_planar, int w, int h) {
  if (!c || !bgr_planar) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }  // (batches prepared ahead read the old pool / crops)
  if (!c->pool_mixed || index < 0 || index >= c->pool_n || w < 2 || h < 2) {
    c->err = "pool_upload_mixed: no mixed pool (ofdg_pool_alloc_mixed), bad index or image smaller than 2 x 2";
    return OFDG_ETEXTURES;
  }
  const int W = c->prm.width, H = c->prm.height;
  uint8_t* tmp = nullptr;
  uint32_t* img = nullptr;
  const size_t n = (size_t)w * h;
  HIP_OK(c, hipDeviceSynchronize());
  HIP_OK(c, hipMalloc((void**)&tmp, 3 * n));
  HIP_OK(c, hipMalloc((void**)&img, n * sizeof(uint32_t)));
  HIP_OK(c, hipMemcpy(tmp, bgr_planar, 3 * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pool_pack_kernel, dim3(1024), dim3(256), 0, 0, tmp, img, w, h);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipDeviceSynchronize());
  int rc = texture_of_image(c, img, w, h, W, H, c->pool_fg + (size_t)index * W * H);
  if (rc == OFDG_OK) rc = texture_of_image(c, img, w, h, 2 * W, 2 * H, c->pool_bg + (size_t)index * 4 * W * H);
  HIP_OK(c, hipDeviceSynchronize());
  (void)hipFree(tmp);
  if (rc == OFDG_OK && c->prm.background_prep) {  // the background preparation reads the original image
    if (c->mixed_images[(size_t)index]) (void)hipFree(c->mixed_images[(size_t)index]);
    c->mixed_images[(size_t)index] = img;
    if (c->d_tex_table) { (void)hipFree(c->d_tex_table); c->d_tex_table = nullptr; }
  } else {
    (void)hipFree(img);
  }
  if (rc == OFDG_OK) c->mixed_sizes[(size_t)index] = std::make_pair(w, h);
  return rc;
}

int ofdg_pool_synthetic(ofdg_ctx* c, int n, int w, int h, uint32_t seed) {
  int rc = ofdg_pool_alloc(c, n, w, h);
  if (rc != OFDG_OK) return rc;
  hipLaunchKernelGGL(pool_synth_kernel, dim3(256 * 8), dim3(256), 0, 0, c->pool, n, w, h, seed);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipDeviceSynchronize());
  c->pool_final = false;
  c->pool_kind = OFDG_POOL_SYNTHETIC;
  c->pool_seed = seed;
  return OFDG_OK;
}

int ofdg_pool_upload(ofdg_ctx* c, int index, const uint8_t* bgr_planar, int w, int h) {
  if (!c || !bgr_planar) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }  // (batches prepared ahead read the old pool / crops)
  if (!c->pool || index < 0 || index >= c->pool_n || w != c->pool_w || h != c->pool_h) {
    c->err = "pool_upload: index / size does not match the allocated pool";
    return OFDG_ETEXTURES;
  }
  uint8_t* tmp = nullptr;
  const size_t n = (size_t)w * h;
  HIP_OK(c, hipMalloc((void**)&tmp, 3 * n));
  HIP_OK(c, hipMemcpy(tmp, bgr_planar, 3 * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pool_pack_kernel, dim3(1024), dim3(256), 0, 0, tmp, c->pool + (size_t)index * n, w, h);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipDeviceSynchronize());
  HIP_OK(c, hipFree(tmp));
  c->pool_final = false;
  return OFDG_OK;
}

int ofdg_pool_download(ofdg_ctx* c, int index, uint8_t* bgr_planar) {
  if (!c || !bgr_planar) return OFDG_EINVAL;
  if (!c->pool || index < 0 || index >= c->pool_n) { c->err = "pool_download: bad index"; return OFDG_ETEXTURES; }
  uint8_t* tmp = nullptr;
  const size_t n = (size_t)c->pool_w * c->pool_h;
  HIP_OK(c, hipMalloc((void**)&tmp, 3 * n));
  hipLaunchKernelGGL(pool_unpack_kernel, dim3(1024), dim3(256), 0, 0, c->pool + (size_t)index * n, tmp, c->pool_w, c->pool_h);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipMemcpy(bgr_planar, tmp, 3 * n, hipMemcpyDeviceToHost));
  HIP_OK(c, hipFree(tmp));
  return OFDG_OK;
}

int ofdg_pool_info(const ofdg_ctx* c, int* n, int* w, int* h) {
  if (!c) return OFDG_EINVAL;
  if (n) *n = c->pool_n;
  if (w) *w = c->pool_w;
  if (h) *h = c->pool_h;
  return OFDG_OK;
}

// The resident pool as raw device memory (n * h * w BGRX texels): for filling it from another GPU - rank 0 loads
// the texture collection, the other ranks receive it with one RCCL broadcast over xGMI into their own replica.
// `mark_written` != 0 tells the context that the contents changed (derived textures are rebuilt at the next use).
int ofdg_pool_device(ofdg_ctx* c, void** ptr, unsigned long long* bytes, int mark_written) {
  if (!c || !ptr || !bytes) return OFDG_EINVAL;
  if (!c->pool || c->pool_mixed) { c->err = "pool_device: needs a pool of one image size (ofdg_pool_alloc / ofdg_pool_synthetic)"; return OFDG_ETEXTURES; }
  HIP_OK(c, hipDeviceSynchronize());
  *ptr = (void*)c->pool;
  *bytes = (unsigned long long)c->pool_n * c->pool_w * c->pool_h * sizeof(uint32_t);
  if (mark_written) {
    c->pool_final = false;
    int rcd = discard_all_prepared(c);  // (batches prepared ahead would render the old contents)
    if (rcd != OFDG_OK) return rcd;
  }
  return OFDG_OK;
}

int ofdg_pool_device_mixed(ofdg_ctx* c, void** fg, unsigned long long* fg_bytes, void** bg, unsigned long long* bg_bytes) {
  if (!c || !fg || !fg_bytes || !bg || !bg_bytes) return OFDG_EINVAL;
  if (!c->pool_mixed) { c->err = "pool_device_mixed: needs a mixed pool (ofdg_pool_alloc_mixed)"; return OFDG_ETEXTURES; }
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }  // (the caller is about to write the textures)
  HIP_OK(c, hipDeviceSynchronize());
  const unsigned long long px = (unsigned long long)c->prm.width * c->prm.height;
  *fg = (void*)c->pool_fg; *fg_bytes = (unsigned long long)c->pool_n * px * sizeof(uint32_t);
  *bg = (void*)c->pool_bg; *bg_bytes = (unsigned long long)c->pool_n * 4 * px * sizeof(uint32_t);
  return OFDG_OK;
}

// ---- multi-GPU start-up: what the root broadcasts (csrc/comm.cpp) ------------------------------------
int ofdg_setup_of(const ofdg_ctx* c, ofdg_setup* su, ofdg_tex_entry* table, int table_cap) {
  if (!c || !su || table_cap < 0 || (table_cap > 0 && !table)) return OFDG_EINVAL;
  std::memset(su, 0, sizeof(*su));
  const ofdg_params& p = c->prm;
  su->seed = p.seed; su->mode = p.mode; su->width = p.width; su->height = p.height; su->num_objects = p.num_objects;
  su->use_antialiasing = p.use_antialiasing; su->batch_size = p.batch_size; su->sampler = p.sampler;
  su->background_prep = p.background_prep; su->max_shapes_per_sample = p.max_shapes_per_sample;
  // (a pool of images of different sizes needs every entry of the table: a table that does not fit is the root's failure,
  //  which the broadcast carries to every rank)
  if (c->pool_mixed && c->pool_n > table_cap) su->status = OFDG_ECAPACITY;
  su->n_tex = c->pool_n; su->pool_kind = c->pool_kind; su->pool_w = c->pool_w; su->pool_h = c->pool_h; su->pool_seed = c->pool_seed;
  su->n_table = std::min(c->pool_n, table_cap);
  const uint64_t W = (uint64_t)p.width, H = (uint64_t)p.height;
  for (int i = 0; i < su->n_table; ++i) {
    ofdg_tex_entry& e = table[i];
    std::memset(&e, 0, sizeof(e));
    if (c->pool_mixed) {  // the path reads the derived [n][H][W] foreground textures
      e.offset = (uint64_t)i * W * H; e.pitch = (uint32_t)W;
      e.w = (uint32_t)c->mixed_sizes[(size_t)i].first; e.h = (uint32_t)c->mixed_sizes[(size_t)i].second;
    } else {
      e.offset = (uint64_t)i * (uint64_t)c->pool_w * (uint64_t)c->pool_h; e.pitch = (uint32_t)c->pool_w;
      e.w = (uint32_t)c->pool_w; e.h = (uint32_t)c->pool_h;
    }
  }
  return OFDG_OK;
}

int ofdg_setup_alloc_pool(ofdg_ctx* c, const ofdg_setup* su, const ofdg_tex_entry* table) {
  if (!c || !su) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }  // (batches prepared ahead read the old pool / crops)
  if (su->width != c->prm.width || su->height != c->prm.height) { c->err = "setup_alloc_pool: the context was not created from this setup"; return OFDG_EINVAL; }
  if (su->pool_kind == OFDG_POOL_SYNTHETIC) return ofdg_pool_synthetic(c, su->n_tex, su->pool_w, su->pool_h, su->pool_seed);
  if (su->pool_kind != OFDG_POOL_MIXED) return ofdg_pool_alloc(c, su->n_tex, su->pool_w, su->pool_h);
  int rc = ofdg_pool_alloc_mixed(c, su->n_tex);
  if (rc != OFDG_OK) return rc;
  if (!table || su->n_table < su->n_tex) { c->err = "setup_alloc_pool: a mixed pool needs the index table (image sizes)"; return OFDG_EINVAL; }
  for (int i = 0; i < su->n_tex; ++i) {
    c->mixed_sizes[(size_t)i] = std::make_pair((int)table[i].w, (int)table[i].h);
    if (c->prm.background_prep)  // the whole images arrive with ofdg_comm_bcast_pool
      HIP_OK(c, hipMalloc((void**)&c->mixed_images[(size_t)i], (size_t)table[i].w * table[i].h * sizeof(uint32_t)));
  }
  return OFDG_OK;
}

// image `index` of a mixed pool kept whole for the background preparation, as raw device memory (w * h BGRX texels)
int ofdg_pool_device_image(ofdg_ctx* c, int index, void** ptr, unsigned long long* bytes) {
  if (!c || !ptr || !bytes) return OFDG_EINVAL;
  if (!c->pool_mixed || !c->prm.background_prep || index < 0 || index >= c->pool_n || !c->mixed_images[(size_t)index]) {
    c->err = "pool_device_image: no whole image " + std::to_string(index) + " (mixed pool with background_prep only)";
    return OFDG_ETEXTURES;
  }
  HIP_OK(c, hipDeviceSynchronize());
  *ptr = (void*)c->mixed_images[(size_t)index];
  *bytes = (unsigned long long)c->mixed_sizes[(size_t)index].first * c->mixed_sizes[(size_t)index].second * sizeof(uint32_t);
  return OFDG_OK;
}

// ---- sampler ---------------------------------------------------------------------------
int ofdg_sample(ofdg_ctx* c, int n_tasks, ofdg_task* tasks, ofdg_blueprint* bps, int bps_capacity, int* n_bps) {
  if (!c || !tasks || !bps || !n_bps || n_tasks < 0) return OFDG_EINVAL;
  if (c->prm.sampler != OFDG_SAMPLER_REF) { c->err = "only OFDG_SAMPLER_REF is implemented on the host"; return OFDG_EINVAL; }
  std::vector<ofdg_blueprint> pool;
  for (int i = 0; i < n_tasks; ++i) {
    int rc = c->sampler->next_task(&pool, &tasks[i], &c->err);
    if (rc != OFDG_OK) return rc;
  }
  *n_bps = (int)pool.size();
  if ((int)pool.size() > bps_capacity) { c->err = "blueprint capacity exceeded"; return OFDG_ECAPACITY; }
  std::memcpy(bps, pool.data(), pool.size() * sizeof(ofdg_blueprint));
  return OFDG_OK;
}

// block masks of a slot: two parities, cleared once here; afterwards raster_kernel clears
// the other parity every launch
static int reserve_blockmask(ofdg_ctx* c, ofdg_ctx::Slot& sl, int n_samples) {
  const int W = c->prm.width, H = c->prm.height;
  const size_t words = (size_t)n_samples * ((W + kTileW - 1) / kTileW) * ((H + kBandRows - 1) / kBandRows) * 2;
  if (words > sl.box_stride) {
    HIP_OK(c, hipDeviceSynchronize());
    HIP_OK(c, sl.d_blockmask.reserve((words + 16) * 2));
    sl.box_stride = sl.d_blockmask.cap / 2;
    HIP_OK(c, hipMemset(sl.d_blockmask.p, 0, sl.d_blockmask.cap * sizeof(unsigned long long)));
    HIP_OK(c, hipDeviceSynchronize());  // (the fill runs on the null stream; the kernels use non-blocking streams)
    sl.mask_used[0] = sl.mask_used[1] = 0;
  }
  return OFDG_OK;
}

// CImg linear resize tables (X then Y), boundary 0, upscaling branch: (W+1) x (H+1) -> 2W x 2H (DG:1197-1200)
static int ensure_resize_tables(ofdg_ctx* c) {
  const int W = c->prm.width, H = c->prm.height;
  if (c->rs_w == W && c->rs_h == H) return OFDG_OK;
  auto table = [](int w, int sx, std::vector<int>& idx, std::vector<double>& alpha) {
    idx.resize(sx); alpha.resize(sx);
    const double f = (sx > w) ? (sx > 1 ? (w - 1.) / (sx - 1) : 0) : (double)w / sx;
    double curr = 0, old = 0;
    int at = 0;
    for (int x = 0; x < sx; ++x) {
      alpha[x] = curr - (unsigned int)curr;
      idx[x] = at;
      old = curr;
      curr = std::min(w - 1., curr + f);
      at += (int)((unsigned int)curr - (unsigned int)old);
    }
  };
  std::vector<int> xi, yi;
  std::vector<double> xa, ya;
  table(W + 1, 2 * W, xi, xa);
  table(H + 1, 2 * H, yi, ya);
  if (c->d_rs_xi) { (void)hipFree(c->d_rs_xi); (void)hipFree(c->d_rs_xa); (void)hipFree(c->d_rs_yi); (void)hipFree(c->d_rs_ya); }
  HIP_OK(c, hipMalloc((void**)&c->d_rs_xi, xi.size() * sizeof(int)));
  HIP_OK(c, hipMalloc((void**)&c->d_rs_xa, xa.size() * sizeof(double)));
  HIP_OK(c, hipMalloc((void**)&c->d_rs_yi, yi.size() * sizeof(int)));
  HIP_OK(c, hipMalloc((void**)&c->d_rs_ya, ya.size() * sizeof(double)));
  HIP_OK(c, hipMemcpy(c->d_rs_xi, xi.data(), xi.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(c, hipMemcpy(c->d_rs_xa, xa.data(), xa.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_OK(c, hipMemcpy(c->d_rs_yi, yi.data(), yi.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(c, hipMemcpy(c->d_rs_ya, ya.data(), ya.size() * sizeof(double), hipMemcpyHostToDevice));
  c->rs_w = W; c->rs_h = H;
  return OFDG_OK;
}

// counter sampler, mode 9: the static crop table (see ofdg_ctx::d_cs_croptab)
static int ensure_counter_croptab(ofdg_ctx* c) {
  if (c->d_cs_croptab) return OFDG_OK;
  const int n = c->crop_server.n_crops;
  if (!c->d_warp || n < 1) { c->err = "mode 9 needs warp fields: call ofdg_warp_generate or ofdg_warp_upload first"; return OFDG_EINVAL; }
  int rc = ensure_resize_tables(c);
  if (rc != OFDG_OK) return rc;
  const int W = c->prm.width, H = c->prm.height;
  const size_t crop_floats = (size_t)4 * (W + 1) * (H + 1), bg_floats = (size_t)4 * 2 * W * 2 * H;
  HIP_OK(c, hipDeviceSynchronize());
  HIP_OK(c, hipMalloc((void**)&c->d_cs_bgwarp, (size_t)n * bg_floats * sizeof(float)));
  HIP_OK(c, hipMalloc((void**)&c->d_cs_bgwarp_max, (size_t)n * sizeof(unsigned)));
  HIP_OK(c, hipMemset(c->d_cs_bgwarp_max, 0, (size_t)n * sizeof(unsigned)));
  HIP_OK(c, hipMalloc((void**)&c->d_cs_croptab, (size_t)2 * n * sizeof(DevCropRef)));
  std::vector<DevCropRef> tab((size_t)2 * n);
  for (int k = 0; k < n; ++k) {
    const float* src = c->d_warp + (size_t)k * crop_floats;
    float* dst = c->d_cs_bgwarp + (size_t)k * bg_floats;
    hipLaunchKernelGGL(wf_resize2_kernel, dim3((2 * W * 2 * H + 255) / 256), dim3(256), 0, 0, src, W + 1, H + 1, 2 * W, 2 * H,
                       c->d_rs_xi, c->d_rs_xa, c->d_rs_yi, c->d_rs_ya, dst, c->d_cs_bgwarp_max + k);
    HIP_OK(c, hipGetLastError());
    tab[k] = make_crop_ref(src, c->d_warp_max + k, W + 1, H + 1);
    tab[(size_t)n + k] = make_crop_ref(dst, c->d_cs_bgwarp_max + k, 2 * W, 2 * H);
  }
  HIP_OK(c, hipMemcpy(c->d_cs_croptab, tab.data(), tab.size() * sizeof(DevCropRef), hipMemcpyHostToDevice));
  HIP_OK(c, hipDeviceSynchronize());
  return OFDG_OK;
}
static void drop_counter_croptab(ofdg_ctx* c) {
  if (c->d_cs_croptab) { (void)hipFree(c->d_cs_croptab); c->d_cs_croptab = nullptr; }
  if (c->d_cs_bgwarp) { (void)hipFree(c->d_cs_bgwarp); c->d_cs_bgwarp = nullptr; }
  if (c->d_cs_bgwarp_max) { (void)hipFree(c->d_cs_bgwarp_max); c->d_cs_bgwarp_max = nullptr; }
}

// CImg<unsigned char>::get_resize(tw, th, -100, -100, 3) of every pool image (X pass, then Y pass, u8 in
// between) into dst[n][th][tw]
static int resize_images(ofdg_ctx* c, const uint32_t* images, int n, int w, int h, int tw, int th, uint32_t* out_images) {
  uint32_t* mid = nullptr;  // [n][h][tw]
  HIP_OK(c, hipMalloc((void**)&mid, (size_t)n * tw * h * sizeof(uint32_t)));
  auto pass = [&](const uint32_t* src, uint32_t* out, int sw, int sh, int s, int along_x) -> int {
    const int len = along_x ? sw : sh;
    int* d_at = nullptr;
    double* d_alpha = nullptr;
    if (s > len) {  // enlarging: CImg's running sums (boundary 0)
      std::vector<int> at(s);
      std::vector<double> alpha(s);
      const double f = s > 1 ? (len - 1.) / (s - 1) : 0;
      double curr = 0, old = 0;
      int pos = 0;
      for (int x = 0; x < s; ++x) {
        alpha[x] = curr - (unsigned int)curr;
        at[x] = pos;
        old = curr;
        curr = std::min(len - 1., curr + f);
        pos += (int)((unsigned int)curr - (unsigned int)old);
      }
      HIP_OK(c, hipMalloc((void**)&d_at, s * sizeof(int)));
      HIP_OK(c, hipMalloc((void**)&d_alpha, s * sizeof(double)));
      HIP_OK(c, hipMemcpy(d_at, at.data(), s * sizeof(int), hipMemcpyHostToDevice));
      HIP_OK(c, hipMemcpy(d_alpha, alpha.data(), s * sizeof(double), hipMemcpyHostToDevice));
    }
    if (s == len) {
      HIP_OK(c, hipMemcpy(out, src, (size_t)n * sw * sh * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    } else {
      hipLaunchKernelGGL(pool_resize_axis_kernel, dim3(2048), dim3(256), 0, 0, src, out, n, sw, sh, s, along_x, d_at, d_alpha);
      HIP_OK(c, hipGetLastError());
    }
    HIP_OK(c, hipDeviceSynchronize());
    if (d_at) { (void)hipFree(d_at); (void)hipFree(d_alpha); }
    return OFDG_OK;
  };
  int rc = pass(images, mid, w, h, tw, 1);
  if (rc == OFDG_OK) rc = pass(mid, out_images, tw, h, th, 0);
  (void)hipFree(mid);
  return rc;
}
static int pool_resized_copy(ofdg_ctx* c, int tw, int th, uint32_t** dst) {
  if (*dst) { HIP_OK(c, hipFree(*dst)); *dst = nullptr; }
  HIP_OK(c, hipMalloc((void**)dst, (size_t)c->pool_n * tw * th * sizeof(uint32_t)));
  return resize_images(c, c->pool, c->pool_n, c->pool_w, c->pool_h, tw, th, *dst);
}
// the tw x th texture of ONE image of any size (getRandomizedCrop with default arguments, DG:96-106):
// its centre crop if it is large enough, else the resized whole image
static int texture_of_image(ofdg_ctx* c, const uint32_t* image, int w, int h, int tw, int th, uint32_t* out) {
  if (w >= tw && h >= th) {
    HIP_OK(c, hipMemcpy2D(out, (size_t)tw * 4, image + (size_t)(h / 2 - th / 2) * w + (w / 2 - tw / 2), (size_t)w * 4, (size_t)tw * 4, th,
                          hipMemcpyDeviceToDevice));
    return OFDG_OK;
  }
  return resize_images(c, image, 1, w, h, tw, th, out);
}

// after the pool contents are final: where foreground / background textures are read from
static int finalise_pool(ofdg_ctx* c) {
  if (c->pool_final) return OFDG_OK;
  if (c->pool_mixed) { c->pool_final = true; return OFDG_OK; }
  HIP_OK(c, hipDeviceSynchronize());
  const int W = c->prm.width, H = c->prm.height, w = c->pool_w, h = c->pool_h;
  const uint64_t img = (uint64_t)w * h;
  if (c->pool_fg) { (void)hipFree(c->pool_fg); c->pool_fg = nullptr; }
  if (c->pool_bg) { (void)hipFree(c->pool_bg); c->pool_bg = nullptr; }
  if (w >= W && h >= H) {
    c->fg_src = TexSource{img, (uint64_t)(h / 2 - H / 2) * w + (uint64_t)(w / 2 - W / 2), w, 0};
  } else {
    int rc = pool_resized_copy(c, W, H, &c->pool_fg);
    if (rc != OFDG_OK) return rc;
    c->fg_src = TexSource{(uint64_t)W * H, 0, W, 0};
  }
  if (w >= 2 * W && h >= 2 * H) {
    c->bg_src = TexSource{img, (uint64_t)(h / 2 - H) * w + (uint64_t)(w / 2 - W), w, 0};
  } else {
    int rc = pool_resized_copy(c, 2 * W, 2 * H, &c->pool_bg);
    if (rc != OFDG_OK) return rc;
    c->bg_src = TexSource{(uint64_t)4 * W * H, 0, 2 * W, 0};
  }
  c->pool_final = true;
  return OFDG_OK;
}

// background_prep on a mixed pool: the per-image table (address, size) on the device, and the workspace the staged chain
// needs for the largest crop of a rotated image
static int ensure_tex_table(ofdg_ctx* c) {
  if (!c->pool_mixed || !c->prm.background_prep || c->d_tex_table) return OFDG_OK;
  c->tex_table.resize((size_t)c->pool_n);
  for (int i = 0; i < c->pool_n; ++i) {
    if (!c->mixed_images[(size_t)i]) { c->err = "background_prep: image " + std::to_string(i) + " of the mixed pool has not been uploaded"; return OFDG_ETEXTURES; }
    c->tex_table[(size_t)i] = DevTexEntry{(uint64_t)(uintptr_t)c->mixed_images[(size_t)i], c->mixed_sizes[(size_t)i].first, c->mixed_sizes[(size_t)i].second};
  }
  HIP_OK(c, hipMalloc((void**)&c->d_tex_table, c->tex_table.size() * sizeof(DevTexEntry)));
  HIP_OK(c, hipMemcpy(c->d_tex_table, c->tex_table.data(), c->tex_table.size() * sizeof(DevTexEntry), hipMemcpyHostToDevice));
  return OFDG_OK;
}
// `fusable`: every pool image is at least 2W x 2H, so a crop is at most 4/3 of the texture (beyond that it is refused), which
// is what the tiles of bgprep_stream_kernel hold; smaller images are resized by any factor (DG:102-106): the two-kernel form
static void bgprep_caps(ofdg_ctx* c, int* cap_cw, int* cap_ch, bool* fusable) {
  if (c->bg_cap_n == c->pool_n && c->bg_cap_cw > 0) { *cap_cw = c->bg_cap_cw; *cap_ch = c->bg_cap_ch; *fusable = c->bg_fusable; return; }  // (per pool, not per step)
  const int TW = 2 * c->prm.width, TH = 2 * c->prm.height;
  // crops of the rotated image up to zoom 0.75 (the sampler draws 0.8 .. 1.2) ...
  int cw = (int)((float)TW / 0.75f) + 2, ch = (int)((float)TH / 0.75f) + 2;
  // ... or the whole rotated image when a pool image is smaller than 2W x 2H (DG:102-106; rotation by up to +-3.2 "degrees")
  bool all_large = true;
  auto small = [&](int w, int h) { if (w < TW || h < TH) { all_large = false; cw = std::max(cw, w + h / 8 + 4); ch = std::max(ch, h + w / 8 + 4); } };
  if (c->pool_mixed) for (const auto& wh : c->mixed_sizes) small(wh.first, wh.second);
  else small(c->pool_w, c->pool_h);
  c->bg_cap_cw = cw; c->bg_cap_ch = ch; c->bg_cap_n = c->pool_n; c->bg_fusable = all_large;
  *cap_cw = cw; *cap_ch = ch; *fusable = all_large;
}

// per-chain coverage workspaces of a batch of n_shapes outlines.  Growing them waits for the device: every chain may be
// reading its own.
static int reserve_workspaces(ofdg_ctx* c, size_t n_shapes) {
  const size_t need_cov = n_shapes * 2 * (size_t)c->prm.width * c->prm.height + 16;
  if (need_cov <= c->chains[0].cov.cap) return OFDG_OK;
  HIP_OK(c, hipDeviceSynchronize());
  for (int k = 0; k < c->n_chains; ++k) HIP_OK(c, c->chains[k].cov.reserve(need_cov));
  return OFDG_OK;
}

// ---- render -------------------------------------------------------------------------------
// device counter sampler + device realize fill the slot's records (no host data)
static int prepare_backgrounds(ofdg_ctx* c, ofdg_ctx::Slot& sl, int n, bool records_resident, hipStream_t s, uint32_t* err, hipEvent_t stop = nullptr);
static int launch_counter_sampler(ofdg_ctx* c, ofdg_ctx::Slot& sl, long long first_index, hipStream_t s, uint32_t* err) {
  const int stride = sl.res_shapes / sl.res_samples;
  const int prep = c->prm.background_prep ? 1 : 0;
  CsRealizeDims D{c->prm.width, c->prm.height, c->pool_n, c->pool_w, c->pool_h, sl.res_samples, stride, prep,
                  c->prm.mode == 9 ? c->crop_server.n_crops : 0, c->fg_src.stride, c->fg_src.origin, c->bg_src.stride, c->bg_src.origin,
                  (unsigned long long)(uintptr_t)c->pool, c->d_tex_table, c->prm.mode == 9 ? c->d_cs_croptab : nullptr};
  hipLaunchKernelGGL(cs_sample_realize_kernel, dim3(sl.res_samples * kCsGroups), dim3(64), 0, s, c->cs_mode, D, first_index,
                     sl.d_shapes.p, sl.d_objects.p, sl.d_samples.p, err, sl.d_bgprep.p);
  HIP_OK(c, hipGetLastError());
  sl.bgprep_pending = prep != 0;  // (rendered behind raster: launch_prepare)
  return OFDG_OK;
}

// OFDG_STREAM_OWN as a call's `stream`: the internal stream that call works on (what ofdg_stream() returns right before it)
static void* own_stream(ofdg_ctx* c, void* stream) { return stream == OFDG_STREAM_OWN ? ofdg_stream(c) : stream; }

static ofdg_ctx::Chain& take_chain(ofdg_ctx* c) {
  const int k = (int)(c->next_chain % (unsigned)c->n_chains);
  c->next_chain++;
  c->last_chain = k;
  return c->chains[k];
}
// the stream chain `ch` works on for a call made with the caller's stream `st`
static hipStream_t chain_stream(const ofdg_ctx* c, const ofdg_ctx::Chain& ch, hipStream_t st) { return c->overlap ? ch.stream : st; }

// A prepared batch that will never be composed (the caller's sequence jumped, the pool changed, the chain's private slot is
// needed for something else): forget it.  compose is what resets the slot's raster work list, so do that here.
static int discard_prepared(ofdg_ctx* c, ofdg_ctx::Chain& ch) {
  if (!ch.prep.valid) return OFDG_OK;
  ch.prep.valid = false;
  if (ch.prep.slot && ch.prep.slot->d_item_count) HIP_OK(c, hipMemsetAsync(ch.prep.slot->d_item_count, 0, sizeof(int), ch.prep.stream));
  // ... and what its kernels flagged is nobody's: the batch is never handed over (its word must not speak for ticket + kErrWords,
  // nor make ofdg_synchronize report a batch that was never composed)
  if (ch.prep.ticket >= 0 && ch.prep.ticket + ofdg_ctx::kErrWords > c->ticket) {
    HIP_OK(c, hipMemsetAsync(err_word(c, ch.prep.ticket), 0, sizeof(uint32_t), ch.prep.stream));
    c->word_clean[(size_t)(ch.prep.ticket % ofdg_ctx::kErrWords)] = true;
  }
  return OFDG_OK;
}
static int discard_all_prepared(ofdg_ctx* c) {
  c->bg_cap_n = -1;  // (called by everything that changes the pool)
  for (int k = 0; k < c->n_chains; ++k) { int rc = discard_prepared(c, c->chains[k]); if (rc != OFDG_OK) return rc; }
  return OFDG_OK;
}

static RenderDims render_dims(const ofdg_ctx* c, const ofdg_ctx::Slot& sl) {
  const int W = c->prm.width, H = c->prm.height;
  RenderDims dm;
  dm.W = W; dm.H = H; dm.pool_w = c->pool_w; dm.pool_h = c->pool_h;
  dm.use_aa = c->prm.use_antialiasing ? 1 : 0;
  dm.n_samples = sl.res_samples;
  dm.n_shapes = sl.res_shapes;
  dm.tiles_x = (W + kTileW - 1) / kTileW;
  dm.tiles_y = (H + kTileH - 1) / kTileH;
  dm.bg_pitch = c->prm.background_prep ? 2 * W : c->bg_src.pitch;
  dm.fg_pitch = c->fg_src.pitch;
  return dm;
}

// The preparation kernels of the batch resident in `sl`, in order on chain `ch`: [counter sampler ->] geom -> raster
//   `st` is the stream of the call the batch is prepared for (only the serial mode prepares on it).
static int launch_prepare(ofdg_ctx* c, ofdg_ctx::Chain& ch, ofdg_ctx::Slot& sl, hipStream_t st, long long cs_first_index = -1,
                          bool hand_over = true) {
  const int W = c->prm.width, H = c->prm.height;
  const int n_sf = sl.res_shapes * 2;
  const RenderDims dm = render_dims(c, sl);
  { int rc = discard_prepared(c, ch); if (rc != OFDG_OK) return rc; }
  const long long ticket = c->ticket++;  // this batch's number: its kernels raise their flags in its own word
  uint32_t* const err = err_word(c, ticket);
  hipEvent_t* ev = nullptr;
  if (c->profiling && c->ev_sets > 0 && (c->launch_count % c->ev_stride) == 0) {
    const size_t set = (size_t)(c->ev_alloc++ % c->ev_sets);
    ev = &c->ev[set * 6];
    c->ev_composed[set] = 0;
    c->ev_bgprep[set] = 0;
  }
  c->launch_count++;
  if (!c->overlap) {
    // everything runs on the caller's stream; a caller that switches streams loses the ordering
    if (c->have_last_user_st && c->last_user_st != st) HIP_OK(c, hipDeviceSynchronize());
    c->last_user_st = st; c->have_last_user_st = true;
  }
  hipStream_t S = chain_stream(c, ch, st);
  uint8_t* cov = ch.cov.p;
  // the slot's records: written on another stream, or still read by a compose of another chain
  // (a user slot rendered again; the chain's private slot only ever sees its own stream)
  if (sl.upload_pending && sl.upload_stream != S) {
    if (hipEventQuery(sl.ev_uploaded) == hipSuccess) sl.upload_pending = false;
    else HIP_OK(c, hipStreamWaitEvent(S, sl.ev_uploaded, 0));
  }
  // the chain's workspace (and private slot) may still be read by its previous compose if that ran on a caller's stream
  if (ch.done_pending && ch.done_stream != S) {
    if (hipEventQuery(ch.ev_done) != hipSuccess) HIP_OK(c, hipStreamWaitEvent(S, ch.ev_done, 0));
    ch.done_pending = false;  // (the chain's stream is ordered behind it from here on)
  }
  if (sl.compose_pending && sl.compose_stream != S) {
    if (hipEventQuery(sl.compose_event) == hipSuccess) sl.compose_pending = false;
    else HIP_OK(c, hipStreamWaitEvent(S, sl.compose_event, 0));
  }
  // mode 9: the batch's own crop table (host path) or the static table of all crops (counter sampler)
  const DevCropRef* croptab = cs_first_index >= 0 ? c->d_cs_croptab : sl.d_croptab.p;
  HIP_OK(c, take_err_word(c, ticket, S));  // (nothing is enqueued unless this caller asks by ticket AND the word's last owner was never asked about)
  if (cs_first_index >= 0) {  // device counter sampler + device realize
    int rc = launch_counter_sampler(c, sl, cs_first_index, S, err);
    if (rc != OFDG_OK) return rc;
  }
  // (the background preparation of the batch goes LAST, right in front of compose: below)
  // geom: outlines, bounding boxes, per-object boxes (parity `bp`), raster work list
  const int bp = sl.box_parity;
  sl.box_parity ^= 1;
  unsigned long long* box_cur = sl.d_blockmask.p + (size_t)bp * sl.box_stride;
  unsigned long long* box_next = sl.d_blockmask.p + (size_t)(bp ^ 1) * sl.box_stride;
  sl.mask_used[bp] = sl.res_samples * dm.tiles_x * ((H + kBandRows - 1) / kBandRows) * 2;
  const int n_mask_words = sl.mask_used[bp ^ 1];
  const bool prof_prep = ev && c->profiling == 2;
  // Profiled launches: a kernel's time is the span between the completion of its predecessor on the
  // chain's stream and its own completion - both taken from the kernels' own dispatch packets (stop
  // events).  A start event is a marker packet in front of the kernel, which delays its dispatch by
  // ~6-10 us: geom (first of the three) and compose (ev[4], launch_compose) have one, raster has none.
  // geom: outlines, boxes, raster work list -> raster: coverage slots + block masks (persistent waves over the list)
  hipExtLaunchKernelGGL(geom_kernel, dim3(std::max(1, (n_sf + kGeomWaves - 1) / kGeomWaves)), dim3(64 * kGeomWaves), 0, S,
                        prof_prep ? ev[0] : nullptr, prof_prep ? ev[1] : nullptr, 0, sl.d_shapes.p, sl.res_shapes, c->d_cs_tab, W, H,
                        sl.d_frames.p, sl.d_verts.p, box_cur, err, sl.d_item_count, sl.d_items.p, croptab);
  HIP_OK(c, hipGetLastError());
  // The batch's last preparation kernel - raster, or the background preparation behind it - carries two things on its own
  // packet: the hand-over event of a compose on a caller's stream, or (profiling 1) the start of the compose launch's time.
  // The ALU-bound background preparation runs LAST, right in front of compose: sampler -> geom -> raster -> preparation ->
  // compose is 4 % faster in the steady state than with the preparation behind the sampler (the latency-bound kernels of a
  // chain follow each other, the two heavy ones too; profiles/r04_experiments_log.md section 11).
  const bool prep_last = sl.bgprep_pending;
  hipEvent_t last_stop = ev ? (c->profiling == 2 ? nullptr : ev[4]) : (hand_over ? ch.ev_prep : nullptr);
  // (a profiled batch with a preparation behind raster also takes raster's completion: the preparation is timed from there)
  hipExtLaunchKernelGGL(raster_kernel, dim3(kRasterGrid * 4 / kRasterWaves), dim3(64 * kRasterWaves), 0, S, nullptr,
                        (ev && (c->profiling == 2 || prep_last)) ? ev[3] : (prep_last ? nullptr : last_stop), 0,
                        sl.d_frames.p, sl.d_items.p, sl.d_item_count, sl.d_verts.p, W, H, cov, box_next, n_mask_words, box_cur);
  HIP_OK(c, hipGetLastError());
  if (prep_last) {
    int rcb = prepare_backgrounds(c, sl, sl.res_samples, /*records_resident=*/true, S, err, (ev && c->profiling == 2) ? ev[2] : last_stop);
    if (rcb != OFDG_OK) return rcb;
    sl.bgprep_pending = false;
    if (ev) c->ev_bgprep[(size_t)(ev - c->ev.data()) / 6] = 1;
  }
  if (ev && hand_over) HIP_OK(c, hipEventRecord(ch.ev_prep, S));
  ch.prep.valid = true; ch.prep.slot = &sl; ch.prep.first_index = cs_first_index; ch.prep.n = sl.res_samples;
  ch.prep.box_cur = box_cur; ch.prep.croptab = croptab; ch.prep.ev = ev; ch.prep.stream = S; ch.prep.ticket = ticket;
  ch.prep.ahead = false;
  return OFDG_OK;
}

// compose of the batch chain `ch` has prepared.  `st` is the caller's stream: if it is not the chain's own stream
// (ofdg_stream), compose runs on `st` instead, behind what the caller enqueued there (the outputs may still be read) and
// behind the chain's preparation kernels.
static int launch_compose(ofdg_ctx* c, ofdg_ctx::Chain& ch, float* d_img0, float* d_img1, float* d_flow, hipStream_t st) {
  if (!ch.prep.valid) { c->err = "internal: compose without a prepared batch"; return OFDG_EINVAL; }
  ofdg_ctx::Slot& sl = *ch.prep.slot;
  const int W = c->prm.width, H = c->prm.height;
  const RenderDims dm = render_dims(c, sl);
  const int compose_grid = dm.tiles_x * dm.tiles_y * dm.n_samples * 4;  // one 64 x 4 strip per single-wave workgroup
  hipEvent_t* ev = ch.prep.ev;
  unsigned long long* box_cur = ch.prep.box_cur;
  const DevCropRef* croptab = ch.prep.croptab;
  uint8_t* cov = ch.cov.p;
  const hipStream_t S = ch.prep.stream;
  const bool foreign = S != st;  // the caller's stream is not the chain's
  const uint32_t* bgpool = c->prm.background_prep ? sl.d_bgtex.p : (c->pool_bg ? c->pool_bg : c->pool);  // (after the slot's buffers are final)
  const uint32_t* fgpool = c->pool_fg ? c->pool_fg : c->pool;
  if (c->prm.background_prep && !bgpool) { c->err = "background_prep: the slot has no prepared backgrounds"; return OFDG_EINVAL; }
  c->last_slot = &sl;
  c->last_ticket = ch.prep.ticket;
  ch.prep.valid = false;  // (consumed from here on; an argument error above leaves it to discard_prepared, which resets the work list)
  // Where compose runs: on the chain's stream, right behind the preparation - or, if the caller passed another stream,
  // on THAT stream (in order with the caller's own work, which may still read the outputs) once the batch is prepared.
  // It is tracked by the chain's event whenever somebody else may have to wait for it: the chain
  // itself (workspace, private slot) after a compose on a caller's stream, other chains for a shared slot.
  hipStream_t CS = S;
  if (foreign) {
    HIP_OK(c, hipStreamWaitEvent(st, ch.ev_prep, 0));
    CS = st;
  }
  const bool shared_slot = &sl != &ch.slot;
  hipEvent_t done = (foreign || shared_slot) ? ch.ev_done : nullptr;
  // (profiled launches: start and stop are the timestamps of the compose kernel's own dispatch packet)
  // profiling 1 (compose only, what bench.py runs the timed region with): NO start marker - the compose launch is timed from
  // the completion of its predecessor on the chain (the last preparation kernel's own packet, ev[4], launch_prepare) to its
  // own completion: its dispatch gap (1 - 2 us) is counted with it, and nothing is added to the stream.  profiling 2: the
  // kernel's own start (a marker, ev[4]) and end.
  hipEvent_t k_start = (ev && c->profiling == 2) ? ev[4] : nullptr, k_stop = ev ? ev[5] : done;
  if (c->prm.mode == 9 && (W & (W - 1)) == 0)
    hipExtLaunchKernelGGL(compose_deform_pow2_kernel, dim3(compose_grid), dim3(64), 0, CS, k_start, k_stop, 0, dm, sl.d_samples.p,
                          sl.d_objects.p, box_cur, cov, fgpool, bgpool, d_img0, d_img1, d_flow, sl.d_frames.p, croptab,
                          sl.d_item_count);
  else if (c->prm.mode == 9)
    hipExtLaunchKernelGGL(compose_deform_kernel, dim3(compose_grid), dim3(64), 0, CS, k_start, k_stop, 0, dm, sl.d_samples.p,
                          sl.d_objects.p, box_cur, cov, fgpool, bgpool, d_img0, d_img1, d_flow, sl.d_frames.p, croptab,
                          sl.d_item_count);
  else if ((W & (W - 1)) == 0)
    hipExtLaunchKernelGGL(compose_rigid_pow2_kernel, dim3(compose_grid), dim3(64), 0, CS, k_start, k_stop, 0, sl.d_samples.p, box_cur,
                          sl.d_objects.p, cov, compose_grid, dm.tiles_x, dm.tiles_y, W, H, dm.use_aa, dm.bg_pitch, dm.fg_pitch, fgpool, bgpool,
                          d_img0, d_img1, d_flow, sl.d_frames.p, sl.d_item_count);
  else
    hipExtLaunchKernelGGL(compose_rigid_kernel, dim3(compose_grid), dim3(64), 0, CS, k_start, k_stop, 0, sl.d_samples.p, box_cur,
                          sl.d_objects.p, cov, compose_grid, dm.tiles_x, dm.tiles_y, W, H, dm.use_aa, dm.bg_pitch, dm.fg_pitch, fgpool, bgpool,
                          d_img0, d_img1, d_flow, sl.d_frames.p, sl.d_item_count);
  HIP_OK(c, hipGetLastError());
  if (ev) {
    if (done) HIP_OK(c, hipEventRecord(done, CS));
    // profiling 1 times compose from the completion of the chain's last preparation kernel: that is the launch's time only
    // when compose is enqueued right behind it on the same stream.  A batch prepared ahead by an earlier call, or composed
    // on a caller's stream behind the hand-over event, would count host and queue idle time: such a set is not a sample.
    const bool span_is_the_launch = c->profiling == 2 || (!foreign && !ch.prep.ahead);
    if (span_is_the_launch) {
      c->ev_count++;
      c->ev_composed[(size_t)(ev - c->ev.data()) / 6] = 1;
    }
  }
  if (done) {
    sl.compose_pending = true; sl.compose_stream = CS; sl.compose_event = done;
    ch.done_pending = true; ch.done_stream = CS;
  } else {
    sl.compose_pending = false;  // private slot on its own chain: stream order is all it needs
  }
  return OFDG_OK;
}

// preparation + compose of the batch resident in `sl`, in order on chain `ch`
static int launch_resident(ofdg_ctx* c, ofdg_ctx::Chain& ch, ofdg_ctx::Slot& sl, float* d_img0, float* d_img1, float* d_flow,
                           hipStream_t st, long long cs_first_index = -1) {
  // (the preparation's completion event is only needed when compose runs on another stream than the chain's)
  int rc = launch_prepare(c, ch, sl, st, cs_first_index, chain_stream(c, ch, st) != st);
  if (rc != OFDG_OK) return rc;
  return launch_compose(c, ch, d_img0, d_img1, d_flow, st);
}

// CImg get_resize(.., 3), enlarging branch: source index and weight of every destination pixel (running double sums,
// boundary 0) for EVERY source length n < s, entry [n * s + x]; once per context and frame size
constexpr int kPrepGrid = 4096;  // single-wave workgroups of bgprep_stream_kernel: four per SIMD (2048 / 3072 / 8192: -6 % / -2 % / -1 %, profiles/r05_experiments_log.md section 2)
static int ensure_bgprep_tables(ofdg_ctx* c) {
  const int TW = 2 * c->prm.width, TH = 2 * c->prm.height;
  if (c->bg_tab_w == TW && c->bg_tab_h == TH) return OFDG_OK;
  auto build = [&](int s, DevBuf<uint16_t>& d_at, DevBuf<double>& d_alpha) -> int {
    std::vector<uint16_t> at((size_t)s * s, 0);
    std::vector<double> alpha((size_t)s * s, 0.0);
    for (int n = 1; n < s; ++n) {
      const double f = s > 1 ? (n - 1.) / (s - 1) : 0;
      double curr = 0, old = 0;
      int pos = 0;
      for (int x = 0; x < s; ++x) {
        alpha[(size_t)n * s + x] = curr - (unsigned int)curr;
        at[(size_t)n * s + x] = (uint16_t)pos;
        old = curr;
        curr = std::min(n - 1., curr + f);
        pos += (int)((unsigned int)curr - (unsigned int)old);
      }
    }
    HIP_OK(c, d_at.reserve(at.size()));
    HIP_OK(c, d_alpha.reserve(alpha.size()));
    HIP_OK(c, hipMemcpy(d_at.p, at.data(), at.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_OK(c, hipMemcpy(d_alpha.p, alpha.data(), alpha.size() * sizeof(double), hipMemcpyHostToDevice));
    return OFDG_OK;
  };
  if (TW > 65535 || TH > 65535) { c->err = "background_prep: frame too large for the resize tables"; return OFDG_EINVAL; }
  HIP_OK(c, hipDeviceSynchronize());
  int rc = build(TW, c->d_bg_at_x, c->d_bg_alpha_x);
  if (rc == OFDG_OK) rc = build(TH, c->d_bg_at_y, c->d_bg_alpha_y);
  if (rc != OFDG_OK) return rc;
  c->bg_tab_w = TW; c->bg_tab_h = TH;
  return OFDG_OK;
}

// background_prep: render the 2W x 2H background textures of n samples into the slot's buffer on stream `s`; their
// records are in sl.d_bgprep already (written by the device sampler, or uploaded with the batch's other records)
//   stop: an event for the completion of the LAST kernel launched here (on that kernel's own packet)
static int prepare_backgrounds(ofdg_ctx* c, ofdg_ctx::Slot& sl, int n, bool records_resident, hipStream_t s, uint32_t* err, hipEvent_t stop) {
  const int W = c->prm.width, H = c->prm.height;
  if (!records_resident || !sl.d_bgprep.p || sl.d_bgprep.cap < (size_t)n) { c->err = "internal: background preparation without its records"; return OFDG_EINVAL; }
  HIP_OK(c, sl.d_bgtex.reserve((size_t)n * 4 * W * H));
  int cap_cw, cap_ch;
  bool fusable;
  bgprep_caps(c, &cap_cw, &cap_ch, &fusable);
  const bool staged = c->prm.background_prep == 1;
  constexpr int kBgPrepBlocks = 192;  // x 256 threads per sample, grid-stride over the (device-known) region
  if (staged) {
    int rct = ensure_bgprep_tables(c);
    if (rct != OFDG_OK) return rct;
  }
  if (!staged) {
    hipExtLaunchKernelGGL(bgprep_kernel, dim3((W * H + 255) / 256, n), dim3(256), 0, s, nullptr, stop, 0, sl.d_bgprep.p, W, H, sl.d_bgtex.p);
    HIP_OK(c, hipGetLastError());
    return OFDG_OK;
  }
  const DevResizeTabs T{c->d_bg_at_x.p, c->d_bg_alpha_x.p, c->d_bg_at_y.p, c->d_bg_alpha_y.p};
  if (fusable && n <= kPrepMaxSamples) {
    hipExtLaunchKernelGGL(bgprep_stream_kernel, dim3(kPrepGrid), dim3(64), (size_t)(n + 1) * sizeof(int), s, nullptr, stop, 0, sl.d_bgprep.p, T, W, H, n, cap_cw, cap_ch, sl.d_bgtex.p, err,
                          c->d_prep_paths);
    HIP_OK(c, hipGetLastError());
    return OFDG_OK;
  }
  HIP_OK(c, sl.d_bgC.reserve((size_t)n * cap_cw * cap_ch));
  hipLaunchKernelGGL(bgprep_rotcrop_kernel, dim3(kBgPrepBlocks, n), dim3(256), 0, s, sl.d_bgprep.p, T, W, H, cap_cw, cap_ch, sl.d_bgC.p,
                     err);
  hipExtLaunchKernelGGL(bgprep_resize_kernel, dim3(kBgPrepBlocks, n), dim3(256), 0, s, nullptr, stop, 0, sl.d_bgprep.p, T, W, H, cap_cw, cap_ch, sl.d_bgC.p,
                        sl.d_bgtex.p);
  HIP_OK(c, hipGetLastError());
  return OFDG_OK;
}

// realise on the host, stage, and copy the records of one batch into slot `sl`
//   `shared`: the slot may be rendered on other streams than `st` (a caller's slot): the upload is tracked by an event
static int upload_slot(ofdg_ctx* c, ofdg_ctx::Slot& sl, const ofdg_task* tasks, int n_tasks, const ofdg_blueprint* bps,
                       int n_bps, hipStream_t st, ofdg_ctx::Stage& stage, bool shared) {
  if (!c->pool && !c->pool_mixed) { c->err = "Could not open texture collection (no texture pool)"; return OFDG_ETEXTURES; }
  { int rcf = finalise_pool(c); if (rcf != OFDG_OK) return rcf; }
  { int rct = ensure_tex_table(c); if (rct != OFDG_OK) return rct; }
  RealizeConfig cfg{c->prm.width, c->prm.height, c->prm.mode, c->pool_n, c->pool_w, c->pool_h, c->prm.background_prep};
  cfg.pool_addr = (uint64_t)(uintptr_t)c->pool;
  cfg.tex_table = c->pool_mixed && c->prm.background_prep ? c->tex_table.data() : nullptr;
  cfg.fg_stride = c->fg_src.stride; cfg.fg_origin = c->fg_src.origin; cfg.bg_stride = c->bg_src.stride; cfg.bg_origin = c->bg_src.origin;
  // the previous copies out of this staging buffer must have left it
  if (stage.pending) {  // (normally long done - the chain has rendered three other batches since: a query costs 0.1 us, a wait 1 us)
    if (hipEventQuery(stage.free_ev) != hipSuccess) HIP_OK(c, hipEventSynchronize(stage.free_ev));
    stage.pending = false;
  }
  // a compose on another stream may still read the records this upload replaces
  if (sl.compose_pending && sl.compose_stream != st) {
    HIP_OK(c, hipStreamWaitEvent(st, sl.compose_event, 0));
    sl.compose_pending = false;
  }
  sl.res_samples = 0;
  int rc = realize_batch(cfg, tasks, n_tasks, bps, n_bps, &sl.batch, &c->err, &c->crop_server);
  if (rc != OFDG_OK) return rc;
  const RealizedBatch& B = sl.batch;
  const size_t n_shapes = B.shapes.size(), n_obj = B.objects.size();
  // the batch's records - shapes | objects | samples - travel as ONE copy into one allocation (three copies cost the host
  // 7 us a batch, one 2.5: tools/microbench/launch_cost.hip - most of what a batch of one sample costs)
  const size_t b_shapes = (n_shapes * sizeof(DevShape) + 255) & ~(size_t)255, b_obj = (n_obj * sizeof(DevObject) + 255) & ~(size_t)255,
               b_smp = ((size_t)n_tasks * sizeof(DevSample) + 255) & ~(size_t)255,
               b_prep = c->prm.background_prep ? B.bgprep.size() * sizeof(DevBgPrep) : 0;  // (the background preparation's records ride along)
  const size_t need = b_shapes + b_obj + b_smp + b_prep + 64;
  if (need > sl.d_rec.cap) {  // (growing waits for the device: a compose of this slot may still read the old arena)
    HIP_OK(c, hipDeviceSynchronize());
    HIP_OK(c, sl.d_rec.reserve(need));
  }
  HIP_OK(c, sl.d_shapes.alias((DevShape*)sl.d_rec.p, n_shapes));
  HIP_OK(c, sl.d_objects.alias((DevObject*)(sl.d_rec.p + b_shapes), n_obj));
  HIP_OK(c, sl.d_samples.alias((DevSample*)(sl.d_rec.p + b_shapes + b_obj), (size_t)n_tasks));
  if (b_prep) HIP_OK(c, sl.d_bgprep.alias((DevBgPrep*)(sl.d_rec.p + b_shapes + b_obj + b_smp), B.bgprep.size()));
  HIP_OK(c, sl.d_frames.reserve(n_shapes * 2));
  HIP_OK(c, sl.d_verts.reserve(n_shapes * 2 * kMaxVerts));
  {
    const int W = c->prm.width, H = c->prm.height;
    { int rcw = reserve_workspaces(c, n_shapes); if (rcw != OFDG_OK) return rcw; }
    HIP_OK(c, sl.d_items.reserve(n_shapes * 2 * (size_t)((H + kBandRows - 1) / kBandRows) * ((W + kChunkW - 1) / kChunkW) + 1));
    { int rcm = reserve_blockmask(c, sl, n_tasks); if (rcm != OFDG_OK) return rcm; }
    if (!sl.d_item_count) {
      HIP_OK(c, hipMalloc((void**)&sl.d_item_count, sizeof(int)));
      HIP_OK(c, hipMemset(sl.d_item_count, 0, sizeof(int)));
      HIP_OK(c, hipDeviceSynchronize());
    }
  }
  if (need > stage.bytes) {
    if (stage.h) HIP_OK(c, hipHostFree(stage.h));
    stage.h = nullptr;
    HIP_OK(c, hipHostMalloc(&stage.h, need * 2, hipHostMallocDefault));
    stage.bytes = need * 2;
  }
  char* hs = (char*)stage.h;
  if (n_shapes) std::memcpy(hs, B.shapes.data(), n_shapes * sizeof(DevShape));
  std::memcpy(hs + b_shapes, B.objects.data(), n_obj * sizeof(DevObject));
  std::memcpy(hs + b_shapes + b_obj, B.samples.data(), (size_t)n_tasks * sizeof(DevSample));
  if (b_prep) std::memcpy(hs + b_shapes + b_obj + b_smp, B.bgprep.data(), b_prep);
  HIP_OK(c, hipMemcpyAsync(sl.d_rec.p, hs, b_shapes + b_obj + b_smp + b_prep, hipMemcpyHostToDevice, st));
  if (!B.crops.empty()) {  // mode 9: this batch's crop table (+ upscaled background copies)
    const int W = c->prm.width, H = c->prm.height;
    const size_t crop_floats = (size_t)4 * (W + 1) * (H + 1), bg_floats = (size_t)4 * 2 * W * 2 * H;
    size_t n_bg = 0;
    for (const CropUse& u : B.crops) n_bg += u.background ? 1 : 0;
    HIP_OK(c, sl.d_croptab.reserve(B.crops.size()));
    if (n_bg) {
      HIP_OK(c, sl.d_bgwarp.reserve(n_bg * bg_floats));
      HIP_OK(c, sl.d_bgwarp_max.reserve(n_bg));
      HIP_OK(c, hipMemsetAsync(sl.d_bgwarp_max.p, 0, n_bg * sizeof(unsigned), st));
      { int rct = ensure_resize_tables(c); if (rct != OFDG_OK) return rct; }
    }
    std::vector<DevCropRef> tab(B.crops.size());
    size_t bg_at = 0;
    for (size_t k = 0; k < B.crops.size(); ++k) {
      const float* src = c->d_warp + (size_t)B.crops[k].crop * crop_floats;
      if (B.crops[k].background) {
        float* dst = sl.d_bgwarp.p + bg_at * bg_floats;
        hipLaunchKernelGGL(wf_resize2_kernel, dim3((2 * W * 2 * H + 255) / 256), dim3(256), 0, st, src, W + 1, H + 1, 2 * W, 2 * H,
                           c->d_rs_xi, c->d_rs_xa, c->d_rs_yi, c->d_rs_ya, dst, sl.d_bgwarp_max.p + bg_at);
        HIP_OK(c, hipGetLastError());
        tab[k] = make_crop_ref(dst, sl.d_bgwarp_max.p + bg_at, 2 * W, 2 * H);
        ++bg_at;
      } else {
        tab[k] = make_crop_ref(src, c->d_warp_max + B.crops[k].crop, W + 1, H + 1);
      }
    }
    // small table: a synchronous copy from pageable memory is fine here (upload path)
    HIP_OK(c, hipMemcpyAsync(sl.d_croptab.p, tab.data(), tab.size() * sizeof(DevCropRef), hipMemcpyHostToDevice, st));
    HIP_OK(c, hipStreamSynchronize(st));  // `tab` is a stack vector
  }
  if (c->prm.background_prep) {
    if (shared) {  // a caller's slot is prepared once, here, and rendered any number of times
      // (flags go into the word of the call that renders this batch next; that call must not clear them: word_reserved)
      if (c->word_reserved != c->ticket) { HIP_OK(c, take_err_word(c, c->ticket, st)); c->word_reserved = c->ticket; }
      int rcb = prepare_backgrounds(c, sl, n_tasks, /*records_resident=*/true, st, err_word(c, c->ticket));
      if (rcb != OFDG_OK) return rcb;
      sl.bgprep_pending = false;
    } else {
      sl.bgprep_pending = true;  // (a chain's private slot: behind raster, launch_prepare)
    }
  }
  HIP_OK(c, hipEventRecord(stage.free_ev, st));
  stage.pending = true;
  if (shared) {
    if (!sl.ev_uploaded) HIP_OK(c, hipEventCreateWithFlags(&sl.ev_uploaded, hipEventDisableTiming));
    HIP_OK(c, hipEventRecord(sl.ev_uploaded, st));
    sl.upload_pending = true;
    sl.upload_stream = st;
  } else {
    sl.upload_pending = false;  // (a chain's private slot: uploaded and rendered in order on the chain's own stream)
  }
  sl.res_samples = n_tasks;
  sl.res_shapes = (int)n_shapes;
  sl.res_objects = (int)n_obj;
  return OFDG_OK;
}

int ofdg_render(ofdg_ctx* c, const ofdg_task* tasks, int n_tasks, const ofdg_blueprint* bps, int n_bps,
                float* d_img0, float* d_img1, float* d_flow, void* stream) {
  if (!c || !tasks || !bps || n_tasks < 1 || !d_img0 || !d_img1 || !d_flow) {
    if (c) c->err = "ofdg_render: invalid argument";
    return OFDG_EINVAL;
  }
  stream = own_stream(c, stream);
  // the batch's records travel on the chain's own stream into its private slot
  ofdg_ctx::Chain& ch = take_chain(c);
  int rc = upload_slot(c, ch.slot, tasks, n_tasks, bps, n_bps, chain_stream(c, ch, (hipStream_t)stream), ch.stage, false);
  if (rc != OFDG_OK) return rc;
  return launch_resident(c, ch, ch.slot, d_img0, d_img1, d_flow, (hipStream_t)stream);
}

int ofdg_upload_slot(ofdg_ctx* c, int slot, const ofdg_task* tasks, int n_tasks, const ofdg_blueprint* bps, int n_bps,
                     void* stream) {
  if (!c || !tasks || !bps || n_tasks < 1 || slot < 0 || slot >= ofdg_ctx::kUserSlots) {
    if (c) c->err = "ofdg_upload_slot: invalid argument";
    return OFDG_EINVAL;
  }
  return upload_slot(c, c->slots[slot], tasks, n_tasks, bps, n_bps, (hipStream_t)stream, c->user_stage, true);
}

int ofdg_render_slot(ofdg_ctx* c, int slot, float* d_img0, float* d_img1, float* d_flow, void* stream) {
  if (!c || !d_img0 || !d_img1 || !d_flow || slot < 0 || slot >= ofdg_ctx::kUserSlots) return OFDG_EINVAL;
  if (c->slots[slot].res_samples <= 0) { c->err = "ofdg_render_slot: no batch is resident in this slot"; return OFDG_EINVAL; }
  stream = own_stream(c, stream);
  return launch_resident(c, take_chain(c), c->slots[slot], d_img0, d_img1, d_flow, (hipStream_t)stream);
}

int ofdg_render_resident(ofdg_ctx* c, float* d_img0, float* d_img1, float* d_flow, void* stream) {
  if (!c || !d_img0 || !d_img1 || !d_flow) return OFDG_EINVAL;
  if (!c->last_slot || c->last_slot->res_samples <= 0) { c->err = "ofdg_render_resident: nothing has been rendered yet"; return OFDG_EINVAL; }
  stream = own_stream(c, stream);
  // a chain's private slot is not tracked by events while only that chain uses it: let its owner drain first
  for (int k = 0; k < c->n_chains; ++k)
    if (&c->chains[k].slot == c->last_slot && !c->last_slot->compose_pending) HIP_OK(c, hipStreamSynchronize(c->chains[k].stream));
  return launch_resident(c, take_chain(c), *c->last_slot, d_img0, d_img1, d_flow, (hipStream_t)stream);
}

// size slot `sl` for n device-sampled samples: a fixed number of shape slots per sample
// (unused ones are typed 0 and produce no outline)
static int prepare_counter_slot(ofdg_ctx* c, ofdg_ctx::Slot& sl, int n) {
  if (c->prm.mode == 9) { int rcw = ensure_counter_croptab(c); if (rcw != OFDG_OK) return rcw; }
  { int rcf = finalise_pool(c); if (rcf != OFDG_OK) return rcf; }
  { int rct = ensure_tex_table(c); if (rct != OFDG_OK) return rct; }
  if (!c->pool && !c->pool_mixed) { c->err = "Could not open texture collection (no texture pool)"; return OFDG_ETEXTURES; }
  if (n < 1 || n > 512) { c->err = "counter sampler: batch must be 1..512 samples"; return OFDG_EINVAL; }
  const int W = c->prm.width, H = c->prm.height;
  // outline slots per sample: the worst case, so that no sample can run out (composites have up to 7 parts)
  const int max_objects = c->prm.num_objects > 0 ? std::min(c->prm.num_objects, kCsMaxObjects) : 24;
  const size_t shapes_cap = (size_t)n * (size_t)(c->prm.mode >= 6 ? max_objects * 7 : max_objects);
  const size_t n_obj = (size_t)n * (1 + kCsMaxObjects);
  HIP_OK(c, sl.d_shapes.reserve(shapes_cap));
  HIP_OK(c, sl.d_frames.reserve(shapes_cap * 2));
  HIP_OK(c, sl.d_verts.reserve(shapes_cap * 2 * kMaxVerts));
  HIP_OK(c, sl.d_objects.reserve(n_obj));
  HIP_OK(c, sl.d_samples.reserve(n));
  { int rcw = reserve_workspaces(c, shapes_cap); if (rcw != OFDG_OK) return rcw; }
  HIP_OK(c, sl.d_items.reserve(shapes_cap * 2 * (size_t)((H + kBandRows - 1) / kBandRows) * ((W + kChunkW - 1) / kChunkW) + 1));
  { int rcm = reserve_blockmask(c, sl, n); if (rcm != OFDG_OK) return rcm; }
  if (c->prm.background_prep) {
    HIP_OK(c, sl.d_bgprep.reserve(n));
    HIP_OK(c, sl.d_bgtex.reserve((size_t)n * 4 * W * H));
  }
  if (!sl.d_item_count) {
    HIP_OK(c, hipMalloc((void**)&sl.d_item_count, sizeof(int)));
    HIP_OK(c, hipMemset(sl.d_item_count, 0, sizeof(int)));
    HIP_OK(c, hipDeviceSynchronize());
  }
  sl.res_samples = n;
  sl.res_shapes = (int)shapes_cap;
  sl.res_objects = (int)n_obj;
  sl.batch.samples.clear();
  return OFDG_OK;
}

// Sample n_samples blueprints with global indices first_index.. on the DEVICE (counter
// sampler: a sample is a pure function of (seed, global index)) and render them.
int ofdg_forward_counter(ofdg_ctx* c, long long first_index, int n_samples, float* d_img0, float* d_img1, float* d_flow,
                         void* stream) {
  if (!c || !d_img0 || !d_img1 || !d_flow || first_index < 0) return OFDG_EINVAL;
  stream = own_stream(c, stream);
  // A sample is a pure function of (seed, global index): the chain samples, realises and prepares the batch on the device
  // and composes it, all in order on its stream.  Like the reference's prefetch thread (data_generation_layer.cpp:141-172)
  // the context runs AHEAD of its caller: after composing batch k it enqueues the preparation of the batches the caller's
  // sequence reaches next (first index + d x the stride of the last two calls, d = 1 .. lookahead) on the chains that
  // will compose them, so a call normally finds its batch prepared and only launches compose.  A call for other samples
  // than the prepared ones prepares its own (nothing is ever rendered from a stale preparation).
  const int k = (int)(c->next_chain % (unsigned)c->n_chains);
  ofdg_ctx::Chain& ch = take_chain(c);
  hipStream_t st = (hipStream_t)stream;
  auto prepare_on = [&](ofdg_ctx::Chain& cj, long long first, hipStream_t s_, bool hand_over) -> int {
    if (cj.prep.valid && cj.prep.slot == &cj.slot && cj.prep.first_index == first && cj.prep.n == n_samples) return OFDG_OK;
    int rc = prepare_counter_slot(c, cj.slot, n_samples);
    if (rc != OFDG_OK) return rc;
    return launch_prepare(c, cj, cj.slot, s_, first, hand_over);
  };
  int rc = prepare_on(ch, first_index, st, chain_stream(c, ch, st) != st);
  if (rc != OFDG_OK) return rc;
  rc = launch_compose(c, ch, d_img0, d_img1, d_flow, st);
  if (rc != OFDG_OK) return rc;
  // (the caller's batch is composed: from here on the call has succeeded, whatever happens to the batches prepared ahead)
  const long long prev_first = c->last_first;
  c->last_first = first_index;
  if (c->overlap && c->lookahead > 0) {
    const long long stride = (prev_first >= 0 && first_index > prev_first) ? first_index - prev_first
                                                                             : (long long)n_samples * std::max(1, c->prm.world_size);
    for (int d = 1; d <= std::min(c->lookahead, c->n_chains - 1); ++d) {
      ofdg_ctx::Chain& cj = c->chains[(k + d) % c->n_chains];
      // a preparation ahead that cannot be enqueued is dropped: the call that needs the batch prepares it itself and
      // reports the failure if it persists
      if (prepare_on(cj, first_index + (long long)d * stride, cj.stream, true) != OFDG_OK) { (void)discard_prepared(c, cj); break; }
      cj.prep.ahead = true;
    }
  }
  return OFDG_OK;
}

// The internal stream the NEXT render / forward call of this context works on (the chains take turns).
// A caller that passes it as that call's `stream` gets the outputs ordered on it and no cross-stream wait
// at all; with any other stream the compose kernel runs on that stream after one event (standard stream
// semantics: the compose kernels of consecutive calls then run one after the other).
int ofdg_num_chains(const ofdg_ctx* c) { return c ? c->n_chains : OFDG_EINVAL; }
void* ofdg_stream(ofdg_ctx* c) {
  if (!c) return nullptr;
  return (void*)c->chains[c->next_chain % (unsigned)c->n_chains].stream;
}

// Download the blueprints the counter sampler produces for samples first_index.. (tests):
// tasks[n], bps[n * 257] in the fixed per-sample layout (background, 32 object slots, 32 x 7
// component slots); unused slots have obj_type 0.
int ofdg_sample_counter(ofdg_ctx* c, long long first_index, int n_samples, ofdg_task* tasks, ofdg_blueprint* bps) {
  if (!c || !tasks || !bps || n_samples < 1 || first_index < 0) return OFDG_EINVAL;
  HIP_OK(c, hipDeviceSynchronize());
  HIP_OK(c, c->d_cs_bps.reserve((size_t)n_samples * kCsBlueprintsPerSample));
  HIP_OK(c, c->d_cs_nobj.reserve(n_samples));
  hipLaunchKernelGGL(cs_sample_kernel, dim3(n_samples * kCsGroups), dim3(64), 0, 0, c->cs_mode, first_index, n_samples,
                     c->prm.mode == 9 ? c->crop_server.n_crops : 0, c->d_cs_bps.p,
                     c->d_cs_nobj.p);
  HIP_OK(c, hipGetLastError());
  std::vector<int> nobj(n_samples);
  HIP_OK(c, hipMemcpy(bps, c->d_cs_bps.p, (size_t)n_samples * kCsBlueprintsPerSample * sizeof(ofdg_blueprint), hipMemcpyDeviceToHost));
  HIP_OK(c, hipMemcpy(nobj.data(), c->d_cs_nobj.p, n_samples * sizeof(int), hipMemcpyDeviceToHost));
  for (int s = 0; s < n_samples; ++s) {
    tasks[s].background = s * kCsBlueprintsPerSample;
    tasks[s].first_object = s * kCsBlueprintsPerSample + 1;
    tasks[s].n_objects = nobj[s];
    tasks[s].reserved = 0;
  }
  return OFDG_OK;
}

// The sharding rule: of the global sample stream, step `step` of rank `rank` renders the `batch` samples
// starting at this index.  The ranks' ranges tile the stream: every index belongs to exactly one (step, rank).
long long ofdg_shard_first_index(long long step, int batch, int world_size, int rank) {
  if (step < 0 || batch < 1 || world_size < 1 || rank < 0 || rank >= world_size) return -1;
  return step * (long long)batch * world_size + (long long)rank * batch;
}

int ofdg_forward(ofdg_ctx* c, float* d_img0, float* d_img1, float* d_flow, void* stream) {
  if (!c) return OFDG_EINVAL;
  if (c->prm.sampler == OFDG_SAMPLER_COUNTER) {
    // rank r owns global indices step*B*world + r*B + [0, B)
    const int B = c->prm.batch_size, world = c->prm.world_size, rank = c->prm.rank;
    if (B < 1 || rank < 0 || rank >= world) { c->err = "ofdg_forward: bad batch_size / rank"; return OFDG_EINVAL; }
    const long long first = ofdg_shard_first_index(c->step, B, world, rank);
    const int rc = ofdg_forward_counter(c, first, B, d_img0, d_img1, d_flow, stream);
    if (rc == OFDG_OK) c->step++;  // (a failed call does not advance the checkpoint counter)
    return rc;
  }
  const int B = c->prm.batch_size, world = c->prm.world_size, rank = c->prm.rank;
  if (B < 1 || rank < 0 || rank >= world) { c->err = "ofdg_forward: bad batch_size / rank"; return OFDG_EINVAL; }
  // every rank walks the identical sequential stream and keeps its own block of
  // B consecutive tasks out of each B*world (disjoint shards, no communication)
  c->fw_bps.clear();
  c->fw_tasks.assign((size_t)B * world, ofdg_task());
  for (int i = 0; i < B * world; ++i) {
    int rc = c->sampler->next_task(&c->fw_bps, &c->fw_tasks[i], &c->err);
    if (rc != OFDG_OK) return rc;
  }
  const int rc = ofdg_render(c, c->fw_tasks.data() + (size_t)rank * B, B, c->fw_bps.data(), (int)c->fw_bps.size(), d_img0, d_img1,
                             d_flow, stream);
  // the streams have moved on either way; the batch counts once it is in flight
  if (rc == OFDG_OK) c->step++;
  else ofdg_set_step(c, c->step);  // rewind the streams (and the crop server) to the start of this batch
  return rc;
}

// Checkpoint / resume of ofdg_forward (the reference has none: a restarted job replays its streams from the
// seeds): the step counter is the whole state.  Counter sampler: the next call renders the indices of `step`.
// Reference-stream sampler: the 45 streams are rebuilt and step * batch_size * world_size tasks are drawn and
// dropped (host, ~30 us per task); in mode 9 the crop server (CropGenerator::get_crop's serving order, WF:516-538)
// is replayed as well: one crop per deforming background / top-level object of THIS rank's tasks.
long long ofdg_get_step(const ofdg_ctx* c) { return c ? c->step : -1; }
int ofdg_set_step(ofdg_ctx* c, long long step) {
  if (!c || step < 0) return OFDG_EINVAL;
  if (c->prm.sampler != OFDG_SAMPLER_COUNTER) {
    const long long n = step * (long long)std::max(c->prm.batch_size, 1) * c->prm.world_size;
    c->sampler.reset(new RefSampler(c->prm.mode, c->prm.width, c->prm.height, c->prm.num_objects));
    std::vector<ofdg_blueprint> bps;
    ofdg_task t;
    const int B = std::max(c->prm.batch_size, 1), world = c->prm.world_size, rank = c->prm.rank;
    const bool crops = c->prm.mode == 9 && c->crop_server.n_crops > 0;
    if (crops) { c->crop_server.head = 0; c->crop_server.counter = 0; }
    for (long long i = 0; i < n; ++i) {
      bps.clear();
      int rc = c->sampler->next_task(&bps, &t, &c->err);
      if (rc != OFDG_OK) return rc;
      if (crops && (i / B) % world == rank) {  // realize_batch serves one crop per deforming background / object
        if (bps[t.background].do_warpfield_deformation) (void)c->crop_server.get();
        for (int k = 0; k < t.n_objects; ++k)
          if (bps[t.first_object + k].do_warpfield_deformation) (void)c->crop_server.get();
      }
    }
  }
  c->step = step;
  return OFDG_OK;
}

int ofdg_synchronize(ofdg_ctx* c, void* stream) {
  if (!c) return OFDG_EINVAL;
  if (stream == OFDG_STREAM_OWN) stream = nullptr;  // (every internal stream is waited for below)
  HIP_OK(c, hipStreamSynchronize((hipStream_t)stream));
  for (int k = 0; k < c->n_chains; ++k) {
    HIP_OK(c, hipStreamSynchronize(c->chains[k].stream));
    if (c->chains[k].done_pending) HIP_OK(c, hipEventSynchronize(c->chains[k].ev_done));  // (a compose on another caller stream)
  }
  uint32_t words[ofdg_ctx::kErrWords];
  HIP_OK(c, hipMemcpy(words, c->d_err, sizeof(words), hipMemcpyDeviceToHost));
  uint32_t e = 0;
  long long first_bad = -1;
  for (int i = 0; i < ofdg_ctx::kErrWords; ++i)
    if (words[i]) {
      e |= words[i];
      // the most recent call that used this word
      const long long q = c->ticket - 1 - ((c->ticket - 1 - i) % ofdg_ctx::kErrWords + ofdg_ctx::kErrWords) % ofdg_ctx::kErrWords;
      if (first_bad < 0 || q < first_bad) first_bad = q;
    }
  if (e) HIP_OK(c, hipMemset(c->d_err, 0, sizeof(words)));
  if (c->word_reserved < 0) std::fill(c->word_clean, c->word_clean + ofdg_ctx::kErrWords, true);  // (nothing in flight, every word read)
  if (e) {
    c->err = err_text(e) + " (first in batch " + std::to_string(first_bad) + " of this context, or one " + std::to_string(ofdg_ctx::kErrWords) + " calls earlier)";
    return OFDG_ECAPACITY;
  }
  return OFDG_OK;
}

// The device-side error flags without waiting for anything in flight (a prefetch ring checks them when it hands
// a finished batch over; flags of younger batches are reported at their own hand-over at the latest).
static int poll_words(ofdg_ctx* c, int first, int n, uint32_t* flags) {
  if (!c->h_err) HIP_OK(c, hipHostMalloc((void**)&c->h_err, sizeof(uint32_t), hipHostMallocMapped));
  if (!c->err_stream) HIP_OK(c, hipStreamCreateWithFlags(&c->err_stream, hipStreamNonBlocking));
  // read AND clear in one atomic exchange per word (younger batches are still running and may raise a flag at any time: a
  // copy followed by a memset would lose what is raised in between)
  uint32_t* h_dev = nullptr;
  HIP_OK(c, hipHostGetDevicePointer((void**)&h_dev, c->h_err, 0));
  hipLaunchKernelGGL(err_exchange_kernel, dim3(1), dim3(64), 0, c->err_stream, c->d_err + first, n, h_dev);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipStreamSynchronize(c->err_stream));
  *flags = *c->h_err;
  return OFDG_OK;
}
int ofdg_poll_errors(ofdg_ctx* c) {
  if (!c) return OFDG_EINVAL;
  uint32_t e = 0;
  int rc = poll_words(c, 0, ofdg_ctx::kErrWords, &e);
  if (rc != OFDG_OK) return rc;
  if (e) { c->err = err_text(e); return OFDG_ECAPACITY; }
  return OFDG_OK;
}

// The flags of ONE batch (ticket = ofdg_last_ticket() right after the call that rendered it), without waiting for anything:
// what a prefetch ring asks when it hands that batch over, having waited for the batch's own completion event.  The word
// is cleared.  Tickets older than kErrWords calls are gone (OFDG_EINVAL).
long long ofdg_last_ticket(const ofdg_ctx* c) { return c ? c->last_ticket : -1; }
int ofdg_poll_errors_of(ofdg_ctx* c, long long ticket) {
  if (!c) return OFDG_EINVAL;
  if (ticket < 0 || ticket >= c->ticket || ticket + ofdg_ctx::kErrWords <= c->ticket) { c->err = "ofdg_poll_errors_of: no such batch (tickets of the last " + std::to_string(ofdg_ctx::kErrWords) + " calls are kept)"; return OFDG_EINVAL; }
  uint32_t e = 0;
  c->asked_by_ticket = true;
  int rc = poll_words(c, (int)(ticket % ofdg_ctx::kErrWords), 1, &e);
  if (rc != OFDG_OK) return rc;
  c->word_clean[(size_t)(ticket % ofdg_ctx::kErrWords)] = true;  // (read and cleared; the caller waited for this batch before asking)
  if (e) { c->err = "batch " + std::to_string(ticket) + ": " + err_text(e); return OFDG_ECAPACITY; }
  return OFDG_OK;
}

// ---- mode 9 warp fields -----------------------------------------------------------------------------
static int warp_alloc(ofdg_ctx* c, int n_crops) {
  const size_t crop_floats = (size_t)4 * (c->prm.width + 1) * (c->prm.height + 1);
  HIP_OK(c, hipDeviceSynchronize());
  drop_counter_croptab(c);
  if (c->d_warp) { HIP_OK(c, hipFree(c->d_warp)); c->d_warp = nullptr; }
  if (c->d_warp_max) { HIP_OK(c, hipFree(c->d_warp_max)); c->d_warp_max = nullptr; }
  HIP_OK(c, hipMalloc((void**)&c->d_warp, (size_t)n_crops * crop_floats * sizeof(float)));
  HIP_OK(c, hipMalloc((void**)&c->d_warp_max, (size_t)n_crops * sizeof(unsigned)));
  HIP_OK(c, hipMemset(c->d_warp_max, 0, (size_t)n_crops * sizeof(unsigned)));
  c->crop_server = CropServer();
  c->crop_server.n_crops = n_crops;
  return OFDG_OK;
}

// Replaces WarpFields::CropGenerator (WF:469-641): n_fields big fields of side 3*max(W,H)
// are generated on the device from seeded displacer lists (seed, seed+1, ...), composed
// 17 times with themselves (x 2^17), cleaned, and cut into (W+1) x (H+1) crops.
int ofdg_warp_generate(ofdg_ctx* c, int n_fields, uint32_t seed) {
  if (!c || n_fields < 1) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }  // (batches prepared ahead read the old pool / crops)
  const int W = c->prm.width, H = c->prm.height;
  const int S = std::max(W, H) * 3;
  std::vector<std::pair<int, int>> org;  // crop grid (WF:617-633)
  for (int y = H / 4; y < S - 5 * H / 4; y += H / 3)
    for (int x = W / 4; x < S - 5 * W / 4; x += W / 3) org.push_back({x, y});
  if (org.empty()) { c->err = "frame too small for warp crops"; return OFDG_EINVAL; }
  int rc = warp_alloc(c, n_fields * (int)org.size());
  if (rc != OFDG_OK) return rc;
  const size_t n = (size_t)S * S;
  float *fa = nullptr, *fb = nullptr;
  uint8_t* flagged = nullptr;
  DevDisplacer* dd = nullptr;
  HIP_OK(c, hipMalloc((void**)&fa, 4 * n * sizeof(float)));
  HIP_OK(c, hipMalloc((void**)&fb, 4 * n * sizeof(float)));
  HIP_OK(c, hipMalloc((void**)&flagged, 2 * n));
  const int cw = W + 1, ch = H + 1;
  const size_t crop_floats = (size_t)4 * cw * ch;
  const int blocks = (int)((n + 255) / 256);
  for (int f = 0; f < n_fields; ++f) {
    std::vector<DevDisplacer> disp = make_device_displacers(make_displacer_params(W, H, seed + (uint32_t)f));
    if (dd) { HIP_OK(c, hipFree(dd)); dd = nullptr; }
    HIP_OK(c, hipMalloc((void**)&dd, std::max<size_t>(1, disp.size()) * sizeof(DevDisplacer)));
    HIP_OK(c, hipMemcpy(dd, disp.data(), disp.size() * sizeof(DevDisplacer), hipMemcpyHostToDevice));
    HIP_OK(c, hipMemset(flagged, 0, 2 * n));
    hipLaunchKernelGGL(wf_sample_kernel, dim3(blocks), dim3(256), 0, 0, dd, (int)disp.size(), S, fa);
    float *from = fa, *to = fb;
    for (int iter = 17; iter > 0; --iter) {  // WF:366, 406
      hipLaunchKernelGGL(wf_compose_kernel, dim3(blocks, 2), dim3(256), 0, 0, from, to, S, flagged);
      std::swap(from, to);
    }
    hipLaunchKernelGGL(wf_finish_kernel, dim3(blocks, 2), dim3(256), 0, 0, from, S, flagged);
    for (size_t k = 0; k < org.size(); ++k) {
      const size_t ci = (size_t)f * org.size() + k;
      hipLaunchKernelGGL(wf_crop_kernel, dim3((cw * ch + 255) / 256), dim3(256), 0, 0, from, S, org[k].first, org[k].second, cw, ch,
                         c->d_warp + ci * crop_floats, c->d_warp_max + ci);
    }
    HIP_OK(c, hipGetLastError());
    HIP_OK(c, hipDeviceSynchronize());
  }
  HIP_OK(c, hipFree(fa)); HIP_OK(c, hipFree(fb)); HIP_OK(c, hipFree(flagged));
  if (dd) HIP_OK(c, hipFree(dd));
  return OFDG_OK;
}

// Install caller-provided crops: n x 4 planes (flow x, flow y, iflow x, iflow y) of (H+1)*(W+1) floats.
int ofdg_warp_upload(ofdg_ctx* c, const float* crops, int n) {
  if (!c || !crops || n < 1) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }  // (batches prepared ahead read the old pool / crops)
  int rc = warp_alloc(c, n);
  if (rc != OFDG_OK) return rc;
  const size_t plane = (size_t)(c->prm.width + 1) * (c->prm.height + 1), crop_floats = 4 * plane;
  {  // the kernels read a crop as two planes of interleaved pairs: (flow x, flow y), (iflow x, iflow y)
    std::vector<float> il((size_t)n * crop_floats);
    for (int k = 0; k < n; ++k)
      for (int f = 0; f < 4; ++f) {
        const float* src = crops + (size_t)k * crop_floats + (size_t)f * plane;
        float* dst = il.data() + (size_t)k * crop_floats + (size_t)(f >> 1) * 2 * plane + (f & 1);
        for (size_t i = 0; i < plane; ++i) dst[2 * i] = src[i];
      }
    HIP_OK(c, hipMemcpy(c->d_warp, il.data(), il.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  std::vector<unsigned> mx(n, 0u);
  for (int k = 0; k < n; ++k) {
    float m = 0.f;
    const float* p = crops + (size_t)k * crop_floats + 2 * plane;
    for (size_t i = 0; i < 2 * plane; ++i) if (p[i] == p[i]) m = std::max(m, std::fabs(p[i]));
    std::memcpy(&mx[k], &m, sizeof(float));
  }
  HIP_OK(c, hipMemcpy(c->d_warp_max, mx.data(), (size_t)n * sizeof(unsigned), hipMemcpyHostToDevice));
  return OFDG_OK;
}

int ofdg_warp_info(const ofdg_ctx* c, int* n_crops, int* w, int* h) {
  if (!c) return OFDG_EINVAL;
  if (n_crops) *n_crops = c->crop_server.n_crops;
  if (w) *w = c->prm.width + 1;
  if (h) *h = c->prm.height + 1;
  return OFDG_OK;
}

int ofdg_warp_download(ofdg_ctx* c, int index, float* crop) {
  if (!c || !crop || index < 0 || index >= c->crop_server.n_crops) return OFDG_EINVAL;
  const size_t crop_floats = (size_t)4 * (c->prm.width + 1) * (c->prm.height + 1);
  HIP_OK(c, hipDeviceSynchronize());
  std::vector<float> il(crop_floats);
  HIP_OK(c, hipMemcpy(il.data(), c->d_warp + (size_t)index * crop_floats, crop_floats * sizeof(float), hipMemcpyDeviceToHost));
  const size_t plane = crop_floats / 4;  // (device layout: interleaved pairs; the API's: four planes)
  for (int f = 0; f < 4; ++f) {
    const float* src = il.data() + (size_t)(f >> 1) * 2 * plane + (f & 1);
    for (size_t i = 0; i < plane; ++i) crop[(size_t)f * plane + i] = src[2 * i];
  }
  return OFDG_OK;
}

// Displacer draws of one big field (host only): n x 9 doubles {type, p0, p1, p2, support cx, cy, sx, sy, angle}.
int ofdg_host_displacers(int width, int height, uint32_t seed, double* out, int cap) {
  std::vector<DisplacerParams> d = make_displacer_params(width, height, seed);
  if ((int)d.size() > cap || !out) return -(int)d.size();
  std::memcpy(out, d.data(), d.size() * sizeof(DisplacerParams));
  return (int)d.size();
}

// ---- inspection ---------------------------------------------------------------------------------
static inline int iround_h(double v) { return int((v < 0.0) ? v - 0.5 : v + 0.5); }

// rasterise ONE outline given as 24.8 vertices over the whole frame (slot 0 serves as scratch)
static int debug_rasterize_verts(ofdg_ctx* c, const std::vector<int2>& v, int n, uint8_t* coverage_host) {
  const int W = c->prm.width, H = c->prm.height;
  int minx = 0x7fffffff, miny = 0x7fffffff, maxx = -0x7fffffff - 1, maxy = -0x7fffffff - 1;
  for (int i = 0; i < n; ++i) {
    minx = std::min(minx, v[i].x); maxx = std::max(maxx, v[i].x);
    miny = std::min(miny, v[i].y); maxy = std::max(maxy, v[i].y);
  }
  DevShapeFrame f = DevShapeFrame();
  f.n_verts = n;
  int x0 = minx >> 8, y0 = miny >> 8, x1 = maxx >> 8, y1 = maxy >> 8;
  if (n < 2 || x1 < 0 || y1 < 0 || x0 > W - 1 || y0 > H - 1) { x0 = 1; x1 = 0; y0 = 1; y1 = 0; }
  else { x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, W - 1); y1 = std::min(y1, H - 1); }
  f.x0 = x0; f.y0 = y0; f.x1 = x1; f.y1 = y1;
  HIP_OK(c, hipDeviceSynchronize());
  ofdg_ctx::Slot& sl = c->slots[0];
  HIP_OK(c, sl.d_frames.reserve(2));
  HIP_OK(c, sl.d_verts.reserve(2 * kMaxVerts));
  HIP_OK(c, c->chains[0].cov.reserve((size_t)2 * W * H + 16));
  HIP_OK(c, hipMemcpy(sl.d_frames.p, &f, sizeof(f), hipMemcpyHostToDevice));
  HIP_OK(c, hipMemcpy(sl.d_verts.p, v.data(), sizeof(int2) * kMaxVerts, hipMemcpyHostToDevice));
  HIP_OK(c, hipMemset(c->chains[0].cov.p, 0xAB, (size_t)W * H));  // poison: every byte must be written
  const int bands = (H + kBandRows - 1) / kBandRows;
  std::vector<int4> items;
  for (int b = 0; b < bands; ++b)
    for (int cx = 0; cx < W; cx += kChunkW) items.push_back(make_int4(0, b, cx, std::min(cx + kChunkW - 1, W - 1)));
  const int n_items = (int)items.size();
  HIP_OK(c, sl.d_items.reserve(items.size()));
  if (!sl.d_item_count) HIP_OK(c, hipMalloc((void**)&sl.d_item_count, sizeof(int)));
  HIP_OK(c, hipMemcpy(sl.d_items.p, items.data(), sizeof(int4) * items.size(), hipMemcpyHostToDevice));
  HIP_OK(c, hipMemcpy(sl.d_item_count, &n_items, sizeof(int), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(raster_kernel, dim3(64 * 4 / kRasterWaves), dim3(64 * kRasterWaves), 0, 0, sl.d_frames.p, sl.d_items.p, sl.d_item_count, sl.d_verts.p,
                     W, H, c->chains[0].cov.p, nullptr, 0, (unsigned long long*)nullptr);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipMemcpy(coverage_host, c->chains[0].cov.p, (size_t)W * H, hipMemcpyDeviceToHost));
  HIP_OK(c, hipMemset(sl.d_item_count, 0, sizeof(int)));
  sl.res_samples = 0;  // (slot 0 served as scratch)
  c->last_slot = nullptr;
  return OFDG_OK;
}

int ofdg_debug_rasterize(ofdg_ctx* c, const double* xy, int n, uint8_t* coverage_host) {
  if (!c || !xy || !coverage_host || n < 1 || n > kMaxVerts) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }
  std::vector<int2> v(kMaxVerts);
  for (int i = 0; i < n; ++i) v[i] = make_int2(iround_h(xy[2 * i] * 256.0), iround_h(xy[2 * i + 1] * 256.0));
  return debug_rasterize_verts(c, v, n, coverage_host);
}

// A path with curve3 segments (types: 1 line_to, 3 curve3 control point followed by its end point; entry 0 is the
// move_to vertex) through the DEVICE's flattening (path_verts / flatten_curve3, what geom_kernel runs) and rasteriser.
int ofdg_debug_rasterize_path(ofdg_ctx* c, const double* xy, const int* types, int n, uint8_t* coverage_host) {
  if (!c || !xy || !types || !coverage_host || n < 1 || n > 64) return OFDG_EINVAL;
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }
  HIP_OK(c, hipDeviceSynchronize());
  double* d_xy = nullptr; int* d_ty = nullptr; int2* d_v = nullptr; int* d_n = nullptr;
  HIP_OK(c, hipMalloc((void**)&d_xy, sizeof(double) * 2 * n));
  HIP_OK(c, hipMalloc((void**)&d_ty, sizeof(int) * n));
  HIP_OK(c, hipMalloc((void**)&d_v, sizeof(int2) * kMaxVerts));
  HIP_OK(c, hipMalloc((void**)&d_n, sizeof(int)));
  HIP_OK(c, hipMemcpy(d_xy, xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
  HIP_OK(c, hipMemcpy(d_ty, types, sizeof(int) * n, hipMemcpyHostToDevice));
  HIP_OK(c, hipMemset(d_v, 0, sizeof(int2) * kMaxVerts));
  uint32_t* const dbg_err = c->d_err + ofdg_ctx::kErrWords;  // (the debug entry points' own word: no batch inherits it)
  hipLaunchKernelGGL(debug_path_kernel, dim3(1), dim3(64), 0, 0, d_xy, d_ty, n, d_v, d_n, dbg_err);
  HIP_OK(c, hipGetLastError());
  uint32_t dbg_flags = 0;
  HIP_OK(c, hipMemcpy(&dbg_flags, dbg_err, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (dbg_flags) HIP_OK(c, hipMemset(dbg_err, 0, sizeof(uint32_t)));
  std::vector<int2> v(kMaxVerts);
  int nv = 0;
  HIP_OK(c, hipMemcpy(v.data(), d_v, sizeof(int2) * kMaxVerts, hipMemcpyDeviceToHost));
  HIP_OK(c, hipMemcpy(&nv, d_n, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(d_xy); (void)hipFree(d_ty); (void)hipFree(d_v); (void)hipFree(d_n);
  if (dbg_flags) { c->err = "debug_rasterize_path: " + err_text(dbg_flags); return OFDG_ECAPACITY; }
  if (nv < 1 || nv > kMaxVerts) { c->err = "debug_rasterize_path: the flattened outline has " + std::to_string(nv) + " vertices"; return OFDG_ECAPACITY; }
  return debug_rasterize_verts(c, v, nv, coverage_host);
}

// The DEVICE's span interpolator (make_row + dda_at, what the texture warps run): (x, y) in 24.8 fixed point of every
// pixel of `rows` output rows of length `len` under the inverse affine inv[6] (AGG member order), before the -128.
int ofdg_debug_dda_rows(ofdg_ctx* c, const double* inv, int rows, int len, int* out_xy) {
  if (!c || !inv || !out_xy || rows < 1 || len < 1 || (size_t)rows * len > (1u << 24)) return OFDG_EINVAL;
  int2* d = nullptr;
  HIP_OK(c, hipMalloc((void**)&d, sizeof(int2) * (size_t)rows * len));
  const Mat m{inv[0], inv[1], inv[2], inv[3], inv[4], inv[5]};
  hipLaunchKernelGGL(debug_dda_kernel, dim3((rows * len + 255) / 256), dim3(256), 0, 0, m, rows, len, d);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipMemcpy(out_xy, d, sizeof(int2) * (size_t)rows * len, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return OFDG_OK;
}

int ofdg_debug_num_shapes(ofdg_ctx* c, int sample) {
  if (!c || !c->last_slot) return OFDG_EINVAL;
  const ofdg_ctx::Slot& sl = *c->last_slot;
  if (sl.res_samples <= 0 || sample < 0 || sample >= (int)sl.batch.samples.size()) return OFDG_EINVAL;
  return sl.batch.samples[sample].n_shapes;
}

int ofdg_debug_coverage(ofdg_ctx* c, int sample, int shape, int frame, uint8_t* coverage_host) {
  if (!c || !coverage_host || frame < 0 || frame > 1 || !c->last_slot || c->last_slot->res_samples <= 0) return OFDG_EINVAL;
  const ofdg_ctx::Slot& sl = *c->last_slot;
  if (sample < 0 || sample >= (int)sl.batch.samples.size()) return OFDG_EINVAL;
  const DevSample& s = sl.batch.samples[sample];
  if (shape < 0 || shape >= s.n_shapes) return OFDG_EINVAL;
  const int W = c->prm.width, H = c->prm.height;
  const size_t sf = (size_t)(s.first_shape + shape) * 2 + frame;
  HIP_OK(c, hipDeviceSynchronize());
  DevShapeFrame f;
  HIP_OK(c, hipMemcpy(&f, sl.d_frames.p + sf, sizeof(f), hipMemcpyDeviceToHost));
  std::vector<uint8_t> tmp((size_t)W * H);
  HIP_OK(c, hipMemcpy(tmp.data(), c->chains[c->last_chain].cov.p + sf * W * H, tmp.size(), hipMemcpyDeviceToHost));
  std::memset(coverage_host, 0, tmp.size());
  for (int y = f.y0; y <= f.y1; ++y)
    for (int x = f.x0; x <= f.x1; ++x) coverage_host[(size_t)y * W + x] = tmp[(size_t)y * W + x];
  return OFDG_OK;
}

// number of raster work items left by the last launch (diagnostics)
int ofdg_debug_item_count(ofdg_ctx* c) {
  if (!c || !c->last_slot || !c->last_slot->d_item_count) return OFDG_EINVAL;
  int n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return OFDG_EHIP;
  if (hipMemcpy(&n, c->last_slot->d_item_count, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return OFDG_EHIP;
  return n;
}

// How bgprep_stream_kernel walked the last batch: the number of its tiles (only known on the device: a sample's read region
// depends on its background motion) and of the workgroups that shared them grid-stride - a test that means to compare the
// SECOND, third ... tile of a workgroup with the oracle asserts tiles > k * workgroups.  0 tiles: the batch took another form
// of the preparation (background_prep != 1, pool images smaller than 2W x 2H, more samples than the one-launch form numbers).
int ofdg_debug_bgprep_tiles(ofdg_ctx* c, int* tiles, int* workgroups) {
  if (!c || !tiles || !workgroups || !c->last_slot) return OFDG_EINVAL;
  const ofdg_ctx::Slot& sl = *c->last_slot;
  *tiles = 0; *workgroups = kPrepGrid;
  int cap_cw, cap_ch;
  bool fusable;
  bgprep_caps(c, &cap_cw, &cap_ch, &fusable);
  const int n = sl.res_samples;
  if (c->prm.background_prep != 1 || !fusable || n > kPrepMaxSamples || n < 1 || !sl.d_bgprep.p) return OFDG_OK;
  HIP_OK(c, hipDeviceSynchronize());
  std::vector<DevBgPrep> rec((size_t)n);
  HIP_OK(c, hipMemcpy(rec.data(), sl.d_bgprep.p, rec.size() * sizeof(DevBgPrep), hipMemcpyDeviceToHost));
  for (const DevBgPrep& q : rec)
    if (prep_sample_fits(q, cap_cw, cap_ch)) *tiles += prep_tile_cols(q) * prep_tile_rows(q);
  return OFDG_OK;
}

// Which forms of bgprep_stream_kernel rendered the tiles since the last call (the first call switches the counting on and returns
// zeros): counts[rotation + 3 * resize], see the kernel.  A guard for the tests: parity cannot tell a batch whose tiles all fell
// back to the general forms from one that took the fast ones.
int ofdg_debug_bgprep_paths(ofdg_ctx* c, unsigned* counts9) {
  if (!c || !counts9) return OFDG_EINVAL;
  HIP_OK(c, hipDeviceSynchronize());
  if (!c->d_prep_paths) {
    HIP_OK(c, hipMalloc((void**)&c->d_prep_paths, 9 * sizeof(uint32_t)));
    HIP_OK(c, hipMemset(c->d_prep_paths, 0, 9 * sizeof(uint32_t)));
  }
  HIP_OK(c, hipMemcpy(counts9, c->d_prep_paths, 9 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP_OK(c, hipMemset(c->d_prep_paths, 0, 9 * sizeof(uint32_t)));
  return OFDG_OK;
}

int ofdg_set_profiling(ofdg_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return OFDG_EINVAL;
  HIP_OK(c, hipDeviceSynchronize());
  { int rcd = discard_all_prepared(c); if (rcd != OFDG_OK) return rcd; }
  HIP_OK(c, hipDeviceSynchronize());
  c->profiling = mode;
  c->ev_count = 0; c->ev_alloc = 0;
  c->launch_count = 0;
  c->ev_stride = (mode == 1) ? 4 : 1;  // mode 1 samples every 4th launch (no marker packets: two completion signals per sampled launch)
  if (mode && c->ev.empty()) {
    c->ev_sets = 256;
    c->ev.resize((size_t)c->ev_sets * 6);
    for (auto& e : c->ev) HIP_OK(c, hipEventCreate(&e));
  }
  c->ev_composed.assign((size_t)c->ev_sets, 0);
  c->ev_bgprep.assign((size_t)c->ev_sets, 0);
  return OFDG_OK;
}

// Average device time (ms) per launch of one kernel over the launches recorded since
// ofdg_set_profiling (at most the last 256), from HIP events attached to the kernels' dispatch
// packets on their launch streams: completion of the predecessor .. completion of the kernel.
int ofdg_kernel_ms(ofdg_ctx* c, const char* kernel, float* ms) {
  if (!c || !kernel || !ms) return OFDG_EINVAL;
  int i = -1;
  if (!std::strcmp(kernel, "geom")) i = 0;
  else if (!std::strcmp(kernel, "raster")) i = 1;
  else if (!std::strcmp(kernel, "compose")) i = 2;
  else if (!std::strcmp(kernel, "background_prep")) i = 3;  // (timed where it runs behind raster: not for a caller's slot prepared at its upload)
  if (i < 0) { c->err = "unknown kernel name"; return OFDG_EINVAL; }
  if (!c->profiling || c->ev_count == 0 || (i < 2 && c->profiling != 2)) {
    c->err = "no profiled launch of that kernel yet (ofdg_set_profiling)";
    return OFDG_EINVAL;
  }
  int n = 0;
  double acc = 0;
  for (int k = 0; k < c->ev_sets; ++k) {
    if (!c->ev_composed[(size_t)k]) continue;  // (never handed out, or its prepared batch was discarded before compose)
    if (i == 3 && !c->ev_bgprep[(size_t)k]) continue;
    ++n;
    hipEvent_t* ev = &c->ev[(size_t)k * 6];
    HIP_OK(c, hipEventSynchronize(ev[5]));
    float t = 0;
    if (i == 3) {  // completion of raster .. completion of the (last) preparation kernel
      HIP_OK(c, hipEventElapsedTime(&t, ev[3], ev[c->profiling == 2 ? 2 : 4]));
      acc += t;
      continue;
    }
    // geom: its own start .. its end; raster: end of geom .. its end; compose: its own start (profiling 2) or the end of the
    // last preparation kernel (profiling 1) .. its end
    HIP_OK(c, hipEventElapsedTime(&t, ev[i == 0 ? 0 : (i == 2 ? 4 : 1)], ev[2 * i + 1]));
    acc += t;
  }
  if (n == 0) { c->err = "no profiled launch of that kernel yet (ofdg_set_profiling)"; return OFDG_EINVAL; }
  *ms = (float)(acc / n);
  return OFDG_OK;
}

// Exhaustive device tables of the per-byte formulas (tests): add/sub [256*256],
// aa [256], blend(d, s_fixed, m) [256*256] indexed d*256+m.
int ofdg_debug_tables(ofdg_ctx* c, uint8_t* add_tbl, uint8_t* sub_tbl, uint8_t* aa_tbl, uint8_t* blend_tbl, int s_fixed) {
  if (!c) return OFDG_EINVAL;
  uint8_t* d = nullptr;
  HIP_OK(c, hipMalloc((void**)&d, 3 * 65536 + 256));
  hipLaunchKernelGGL(tables_kernel, dim3(256), dim3(256), 0, 0, d, d + 65536, d + 3 * 65536, d + 2 * 65536, s_fixed);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipMemcpy(add_tbl, d, 65536, hipMemcpyDeviceToHost));
  HIP_OK(c, hipMemcpy(sub_tbl, d + 65536, 65536, hipMemcpyDeviceToHost));
  HIP_OK(c, hipMemcpy(blend_tbl, d + 2 * 65536, 65536, hipMemcpyDeviceToHost));
  HIP_OK(c, hipMemcpy(aa_tbl, d + 3 * 65536, 256, hipMemcpyDeviceToHost));
  HIP_OK(c, hipFree(d));
  return OFDG_OK;
}

// include/ofdg_detmath.h on the device: n angles -> sin, cos; m floats -> expf (host arrays)
int ofdg_debug_detmath(ofdg_ctx* c, const double* angles, int n, double* sin_out, double* cos_out, const float* x, int m, float* expf_out) {
  if (!c || n < 0 || m < 0 || (n > 0 && (!angles || !sin_out || !cos_out)) || (m > 0 && (!x || !expf_out))) return OFDG_EINVAL;
  double* d = nullptr;
  float* f = nullptr;
  const int k = std::max(std::max(n, m), 1);
  HIP_OK(c, hipMalloc((void**)&d, (size_t)3 * k * sizeof(double)));
  HIP_OK(c, hipMalloc((void**)&f, (size_t)2 * k * sizeof(float)));
  if (n) HIP_OK(c, hipMemcpy(d, angles, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  if (m) HIP_OK(c, hipMemcpy(f, x, (size_t)m * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(detmath_kernel, dim3((k + 255) / 256), dim3(256), 0, 0, d, n, d + k, d + 2 * k, f, m, f + k);
  HIP_OK(c, hipGetLastError());
  if (n) {
    HIP_OK(c, hipMemcpy(sin_out, d + k, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(c, hipMemcpy(cos_out, d + 2 * k, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (m) HIP_OK(c, hipMemcpy(expf_out, f + k, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
  HIP_OK(c, hipFree(d)); HIP_OK(c, hipFree(f));
  return OFDG_OK;
}

}  // extern "C"
