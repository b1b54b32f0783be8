/*
 * ofdg.h -- C-ABI of the MI355X-native optical-flow data generator.
 *
 * This is the drop-in boundary for the reference's per-sample hot path
 * (blueprint -> masks -> warped textures -> composited image0/image1 + flow).
 * Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Every entry point names the reference interface it replaces.  Citations are
 * relative to the reference repository root
 * (lmb-freiburg/optical-flow-2d-data-generation):
 *   DG  = src/caffe/DataGenerator.cpp
 *   DGH = include/caffe/data_generation/DataGenerator.h
 *   LAY = src/caffe/layers/data_generation_layer.cpp
 *   WF  = src/caffe/WarpFields.cpp
 */
#ifndef OFDG_H_
#define OFDG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Error codes (reference: std::runtime_error / glog CHECK, DG:121,    */
/* DG:1143, DG:2004; silently dropped "bad" samples DG:1285-1292).     */
/* ------------------------------------------------------------------ */
#define OFDG_OK          0
#define OFDG_EBADMODE   (-1) /* "BAD MODE"                      DG:2004  */
#define OFDG_ETEXTURES  (-2) /* "Could not open texture ..."    DG:121   */
#define OFDG_EOBJTYPE   (-3) /* "Bad object type ..."           DG:1143  */
#define OFDG_EHIP       (-4) /* HIP runtime failure / extension missing  */
#define OFDG_ECAPACITY  (-5) /* fixed-capacity array exceeded            */
#define OFDG_EINVAL     (-6) /* invalid argument                         */
#define OFDG_ESTARTUP   (-7) /* multi-GPU start-up: another rank failed  */

/* ObjType_t (DGH:369-374) and PolySegmentType_t (DGH:377-381) values. */
#define OFDG_OBJ_DUMMY     0
#define OFDG_OBJ_ELLIPSE   1
#define OFDG_OBJ_POLYGON   2
#define OFDG_OBJ_COMPOSITE 3
#define OFDG_SEG_DUMMY     0
#define OFDG_SEG_LINE      1
#define OFDG_SEG_CURVE3    3

#define OFDG_MAX_SEGMENTS    20 /* RNG_PolyObj_spokes in [3,20]   DG:1687 */
#define OFDG_MAX_COMPONENTS   8 /* RNG_CompObiNumberOfComponents <= 7     */
#define OFDG_BACKGROUND_ID    1 /* BACKGROUND_OBJ_ID              DGH:60  */

/* sampler kinds */
#define OFDG_SAMPLER_REF      0 /* bit-identical 45-stream mt19937 sampler */
#define OFDG_SAMPLER_COUNTER  1 /* device counter-based sampler            */

/*
 * POD mirror of DataGenerator::ObjectBlueprint (DGH:388-421).  Children of a
 * composite live in the same flat array (first_component, n_components)
 * instead of owning heap pointers (DG:934-940).
 */
typedef struct ofdg_blueprint {
  int32_t obj_id;
  int32_t obj_type;
  float   init_rot, init_scale, init_trans_x, init_trans_y;
  float   rot, scale, trans_x, trans_y;
  int32_t tex_id;
  float   tex_rot, tex_scale;
  int32_t tex_shift_x, tex_shift_y;
  float   ellipse_scale_x, ellipse_scale_y;
  int32_t n_segments;
  int32_t segment_type[OFDG_MAX_SEGMENTS];
  float   segment_x[OFDG_MAX_SEGMENTS];
  float   segment_y[OFDG_MAX_SEGMENTS];
  int32_t first_component;
  int32_t n_components;
  int32_t is_additive_component;
  int32_t do_warpfield_deformation;
} ofdg_blueprint;

/*
 * POD mirror of DataGenerator::TaskBucket (DGH:423-437): one sample = one
 * background blueprint + n_objects top-level foreground blueprints, which are
 * contiguous in the flat blueprint array.
 */
typedef struct ofdg_task {
  int32_t background;   /* index of the background blueprint */
  int32_t first_object; /* index of the first top-level foreground blueprint */
  int32_t n_objects;
  int32_t reserved;
} ofdg_task;

/*
 * Options: data_param{batch_size,prefetch} + data_generation_param{mode,
 * first_level_threads, second_level_threads, use_antialiasing}
 * (src/caffe/proto/caffe.proto:6-12, example-prototxt/train.prototxt:9-30)
 * plus extension keys (width/height are #defines in the reference, DGH:55-56).
 */
typedef struct ofdg_params {
  int32_t width;                /* DGEN_WIDTH  (512) */
  int32_t height;               /* DGEN_HEIGHT (384) */
  int32_t mode;                 /* 1..13 */
  int32_t use_antialiasing;     /* default 1 */
  int32_t batch_size;
  int32_t prefetch;
  int32_t first_level_threads;  /* accepted; only used by CPU paths */
  int32_t second_level_threads; /* accepted; only used by CPU paths */
  int32_t num_objects;          /* 0 = reference behaviour (16..23, DG:1666) */
  int32_t sampler;              /* OFDG_SAMPLER_* */
  int32_t seed;                 /* counter sampler seed; ref sampler uses 0..44 */
  int32_t rank, world_size;     /* sample sharding (g = step*B*world + rank*B + i) */
  int32_t device;               /* HIP device ordinal */
  int32_t max_shapes_per_sample;/* 0 = default capacity */
  int32_t background_prep;      /* background texture m_textures[0] (DG:1186-1192):
                                   0 = centre 2W x 2H crop of the pool image (the C-ABI default; parity boundary of round 1);
                                   1 = Texture::getRandomizedCrop(2W, 2H, tex_rot, tex_scale, tex_shift) (DG:87-109) the way
                                       CImg runs it: get_shift -> rotate (linear, mirror, grown canvas) -> u8 -> crop (float ->
                                       int truncation, mirror) -> resize (per axis linear / moving average, u8 in between);
                                   2 = the same geometry as ONE resampling along the composed coordinate map (fast form:
                                       one interpolation less of blur, values differ from 1).  CImg itself: parity unpinned */
  /* how the context schedules its work (no effect on the bytes it renders; tests/test_gpu_sampler.py):
   *   chains     internal in-order streams, each [sampler ->] geom -> raster -> compose of one batch (1..8).  0 = automatic:
   *              4 when the process was started with GPU_MAX_HW_QUEUES >= 8 (one hardware queue per chain), else 3;
   *              ofdg_ctx_info() says which and why
   *   lookahead  counter sampler: batches whose preparation kernels are enqueued ahead of the call that composes them
   *              (the reference's prefetch thread, LAY:141-172); 0 = none
   *   serial     1 = everything on the caller's stream, one kernel after the other (debugging, per-kernel timing) */
  int32_t chains, lookahead, serial;
  int32_t reserved[5];
} ofdg_params;

typedef struct ofdg_ctx ofdg_ctx;

/* Fill *p with the reference's defaults (caffe.proto:6-12, DGH:55-56). */
void ofdg_default_params(ofdg_params* p);

/*
 * Replaces DataGenerationLayer ctor + LayerSetUp (LAY:36-56, 106-132):
 * DataGenerator(param) + ObjectParametersGenerator(param) (DG:990-998,
 * DG:1358-2054).  Fails with OFDG_EHIP if no HIP device / kernel image.
 */
int ofdg_create(const ofdg_params* params, ofdg_ctx** out);
void ofdg_destroy(ofdg_ctx* ctx);
/* The parameters the ctx was created with. */
const ofdg_params* ofdg_ctx_params(const ofdg_ctx* ctx);
/* Message of the last failure on this ctx (or of a failed ofdg_create if ctx==NULL). */
const char* ofdg_last_error(const ofdg_ctx* ctx);
/* One line on how the context is set up: number of chains and what decided it (GPU_MAX_HW_QUEUES as the library found it
 * when it was loaded; HIP reads the variable when the runtime starts, so it must be in the environment of the process),
 * look-ahead, serial mode.  Valid until the ctx is destroyed. */
const char* ofdg_ctx_info(const ofdg_ctx* ctx);

/* ---- texture pool: replaces TextureCollection (DG:117-161) ---------------- */
/* n seeded synthetic w x h textures generated directly in HBM. */
int ofdg_pool_synthetic(ofdg_ctx* ctx, int n, int w, int h, uint32_t seed);
/* Reserve an empty pool of n textures of w x h, to be filled by ofdg_pool_upload. */
int ofdg_pool_alloc(ofdg_ctx* ctx, int n, int w, int h);
/* Upload one texture: planar B,G,R u8 (CImg layout after the swap at DG:129-131). */
int ofdg_pool_upload(ofdg_ctx* ctx, int index, const uint8_t* bgr_planar, int w, int h);
/* A pool of n images of DIFFERENT sizes (real texture lists, TextureCollection DG:117-149): every image is
 * reduced at upload to what the path reads - its W x H foreground texture and its 2W x 2H background texture
 * (centre crop, or the CImg-resized whole image if it is smaller, DG:96-106); with background_prep the whole
 * image stays resident as well (the preparation works on the original image). */
int ofdg_pool_alloc_mixed(ofdg_ctx* ctx, int n);
int ofdg_pool_upload_mixed(ofdg_ctx* ctx, int index, const uint8_t* bgr_planar, int w, int h);
/* Download texture `index` as planar B,G,R u8 (w*h*3 bytes). */
int ofdg_pool_download(ofdg_ctx* ctx, int index, uint8_t* bgr_planar);
int ofdg_pool_info(const ofdg_ctx* ctx, int* n, int* w, int* h);
/* The resident pool (one image size) as raw device memory, n*h*w BGRX texels: lets rank 0 load the texture
 * collection (TextureCollection, DG:117-149) and the other ranks receive their replica with one RCCL broadcast
 * over xGMI instead of reading the files again.  mark_written != 0: the caller is about to overwrite it. */
int ofdg_pool_device(ofdg_ctx* ctx, void** ptr, unsigned long long* bytes, int mark_written);

/*
 * Replaces the sampling half of load_batch (LAY:197-213):
 * generateBackground / generateNumberOfFgObjects / generateForegroundObject
 * (DG:2105-2835).  Appends n_tasks tasks; blueprints go to bps[0..*n_bps).
 * OFDG_SAMPLER_REF continues the 45 seeded streams of this ctx.
 */
int ofdg_sample(ofdg_ctx* ctx, int n_tasks, ofdg_task* tasks,
                ofdg_blueprint* bps, int bps_capacity, int* n_bps);

/*
 * THE HOT PATH.  Replaces commissionNewTask + Process_TaskBucket +
 * retrieveFinishedTask (DG:1175-1254, 1308-1349) and the batch assembly of
 * load_batch (LAY:227-250).  Renders task i into slot i of the caller-owned
 * DEVICE buffers image0 [n,3,H,W], image1 [n,3,H,W], flow [n,2,H,W] (float32,
 * planar, B,G,R order, 0..255).  Asynchronous on `stream` (a hipStream_t): the
 * outputs are written by a kernel on `stream`, after the work enqueued there before
 * the call, and work enqueued there after it sees them.  Internally the context owns
 * a few in-order streams ("chains", taking turns call by call) on which the record
 * upload / device sampler and the outline and coverage kernels of one call run back
 * to back, overlapping the neighbouring calls; the compose kernel follows on
 * `stream` after one event.  Pass ofdg_stream(ctx) as `stream` to run compose on the
 * chain's own stream too (no cross-stream wait at all; the compose kernels of
 * consecutive calls then overlap as well - the fast way to drive a prefetch ring:
 * one output buffer set per call in flight).
 */
/* As a call's `stream`: "the internal stream this call works on" - the same as passing ofdg_stream(ctx) read right before
 * the call, without the extra call. */
#define OFDG_STREAM_OWN ((void*)(~(uintptr_t)0))
int ofdg_render(ofdg_ctx* ctx, const ofdg_task* tasks, int n_tasks,
                const ofdg_blueprint* bps, int n_bps,
                float* d_image0, float* d_image1, float* d_flow, void* stream);

/* Re-run the device half of the last render / forward call (records already
 * resident in HBM): used to time the path with inputs resident. */
int ofdg_render_resident(ofdg_ctx* ctx, float* d_image0, float* d_image1,
                         float* d_flow, void* stream);

/* Prefetch ring (maps data_param.prefetch, LAY:36-56, 141-172): up to 16 batches can be
 * resident in HBM at once.  ofdg_upload_slot realises + uploads one batch into `slot`;
 * ofdg_render_slot renders a resident slot (any number of times).  ofdg_render and
 * ofdg_forward* keep their records in private slots of their own. */
int ofdg_upload_slot(ofdg_ctx* ctx, int slot, const ofdg_task* tasks, int n_tasks,
                     const ofdg_blueprint* bps, int n_bps, void* stream);
int ofdg_render_slot(ofdg_ctx* ctx, int slot, float* d_image0, float* d_image1,
                     float* d_flow, void* stream);

/* The sharding rule of ofdg_forward: step `step` of rank `rank` renders the global sample indices
 * ofdg_shard_first_index(step, batch, world_size, rank) + [0, batch) = step * batch * world_size + rank * batch + [0, batch);
 * over all ranks and steps these ranges tile the stream without gaps or overlap.  -1 for invalid arguments.
 * (The reference has no counterpart: every solver's layer renders the same samples, data_generation_layer.hpp:54.) */
long long ofdg_shard_first_index(long long step, int batch, int world_size, int rank);

/* Replaces Forward_cpu/Forward_gpu (LAY:266-291): sample batch_size tasks and
 * render them. */
int ofdg_forward(ofdg_ctx* ctx, float* d_image0, float* d_image1, float* d_flow,
                 void* stream);

/* Device counter-based sampler (OFDG_SAMPLER_COUNTER): the sampling half of load_batch
 * (LAY:197-213) on the GPU.  Sample g is a pure function of (seed, g); same modes and
 * distributions as the reference stream, statistically (not bitwise) equal to it.
 * ofdg_forward_counter samples + renders global indices first_index .. first_index+n-1
 * with no host data in the loop (ofdg_forward uses it when params.sampler is COUNTER:
 * rank r takes indices step*B*world + r*B + [0,B)).  ofdg_sample_counter downloads the
 * blueprints instead (fixed layout: 257 per sample = background, 32 object slots,
 * 32 x 7 component slots; unused slots have obj_type 0). */
int ofdg_forward_counter(ofdg_ctx* ctx, long long first_index, int n_samples,
                         float* d_image0, float* d_image1, float* d_flow, void* stream);
int ofdg_sample_counter(ofdg_ctx* ctx, long long first_index, int n_samples,
                        ofdg_task* tasks, ofdg_blueprint* bps);

/* Checkpoint / resume of ofdg_forward: the number of batches this context has produced is its whole sampler
 * state (the reference cannot resume: a restarted job replays its 45 streams from their seeds).
 * ofdg_set_step(k) makes the next ofdg_forward produce batch k (counter sampler: at no cost; reference-stream
 * sampler: the streams are rebuilt and k * batch_size * world_size tasks drawn and dropped on the host; mode 9:
 * the crop serving order of this rank is replayed too - install the warp fields BEFORE calling it). */
long long ofdg_get_step(const ofdg_ctx* ctx);
int ofdg_set_step(ofdg_ctx* ctx, long long step);

/* The internal stream the NEXT render / forward call of this context will work on
 * (a hipStream_t; they take turns).  See ofdg_render. */
void* ofdg_stream(ofdg_ctx* ctx);
/* How many internal streams take turns: a caller that cycles k * ofdg_num_chains output buffer sets and passes
 * ofdg_stream() finds every set always written by the same stream (no event needed between its writers). */
int ofdg_num_chains(const ofdg_ctx* ctx);

/* Wait for `stream` and for everything the context has in flight, and report
 * device-side error flags raised by kernels. */
int ofdg_synchronize(ofdg_ctx* ctx, void* stream);

/* The same device-side error flags WITHOUT waiting for anything in flight (all batches rendered so far). */
int ofdg_poll_errors(ofdg_ctx* ctx);
/* ... and per batch.  Every render / forward call has a number in its context ("ticket", ofdg_last_ticket right after the
 * call) and raises its device flags in a word of its own: ofdg_poll_errors_of(ticket) says whether THAT batch was truncated
 * (OFDG_ECAPACITY, the message names the batch) - what a prefetch ring calls when it hands over a batch whose own completion
 * event it has waited for (prefetch_full_.pop, LAY:269), so that an error of batch k is reported at batch k's Forward, not
 * at an older batch's, and batches k - 1 and k + 1 stay valid.  (The reference drops a bad sample silently, DG:1285-1292.)
 * The flags of the last 256 calls are kept; reading a word clears it - the error is reported ONCE, so a caller that wants to
 * go on must drop that batch when it gets the error (ofdg::DataGenerationLayer retires the buffer set before it throws).
 * A word never speaks for another batch: once a caller has asked by ticket, a word whose last owner was not asked about
 * is cleared in front of the first kernel of the call that takes it over (256 calls later), and a batch that was prepared
 * ahead and discarded takes its flags with it. */
long long ofdg_last_ticket(const ofdg_ctx* ctx);
int ofdg_poll_errors_of(ofdg_ctx* ctx, long long ticket);

/* ---- mode 9 (non-rigid deformation) warp fields: replaces WarpFields::CropGenerator
 * (WF:469-641), which DataGenerator::Start launches for MODE == 9 (DG:1016-1020). ---- */
/* Generate n_fields big fields (side 3*max(W,H)) on the device from seeded displacer
 * lists (the reference seeds from std::random_device) and cut them into (W+1)x(H+1)
 * crops; crops are then served like CropGenerator::get_crop (each 3 times, in order).
 * The Gaussian supports' exponential (WF:101-112) is the det_expf of include/ofdg_detmath.h (the fp64 exponential rounded
 * once) for BOTH samplers, not libm's expf: the fields are reproducible bit for bit on any device and in the oracle's detmath
 * mode, and agree with a libm evaluation to a small fraction of a pixel (tests/test_gpu_parity.py). */
int ofdg_warp_generate(ofdg_ctx* ctx, int n_fields, uint32_t seed);
/* Install caller-provided crops instead: n x {flow x, flow y, iflow x, iflow y} planes of
 * (H+1)*(W+1) floats (host memory). */
int ofdg_warp_upload(ofdg_ctx* ctx, const float* crops, int n);
int ofdg_warp_info(const ofdg_ctx* ctx, int* n_crops, int* w, int* h);
int ofdg_warp_download(ofdg_ctx* ctx, int index, float* crop);
/* Host only: displacer draws of one big field, n x 9 doubles {type, p0, p1, p2, support
 * cx, cy, sigma_x, sigma_y, angle} (WF:572-610). Returns n, or -n if cap is too small. */
int ofdg_host_displacers(int width, int height, uint32_t seed, double* out, int cap);

/* ---- inspection (tests / profiling) ---------------------------------------- */
/* Rasterise one polygon (n double vertices, already in screen space) with the
 * device rasteriser and return the raw AGG coverage (0..255) as w*h bytes. */
int ofdg_debug_rasterize(ofdg_ctx* ctx, const double* xy, int n_vertices,
                         uint8_t* coverage_host);
/* The same for a path with curve3 segments, flattened by the DEVICE (types[i]: 1 line_to, 3 curve3 control point
 * followed by its end point; entry 0 is the move_to vertex; n <= 64): conv_curve / curve3_div as geom_kernel runs them
 * (DG:493-517), pinned against compiled AGG in tests/test_gpu_parity.py. */
int ofdg_debug_rasterize_path(ofdg_ctx* ctx, const double* xy, const int* types, int n, uint8_t* coverage_host);
/* The DEVICE's span_interpolator_linear + dda2_line_interpolator (DG:203-216): out_xy[rows][len][2] = (x, y) in 24.8
 * fixed point under the inverse affine inv[6] (AGG member order), before the filter's -128 offset. */
int ofdg_debug_dda_rows(ofdg_ctx* ctx, const double* inv, int rows, int len, int* out_xy);
/* After ofdg_render: raw coverage of rasterised shape `shape` of sample
 * `sample`, frame 0/1, as w*h host bytes (zero outside its bounding box). */
int ofdg_debug_coverage(ofdg_ctx* ctx, int sample, int shape, int frame,
                        uint8_t* coverage_host);
int ofdg_debug_num_shapes(ofdg_ctx* ctx, int sample);
/* Number of raster work items the last launch left unprocessed (diagnostics: 0). */
int ofdg_debug_item_count(ofdg_ctx* ctx);
/* After a render / forward call with background_prep = 1: the number of tiles the one-launch form of the preparation
 * (bgprep_stream_kernel) walked for that batch and the number of workgroups that shared them grid-stride (0 tiles: the batch took
 * another form of the preparation).  Tests use it to make sure they reach a workgroup's 2nd, 3rd ... tile. */
int ofdg_debug_bgprep_tiles(ofdg_ctx* ctx, int* tiles, int* workgroups);
/* Tiles of bgprep_stream_kernel since the last call, by the form that rendered them: counts9[rotation + 3 * resize], rotation
 * 0 = general (mirrored / clamped crop coordinates), 1 = inside the image, 2 = inside and specialised by the side of the shift's
 * mirror lines; resize 0 = decided per row, 1 = both axes enlarge, 2 = both shrink.  The first call switches the counting on and
 * returns zeros.  (Texture::getRandomizedCrop, DG:87-109: which of its branches the batch took.) */
int ofdg_debug_bgprep_paths(ofdg_ctx* ctx, unsigned* counts9);
/* Exhaustive device evaluation of the per-byte formulas: composite add / subtract
 * [u*256+v] (DG:606, 626), AA mask byte [c], draw_image blend [d*256+m] for s=s_fixed. */
int ofdg_debug_tables(ofdg_ctx* ctx, uint8_t* add_tbl, uint8_t* sub_tbl, uint8_t* aa_tbl,
                      uint8_t* blend_tbl, int s_fixed);
/* include/ofdg_detmath.h (the sin / cos / expf the device counter-sampler path is defined with) evaluated
 * on the device: n angles -> sin, cos; m floats -> expf.  Host arrays. */
int ofdg_debug_detmath(ofdg_ctx* ctx, const double* angles, int n, double* sin_out, double* cos_out,
                       const float* x, int m, float* expf_out);
/* Per-kernel device time (ms), averaged over the launches recorded since ofdg_set_profiling (mode 1: the compose launch
 * and the background preparation of every 4th batch, completion signals on the kernels' own packets, nothing added to the
 * streams; mode 2: every kernel of every batch, with start markers; 0: off).  Names: "geom", "raster" (mode 2), "compose",
 * "background_prep" (where the preparation runs behind raster: batches the library prepares itself).
 * Mode 1 takes the compose launch's time from the completion of the chain's last preparation kernel to its own completion
 * (no marker packet in the stream): only launches enqueued right behind their preparation on the chain's own stream are
 * samples - a batch prepared ahead (ofdg_params.lookahead) or composed on a caller's stream is left out, since that span
 * would hold host and queue idle time; with nothing but such launches ofdg_kernel_ms reports "no profiled launch".  Mode 2
 * (start markers) times every launch. */
int ofdg_set_profiling(ofdg_ctx* ctx, int mode);
int ofdg_kernel_ms(ofdg_ctx* ctx, const char* kernel, float* ms);

/* ---- host-side pieces (no HIP device needed) --------------------------------- */
/* The reference-stream sampler on its own: ObjectParametersGenerator (DG:1358-2835)
 * driven like load_batch (LAY:197-213). */
typedef struct ofdg_host_sampler ofdg_host_sampler;
int ofdg_host_sampler_create(int mode, int width, int height, int num_objects,
                             ofdg_host_sampler** out);
int ofdg_host_sampler_next(ofdg_host_sampler* s, int n_tasks, ofdg_task* tasks,
                           ofdg_blueprint* bps, int bps_capacity, int* n_bps);
void ofdg_host_sampler_destroy(ofdg_host_sampler* s);
/* Blueprint -> fp64 affines exactly as RealizeObjectBlueprint / setMotion /
 * addBackgroundMotion compute them (DG:302-335, 1065-1173).  shape_mats: 12 doubles
 * per rasterised shape {intrinsic[6], intrinsic*motion[6]}; object_mats: 12 doubles
 * per blitted object {motion[6], inverse texture warp[6]} (sx,shy,shx,sy,tx,ty). */
int ofdg_host_realize(const ofdg_params* prm, int pool_n, int pool_w, int pool_h,
                      const ofdg_task* tasks, int n_tasks, const ofdg_blueprint* bps,
                      int n_bps, double* shape_mats, int shape_cap, int* n_shapes,
                      double* object_mats, int object_cap, int* n_objects);
/* Host logic of ofdg_params.background_prep = 1 (no GPU): the coordinate-map record of
 * Texture::getRandomizedCrop(2W, 2H, angle, zoom, shift) (DG:87-109) on a pool_w x pool_h image.
 * f[8] = ca, sa, w2, h2, rw2, rh2, fx, fy; i[6] = x0, y0, cw, ch, shift_x, shift_y. */
int ofdg_host_bg_prep(int pool_w, int pool_h, int width, int height, float angle, float zoom,
                      int shift_x, int shift_y, float* f, int* i);

/* One image file of a texture list as the layer's loader decodes it (TextureCollection, DG:117-149: CImg::load, then
 * R <-> B): binary PPM natively, PNG through the system's libpng 1.6 (bound at run time).  planar_bgr: 3 planes of
 * width x height bytes (B, G, R), or NULL to ask for the size only. */
int ofdg_host_decode_image(const char* path, uint8_t* planar_bgr, size_t capacity, int* width, int* height);

/* Parse a `layer { ... }` prototxt block (example-prototxt/train.prototxt). */
int ofdg_parse_prototxt(const char* text, ofdg_params* out, char* texture_dbases,
                        int texture_dbases_cap, int* n_top);
const char* ofdg_host_last_error(void);

/* ---- Caffe-layer-shaped surface: DataGenerationLayer (LAY:36-132, 266-291) ------ */
typedef struct ofdg_layer ofdg_layer;
/* ctor + LayerSetUp: parses the prototxt, opens the texture collection
 * ("synthetic:N:W:H[:seed]" or a list file of binary PPMs), reshapes 3 tops. */
int ofdg_layer_create(const char* layer_prototxt, ofdg_layer** out);
/* Forward_gpu: returns the device pointers of image0/image1/flow and the
 * shape {N,3,H,W} of image0. */
int ofdg_layer_forward(ofdg_layer* layer, float** image0, float** image1, float** flow,
                       int* shape4);
void ofdg_layer_destroy(ofdg_layer* layer);
/* data_param.prefetch > 1: how many of the batches rendered ahead were still unfinished when the last
 * ofdg_layer_forward returned (it waits for the oldest batch only, like prefetch_full_.pop, LAY:269). */
int ofdg_layer_in_flight(const ofdg_layer* layer);

/* ---- multi-GPU start-up (one process per GPU) -----------------------------------------------
 * Samples shard by global index with no data-path collective; what the ranks must agree on is the stream and
 * the pool.  ONE RCCL broadcast (ncclBroadcast over xGMI) carries rank 0's setup header + texture index table.
 * The reference has nothing to replace here: Caffe's multi-GPU solvers each build their own layer from the same
 * prototxt and the same 45 seeds (data_generation_layer.hpp:54, DG:1360) and so render identical samples. */
#define OFDG_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */
#define OFDG_POOL_SYNTHETIC 0    /* every rank generates the pool itself from pool_seed */
#define OFDG_POOL_UNIFORM   1    /* images of one size (pool_w x pool_h), contents replicated with ofdg_comm_bcast_pool */
#define OFDG_POOL_MIXED     2    /* images of different sizes (ofdg_pool_alloc_mixed), contents replicated likewise */
typedef struct ofdg_setup {
  int32_t seed, mode, width, height, num_objects, use_antialiasing, batch_size, sampler, background_prep;
  int32_t n_tex;          /* images in the texture pool */
  int32_t pool_kind;      /* OFDG_POOL_* */
  int32_t pool_w, pool_h; /* image size (0 x 0 for a mixed pool) */
  uint32_t pool_seed;     /* synthetic pools */
  int32_t n_table;        /* valid entries of the index table that travels with the header */
  int32_t status;         /* 0, or the root's error code (< 0): the root could not set itself up; every receiver fails with it */
  int32_t max_shapes_per_sample;
  int32_t reserved;
} ofdg_setup;
/* One texture of the pool as the kernels address it: `offset` texels into the pool the foreground path reads,
 * rows `pitch` texels apart; w x h = size of the source image. */
typedef struct ofdg_tex_entry {
  uint64_t offset;
  uint32_t w, h, pitch, reserved;
} ofdg_tex_entry;
typedef struct ofdg_comm ofdg_comm;
/* rank 0: ncclGetUniqueId into id[OFDG_UNIQUE_ID_BYTES]; hand it to the other processes through the launcher's
 * rendezvous (file, environment, key-value store). */
int ofdg_comm_unique_id(void* id);
/* hipSetDevice(device), then ncclCommInitRank.  Call it before any other HIP work of the process. */
int ofdg_comm_init(const void* id, int rank, int world_size, int device, ofdg_comm** out);
/* Use a communicator the caller already owns (an ncclComm_t); it is not destroyed by ofdg_comm_destroy. */
int ofdg_comm_adopt(void* nccl_comm, int rank, int world_size, int device, ofdg_comm** out);
void ofdg_comm_destroy(ofdg_comm* comm);
int ofdg_comm_rank(const ofdg_comm* comm);
int ofdg_comm_world_size(const ofdg_comm* comm);
const char* ofdg_comm_last_error(const ofdg_comm* comm);
/* THE start-up collective: root's *setup and table[0 .. setup->n_table) reach every rank in one ncclBroadcast.
 * Success or failure is decided together: a root that cannot provide a setup (its context or texture collection failed,
 * its table exceeds table_cap) still broadcasts - a header whose `status` carries its error code - so that every
 * receiver returns that code instead of waiting for a broadcast that never comes (ofdg_comm_bcast_abort is that call
 * for a root that failed before it had a setup at all). */
int ofdg_comm_bcast_setup(ofdg_comm* comm, int root, ofdg_setup* setup, ofdg_tex_entry* table, int table_cap);
int ofdg_comm_bcast_abort(ofdg_comm* comm, int root, int error_code, int table_cap);
/* Ranks of the communicator as RCCL reports them (ncclCommCount): what a multi-GPU run prints as its proof that the
 * native start-up really spanned world_size processes. */
int ofdg_comm_nccl_count(ofdg_comm* comm);
/* Replicate the root's resident pool into the (identically allocated) pools of the other ranks: ncclBroadcast
 * between the HBM pools, so that only the root reads the texture collection from disk (DG:117-149).  Every rank first
 * checks that all of its buffers exist and the ranks agree on that (one ncclAllReduce of a flag) before the first
 * payload broadcast: a rank that cannot take part makes the call fail on every rank. */
int ofdg_comm_bcast_pool(ofdg_comm* comm, int root, ofdg_ctx* ctx);
/* "Did every rank get this far?"  One ncclAllReduce(min) of a flag; every rank of the communicator must call it, with
 * local_ok = 0 if its own step failed (context creation, pool allocation ...).  Returns OFDG_OK on every rank if all
 * passed 1, otherwise OFDG_ESTARTUP on every rank - so a rank that failed between two collectives takes the others
 * out with it instead of leaving them waiting in the next one.  (A failure of the collective itself returns OFDG_EHIP
 * on the rank that saw it; RCCL then aborts the others.) */
int ofdg_comm_agree(ofdg_comm* comm, int local_ok);
/* The header / index table of a context's stream and pool (what the root broadcasts), and the parameters a
 * receiving rank creates its context with (rank, world_size and device come from the communicator). */
int ofdg_setup_of(const ofdg_ctx* ctx, ofdg_setup* setup, ofdg_tex_entry* table, int table_cap);
int ofdg_setup_params(const ofdg_setup* setup, const ofdg_comm* comm, ofdg_params* params);
/* Allocate (synthetic: also fill) this rank's pool as the setup describes it; a mixed pool takes its image sizes
 * from the index table that came with the setup. */
int ofdg_setup_alloc_pool(ofdg_ctx* ctx, const ofdg_setup* setup, const ofdg_tex_entry* table);
/* The derived textures of a mixed pool as raw device memory: [n][H][W] foreground and [n][2H][2W] background. */
int ofdg_pool_device_mixed(ofdg_ctx* ctx, void** fg, unsigned long long* fg_bytes, void** bg,
                           unsigned long long* bg_bytes);
/* ... and, with background_prep, whole image `index` (w * h BGRX texels): getRandomizedCrop reads the original. */
int ofdg_pool_device_image(ofdg_ctx* ctx, int index, void** ptr, unsigned long long* bytes);
/* DataGenerationLayer on a communicator: every rank passes the same prototxt; rank 0's options and texture
 * collection win (one broadcast), rank / world_size / device come from the communicator, and Forward yields
 * this rank's shard of every global batch. */
int ofdg_layer_create_dist(const char* layer_prototxt, ofdg_comm* comm, ofdg_layer** out);

#ifdef __cplusplus
}
#endif
#endif /* OFDG_H_ */
