// Multi-GPU start-up of the data generator: ONE RCCL broadcast over xGMI of the shared stream / pool
// description (seed, mode, frame size, object count, texture index table), after which every rank
// renders its own disjoint shard of the sample stream with no further communication
// (g = step*B*world + rank*B + i).  The reference has no counterpart: Caffe's multi-GPU
// solvers each construct their own DataGenerationLayer from the same prototxt with the same 45 seeds
// (data_generation_layer.hpp:54 ShareInParallel() == false, DataGenerator.cpp:1360) and so render the
// SAME samples on every GPU.
//
// RCCL is bound at run time (dlopen of librccl.so.1: inside a PyTorch process that is the copy torch
// already holds, so the process keeps ONE collective runtime); a single-GPU user needs no RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ofdg.h"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl* rccl(std::string* err) {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.so) break;
    }
    if (!r.so) {
      r.err = std::string("RCCL is not loadable: ") + dlerror();
    } else {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
      r.Broadcast = (decltype(r.Broadcast))dlsym(r.so, "ncclBroadcast");
      r.AllReduce = (decltype(r.AllReduce))dlsym(r.so, "ncclAllReduce");
      r.CommCount = (decltype(r.CommCount))dlsym(r.so, "ncclCommCount");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Broadcast || !r.AllReduce || !r.CommCount || !r.GetErrorString)
        r.err = "RCCL lacks a required symbol";
    }
  }
  if (!r.err.empty()) {
    if (err) *err = r.err;
    return nullptr;
  }
  return &r;
}

thread_local std::string g_comm_error;

}  // namespace

struct ofdg_comm {
  ncclComm_t comm = nullptr;
  bool owned = false;  // created here (ofdg_comm_init) vs adopted from the caller
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  void* d_buf = nullptr;
  size_t d_bytes = 0;
  int* d_flag = nullptr;  // comm_agree
  std::string err;
};

#define COMM_HIP(c, call)                                                       \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
      return OFDG_EHIP;                                                         \
    }                                                                           \
  } while (0)
#define COMM_NCCL(c, R, call)                                                   \
  do {                                                                          \
    ncclResult_t r_ = (call);                                                   \
    if (r_ != ncclSuccess) {                                                    \
      (c)->err = std::string(#call) + ": " + (R)->GetErrorString(r_);           \
      return OFDG_EHIP;                                                         \
    }                                                                           \
  } while (0)

static_assert(sizeof(ncclUniqueId) == OFDG_UNIQUE_ID_BYTES, "ofdg.h promises the size of ncclUniqueId");

extern "C" {

const char* ofdg_comm_last_error(const ofdg_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

int ofdg_comm_unique_id(void* id) {
  if (!id) return OFDG_EINVAL;
  Rccl* R = rccl(&g_comm_error);
  if (!R) return OFDG_EHIP;
  ncclUniqueId u;
  ncclResult_t r = R->GetUniqueId(&u);
  if (r != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + R->GetErrorString(r); return OFDG_EHIP; }
  std::memcpy(id, &u, sizeof(u));
  return OFDG_OK;
}

static int comm_finish(std::unique_ptr<ofdg_comm>& c, ofdg_comm** out) {
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { g_comm_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); return OFDG_EHIP; }
  // the word the ranks agree through (ofdg_comm_agree): allocated here, so that agreeing cannot fail on an allocation
  if ((e = hipMalloc((void**)&c->d_flag, 256)) != hipSuccess || (e = hipMemset(c->d_flag, 0, 256)) != hipSuccess) {
    g_comm_error = std::string("hipMalloc (agreement flag): ") + hipGetErrorString(e);
    (void)hipStreamDestroy(c->stream);
    return OFDG_EHIP;
  }
  *out = c.release();
  return OFDG_OK;
}

int ofdg_comm_init(const void* id, int rank, int world_size, int device, ofdg_comm** out) {
  if (!id || !out || world_size < 1 || rank < 0 || rank >= world_size) { g_comm_error = "ofdg_comm_init: invalid argument"; return OFDG_EINVAL; }
  *out = nullptr;
  Rccl* R = rccl(&g_comm_error);
  if (!R) return OFDG_EHIP;
  // one process per GPU: the device is bound before the communicator touches HIP
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { g_comm_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return OFDG_EHIP; }
  std::unique_ptr<ofdg_comm> c(new ofdg_comm());
  c->rank = rank; c->world = world_size; c->device = device; c->owned = true;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclResult_t r = R->CommInitRank(&c->comm, world_size, u, rank);
  if (r != ncclSuccess) { g_comm_error = std::string("ncclCommInitRank: ") + R->GetErrorString(r); return OFDG_EHIP; }
  return comm_finish(c, out);
}

int ofdg_comm_adopt(void* nccl_comm, int rank, int world_size, int device, ofdg_comm** out) {
  if (!nccl_comm || !out || world_size < 1 || rank < 0 || rank >= world_size) { g_comm_error = "ofdg_comm_adopt: invalid argument"; return OFDG_EINVAL; }
  *out = nullptr;
  if (!rccl(&g_comm_error)) return OFDG_EHIP;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { g_comm_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return OFDG_EHIP; }
  std::unique_ptr<ofdg_comm> c(new ofdg_comm());
  c->comm = (ncclComm_t)nccl_comm; c->rank = rank; c->world = world_size; c->device = device; c->owned = false;
  return comm_finish(c, out);
}

void ofdg_comm_destroy(ofdg_comm* c) {
  if (!c) return;
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (c->d_buf) (void)hipFree(c->d_buf);
  if (c->d_flag) (void)hipFree(c->d_flag);
  if (c->owned && c->comm) { Rccl* R = rccl(nullptr); if (R) (void)R->CommDestroy(c->comm); }
  delete c;
}

int ofdg_comm_rank(const ofdg_comm* c) { return c ? c->rank : OFDG_EINVAL; }
int ofdg_comm_world_size(const ofdg_comm* c) { return c ? c->world : OFDG_EINVAL; }

// THE start-up collective: root's {setup header, texture index table} -> every rank, as ONE ncclBroadcast of one
// packed buffer (header, then n_tex entries; receivers size the buffer by table_cap, the header says how many
// entries are valid).
int ofdg_comm_bcast_setup(ofdg_comm* c, int root, ofdg_setup* setup, ofdg_tex_entry* table, int table_cap) {
  if (!c || !setup || root < 0 || root >= c->world || table_cap < 0 || (table_cap > 0 && !table)) {
    if (c) c->err = "ofdg_comm_bcast_setup: invalid argument";
    return OFDG_EINVAL;
  }
  Rccl* R = rccl(&c->err);
  if (!R) return OFDG_EHIP;
  // a root that cannot provide its setup says so IN the broadcast (status): the receivers are already waiting for it
  int root_failure = OFDG_OK;
  if (c->rank == root) {
    if (setup->status < 0) root_failure = setup->status;
    else if (setup->n_tex < 0 || setup->n_table < 0 || setup->n_table > table_cap) {
      root_failure = OFDG_ECAPACITY;
      c->err = "ofdg_comm_bcast_setup: root's table exceeds table_cap";
    }
  }
  COMM_HIP(c, hipSetDevice(c->device));
  const size_t bytes = sizeof(ofdg_setup) + (size_t)table_cap * sizeof(ofdg_tex_entry);
  if (bytes > c->d_bytes) {
    if (c->d_buf) COMM_HIP(c, hipFree(c->d_buf));
    c->d_buf = nullptr;
    COMM_HIP(c, hipMalloc(&c->d_buf, bytes));
    c->d_bytes = bytes;
  }
  std::vector<char> host(bytes, 0);
  if (c->rank == root) {
    ofdg_setup hdr = *setup;
    if (root_failure != OFDG_OK) { hdr.status = root_failure; hdr.n_table = 0; }
    std::memcpy(host.data(), &hdr, sizeof(ofdg_setup));
    if (hdr.n_table > 0) std::memcpy(host.data() + sizeof(ofdg_setup), table, (size_t)hdr.n_table * sizeof(ofdg_tex_entry));
    COMM_HIP(c, hipMemcpyAsync(c->d_buf, host.data(), bytes, hipMemcpyHostToDevice, c->stream));
  }
  COMM_NCCL(c, R, R->Broadcast(c->d_buf, c->d_buf, bytes, ncclUint8, root, c->comm, c->stream));
  COMM_HIP(c, hipMemcpyAsync(host.data(), c->d_buf, bytes, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(c, hipStreamSynchronize(c->stream));
  if (c->rank == root && root_failure != OFDG_OK) {
    if (c->err.empty()) c->err = "ofdg_comm_bcast_setup: the root reported failure " + std::to_string(root_failure);
    return root_failure;
  }
  std::memcpy(setup, host.data(), sizeof(ofdg_setup));
  if (setup->status < 0) {
    c->err = "start-up failed on the root rank (error code " + std::to_string(setup->status) + "); see its message";
    return setup->status;
  }
  if (setup->n_table < 0 || setup->n_table > table_cap) { c->err = "ofdg_comm_bcast_setup: the root's table does not fit this rank's table_cap"; return OFDG_ECAPACITY; }
  if (setup->n_table > 0) std::memcpy(table, host.data() + sizeof(ofdg_setup), (size_t)setup->n_table * sizeof(ofdg_tex_entry));
  return OFDG_OK;
}

int ofdg_comm_bcast_abort(ofdg_comm* c, int root, int error_code, int table_cap) {
  if (!c) return OFDG_EINVAL;
  ofdg_setup su;
  std::memset(&su, 0, sizeof(su));
  su.status = error_code < 0 ? error_code : OFDG_EINVAL;
  std::vector<ofdg_tex_entry> table((size_t)(table_cap > 0 ? table_cap : 1));
  (void)ofdg_comm_bcast_setup(c, root, &su, table.data(), table_cap);
  return su.status;
}

int ofdg_comm_nccl_count(ofdg_comm* c) {
  if (!c) return OFDG_EINVAL;
  Rccl* R = rccl(&c->err);
  if (!R) return OFDG_EHIP;
  int n = 0;
  COMM_NCCL(c, R, R->CommCount(c->comm, &n));
  return n;
}

// do all ranks say `ok`?  (one ncclAllReduce(min) of a flag)
// The flag's device word is allocated when the communicator is made (comm_finish), so that nothing between "this rank
// decides its flag" and "this rank enters the all-reduce" can fail locally and leave the other ranks in the collective.
static int comm_agree(ofdg_comm* c, Rccl* R, bool ok, bool* all_ok) {
  int flag = ok ? 1 : 0;
  *all_ok = false;
  // (a failed copy of the flag is a local failure like any other: say "not ok" with whatever the word holds - it is
  //  zeroed at allocation and after every agreement - and still enter the collective)
  hipError_t e_in = hipMemcpyAsync(c->d_flag, &flag, sizeof(int), hipMemcpyHostToDevice, c->stream);
  COMM_NCCL(c, R, R->AllReduce(c->d_flag, c->d_flag, 1, ncclInt32, ncclMin, c->comm, c->stream));
  COMM_HIP(c, hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  COMM_HIP(c, hipStreamSynchronize(c->stream));
  if (e_in != hipSuccess) { c->err = std::string("hipMemcpyAsync (agreement flag): ") + hipGetErrorString(e_in); return OFDG_EHIP; }
  *all_ok = flag == 1;
  return OFDG_OK;
}

// Replicate the root's resident texture pool into every rank's (already allocated, identically shaped) pool:
// one ncclBroadcast over xGMI straight between the HBM pools (TextureCollection is then read from disk by the
// root only, DG:117-149).
int ofdg_comm_bcast_pool(ofdg_comm* c, int root, ofdg_ctx* ctx) {
  if (!c || !ctx || root < 0 || root >= c->world) return OFDG_EINVAL;
  Rccl* R = rccl(&c->err);
  if (!R) return OFDG_EHIP;
  // every buffer this rank takes part with, looked up BEFORE the first payload broadcast ...
  void *ptr = nullptr, *ptr2 = nullptr;
  unsigned long long bytes = 0, bytes2 = 0;
  std::vector<std::pair<void*, unsigned long long>> images;
  int rc = ofdg_pool_device(ctx, &ptr, &bytes, c->rank != root);
  if (rc == OFDG_ETEXTURES) rc = ofdg_pool_device_mixed(ctx, &ptr, &bytes, &ptr2, &bytes2);  // images of different sizes
  if (rc != OFDG_OK) c->err = std::string("ofdg_pool_device: ") + ofdg_last_error(ctx);
  if (rc == OFDG_OK && ptr2 && ofdg_ctx_params(ctx)->background_prep) {  // mixed pool + background preparation: the whole images too
    int n = 0;
    ofdg_pool_info(ctx, &n, nullptr, nullptr);
    for (int i = 0; i < n && rc == OFDG_OK; ++i) {
      void* ip = nullptr;
      unsigned long long ib = 0;
      rc = ofdg_pool_device_image(ctx, i, &ip, &ib);
      if (rc != OFDG_OK) c->err = std::string("ofdg_pool_device_image: ") + ofdg_last_error(ctx);
      else images.emplace_back(ip, ib);
    }
  }
  if (hipSetDevice(c->device) != hipSuccess && rc == OFDG_OK) { c->err = "hipSetDevice failed"; rc = OFDG_EHIP; }  // (no early return in front of the agreement)
  // ... and the ranks agree that everybody can: a rank that returned early would leave the others in ncclBroadcast
  bool all_ok = false;
  { int rca = comm_agree(c, R, rc == OFDG_OK, &all_ok); if (rca != OFDG_OK) return rca; }
  if (!all_ok) {
    if (rc == OFDG_OK) { c->err = "ofdg_comm_bcast_pool: another rank could not provide its pool buffers"; rc = OFDG_ETEXTURES; }
    return rc;
  }
  COMM_NCCL(c, R, R->Broadcast(ptr, ptr, (size_t)bytes, ncclUint8, root, c->comm, c->stream));
  if (ptr2) COMM_NCCL(c, R, R->Broadcast(ptr2, ptr2, (size_t)bytes2, ncclUint8, root, c->comm, c->stream));
  for (const auto& im : images) COMM_NCCL(c, R, R->Broadcast(im.first, im.first, (size_t)im.second, ncclUint8, root, c->comm, c->stream));
  COMM_HIP(c, hipStreamSynchronize(c->stream));
  return OFDG_OK;
}

int ofdg_comm_agree(ofdg_comm* c, int local_ok) {
  if (!c) return OFDG_EINVAL;
  Rccl* R = rccl(&c->err);
  if (!R) return OFDG_EHIP;
  bool all_ok = false;
  const int rc = comm_agree(c, R, local_ok != 0, &all_ok);
  if (rc != OFDG_OK) return rc;
  if (!all_ok) {
    c->err = local_ok ? "multi-GPU start-up: another rank failed (see its message)" : "multi-GPU start-up: this rank failed";
    return OFDG_ESTARTUP;
  }
  return OFDG_OK;
}

int ofdg_setup_params(const ofdg_setup* su, const ofdg_comm* c, ofdg_params* p) {
  if (!su || !p) return OFDG_EINVAL;
  ofdg_default_params(p);
  p->seed = su->seed; p->mode = su->mode; p->width = su->width; p->height = su->height; p->num_objects = su->num_objects;
  p->use_antialiasing = su->use_antialiasing; p->batch_size = su->batch_size; p->sampler = su->sampler;
  p->background_prep = su->background_prep; p->max_shapes_per_sample = su->max_shapes_per_sample;
  if (c) { p->rank = c->rank; p->world_size = c->world; p->device = c->device; }
  return OFDG_OK;
}

}  // extern "C"
