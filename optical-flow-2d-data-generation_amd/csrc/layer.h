// C++ host mirror of the reference's Caffe layer surface for this path:
// caffe::DataGenerationLayer<float> (include/caffe/layers/data_generation_layer.hpp:37-89,
// src/caffe/layers/data_generation_layer.cpp) -- same method names, blob shapes and
// error behaviour (std::runtime_error with the reference's messages), constructed
// from the layer's prototxt text instead of a caffe::LayerParameter (Caffe and
// protobuf are not part of this build).  Tops live in HBM.
#pragma once
#include <stddef.h>

#include <string>
#include <vector>

#include "../../include/ofdg.h"

namespace ofdg {

// Minimal stand-in for caffe::Blob<float>: a shape and a device buffer.
class Blob {
 public:
  Blob() {}
  ~Blob();
  Blob(const Blob&) = delete;
  Blob& operator=(const Blob&) = delete;
  void Reshape(const std::vector<int>& shape);  // (re)allocates device memory (unless the data is external)
  // caffe::Blob::set_gpu_data: point the blob at external device memory (not owned; the shape stays)
  void set_gpu_data(float* data);
  const std::vector<int>& shape() const { return shape_; }
  size_t count() const { return count_; }
  size_t offset(int n, int c = 0, int h = 0, int w = 0) const;
  const float* gpu_data() const { return data_; }
  float* mutable_gpu_data() { return data_; }

 private:
  std::vector<int> shape_;
  size_t count_ = 0, capacity_ = 0;
  float* data_ = nullptr;
  bool external_ = false;
};

// What the prototxt subset parser extracts (src/caffe/proto/caffe.proto:6-12 and
// the LMB data_param fields used at data_generation_layer.cpp:44, 109-113).
struct LayerConfig {
  std::string name, type;
  std::vector<std::string> top;
  ofdg_params params;
  std::string texture_dbases;  // list file, or "synthetic:N:W:H[:seed]"
};

// Parses one `layer { ... }` block (protobuf text format subset: nested messages,
// key: value, strings, numbers, true/false, '#' comments).  Throws std::runtime_error.
LayerConfig parse_layer_prototxt(const std::string& text);

class DataGenerationLayer {
 public:
  // comm != nullptr: one layer per GPU/process on an RCCL communicator - rank 0's options and texture collection
  // are broadcast (ofdg_comm_bcast_setup), every rank renders its own shard of each global batch.
  explicit DataGenerationLayer(const std::string& layer_prototxt, ofdg_comm* comm = nullptr);
  ~DataGenerationLayer();

  void LayerSetUp(const std::vector<Blob*>& bottom, const std::vector<Blob*>& top);
  void Reshape(const std::vector<Blob*>&, const std::vector<Blob*>&) {}
  inline bool ShareInParallel() const { return false; }
  inline const char* type() const { return "DataGeneration"; }
  inline int ExactNumBottomBlobs() const { return 0; }
  inline int MinTopBlobs() const { return 1; }

  // The reference's Forward_gpu defers to Forward_cpu (data_generation_layer.cpp:286-291);
  // here both produce the batch on the GPU.  data_param.prefetch = 1: rendered into the top
  // blobs' own memory.  prefetch = P > 1 (the reference's prefetch thread and its queue of P
  // batches, data_generation_layer.cpp:36-56, 141-172, 266-282): P device buffer sets are
  // cycled, P - 1 batches are rendered ahead on the context's internal streams (from
  // LayerSetUp on) while the caller works on the current one, and Forward points the top
  // blobs at the finished set instead of copying it (valid until the next Forward).
  // Forward waits for the OLDEST set's own completion event only (prefetch_full_.pop,
  // data_generation_layer.cpp:269): the younger batches stay in flight behind it.
  // A batch a kernel truncated (a device capacity flag of its own ticket) makes Forward throw AFTER the batch is retired:
  // the tops then point at that batch's buffer set - stable until the next Forward, its contents incomplete - and the
  // next Forward hands out the next batch.
  void Forward_cpu(const std::vector<Blob*>& bottom, const std::vector<Blob*>& top);
  void Forward_gpu(const std::vector<Blob*>& bottom, const std::vector<Blob*>& top);
  void Backward_cpu(const std::vector<Blob*>&, const std::vector<bool>&, const std::vector<Blob*>&) {}
  void Backward_gpu(const std::vector<Blob*>&, const std::vector<bool>&, const std::vector<Blob*>&) {}

  const LayerConfig& config() const { return cfg_; }
  ofdg_ctx* context() { return ctx_; }

 private:
  void enqueue_next();
  LayerConfig cfg_;
  ofdg_ctx* ctx_ = nullptr;
  std::vector<float*> ring_;       // [prefetch][3] device buffers (prefetch > 1)
  std::vector<void*> ring_done_;   // [prefetch] hipEvent_t: the set's batch is complete
  std::vector<long long> ring_ticket_;  // ... and the batch's number in the context (its own device error word)
 public:
  // batches rendered ahead that had not finished when the last Forward returned (diagnostics / tests)
  int in_flight_after_last_forward() const { return in_flight_; }
 private:
  int in_flight_ = 0;
  long long produced_ = 0, consumed_ = 0;
};

// Texture list loader: TextureCollection (DataGenerator.cpp:117-149) for binary PPM
// (P6) files, or a synthetic pool.  Throws std::runtime_error("Could not open texture
// collection") like the reference.
void load_texture_collection(ofdg_ctx* ctx, const std::string& spec);

}  // namespace ofdg
