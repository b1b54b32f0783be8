#include "layer.h"

#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iterator>
#include <memory>
#include <sstream>
#include <stdexcept>

#include "realize.h"
#include "sampler_ref.h"

namespace ofdg {

// ---------------------------------------------------------------------------
// Blob
// ---------------------------------------------------------------------------
Blob::~Blob() {
  if (data_ && !external_) (void)hipFree(data_);
}
void Blob::set_gpu_data(float* data) {
  if (data_ && !external_) (void)hipFree(data_);
  data_ = data;
  external_ = true;
  capacity_ = 0;
}
void Blob::Reshape(const std::vector<int>& shape) {
  size_t n = 1;
  for (int d : shape) {
    if (d < 0) throw std::runtime_error("Blob::Reshape: negative dimension");
    n *= (size_t)d;
  }
  shape_ = shape;
  count_ = n;
  if (external_) return;  // (the owner of the memory sized it)
  if (n > capacity_) {
    if (data_) (void)hipFree(data_);
    data_ = nullptr;
    if (hipMalloc((void**)&data_, n * sizeof(float)) != hipSuccess) throw std::runtime_error("Blob::Reshape: hipMalloc failed");
    capacity_ = n;
  }
}
size_t Blob::offset(int n, int c, int h, int w) const {
  size_t o = (size_t)n;
  o = o * (shape_.size() > 1 ? shape_[1] : 1) + c;
  o = o * (shape_.size() > 2 ? shape_[2] : 1) + h;
  o = o * (shape_.size() > 3 ? shape_[3] : 1) + w;
  return o;
}

// ---------------------------------------------------------------------------
// prototxt subset parser
// ---------------------------------------------------------------------------
namespace {
struct Tok {
  enum Kind { kIdent, kString, kNumber, kLBrace, kRBrace, kColon, kEnd } kind;
  std::string text;
};
class Lexer {
 public:
  explicit Lexer(const std::string& s) : s_(s) {}
  Tok next() {
    for (;;) {
      while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_;
      if (i_ < s_.size() && s_[i_] == '#') { while (i_ < s_.size() && s_[i_] != '\n') ++i_; continue; }
      break;
    }
    if (i_ >= s_.size()) return {Tok::kEnd, ""};
    const char ch = s_[i_];
    if (ch == '{') { ++i_; return {Tok::kLBrace, "{"}; }
    if (ch == '}') { ++i_; return {Tok::kRBrace, "}"}; }
    if (ch == ':') { ++i_; return {Tok::kColon, ":"}; }
    if (ch == '"' || ch == '\'') {
      const char q = ch;
      std::string v;
      ++i_;
      while (i_ < s_.size() && s_[i_] != q) {
        if (s_[i_] == '\\' && i_ + 1 < s_.size()) ++i_;
        v += s_[i_++];
      }
      if (i_ >= s_.size()) throw std::runtime_error("prototxt: unterminated string");
      ++i_;
      return {Tok::kString, v};
    }
    if (std::isalpha((unsigned char)ch) || ch == '_') {
      std::string v;
      while (i_ < s_.size() && (std::isalnum((unsigned char)s_[i_]) || s_[i_] == '_')) v += s_[i_++];
      return {Tok::kIdent, v};
    }
    if (std::isdigit((unsigned char)ch) || ch == '-' || ch == '+' || ch == '.') {
      std::string v;
      while (i_ < s_.size() && (std::isalnum((unsigned char)s_[i_]) || s_[i_] == '-' || s_[i_] == '+' || s_[i_] == '.')) v += s_[i_++];
      return {Tok::kNumber, v};
    }
    throw std::runtime_error(std::string("prototxt: unexpected character '") + ch + "'");
  }

 private:
  const std::string& s_;
  size_t i_ = 0;
};

int to_int(const Tok& t, const std::string& key) {
  if (t.kind == Tok::kIdent && (t.text == "true" || t.text == "false")) return t.text == "true";
  if (t.kind != Tok::kNumber) throw std::runtime_error("prototxt: expected a number for " + key);
  return (int)std::strtol(t.text.c_str(), nullptr, 10);
}

void parse_message(Lexer& lx, const std::string& scope, LayerConfig* cfg, bool top_level) {
  for (;;) {
    Tok k = lx.next();
    if (k.kind == Tok::kEnd) {
      if (!top_level) throw std::runtime_error("prototxt: missing '}'");
      return;
    }
    if (k.kind == Tok::kRBrace) {
      if (top_level) throw std::runtime_error("prototxt: unbalanced '}'");
      return;
    }
    if (k.kind != Tok::kIdent) throw std::runtime_error("prototxt: expected a field name");
    Tok v = lx.next();
    if (v.kind == Tok::kColon) v = lx.next();
    if (v.kind == Tok::kLBrace) {
      parse_message(lx, scope.empty() ? k.text : scope + "." + k.text, cfg, false);
      continue;
    }
    const std::string key = scope.empty() ? k.text : scope + "." + k.text;
    ofdg_params& p = cfg->params;
    if (key == "layer.name" || key == "name") cfg->name = v.text;
    else if (key == "layer.type" || key == "type") cfg->type = v.text;
    else if (key == "layer.top" || key == "top") cfg->top.push_back(v.text);
    else if (key == "layer.data_param.batch_size" || key == "data_param.batch_size") p.batch_size = to_int(v, key);
    else if (key == "layer.data_param.prefetch" || key == "data_param.prefetch") p.prefetch = to_int(v, key);
    else if (key.find("data_generation_param.") != std::string::npos) {
      const std::string f = key.substr(key.rfind('.') + 1);
      if (f == "mode") p.mode = to_int(v, key);
      else if (f == "texture_dbases") { if (cfg->texture_dbases.empty()) cfg->texture_dbases = v.text; }
      else if (f == "first_level_threads") p.first_level_threads = to_int(v, key);
      else if (f == "second_level_threads") p.second_level_threads = to_int(v, key);
      else if (f == "use_antialiasing") p.use_antialiasing = to_int(v, key);
      // extension keys (not in the reference's proto)
      else if (f == "width") p.width = to_int(v, key);
      else if (f == "height") p.height = to_int(v, key);
      else if (f == "num_objects") p.num_objects = to_int(v, key);
      else if (f == "seed") p.seed = to_int(v, key);
      else if (f == "chains") p.chains = to_int(v, key);        // scheduling (extension keys): internal streams,
      else if (f == "lookahead") p.lookahead = to_int(v, key);  // batches prepared ahead of the Forward that composes them
      else if (f == "background_prep")  // true / 1: the CImg chain stage by stage; fast / 2: one resampling; false / 0: centre crop
        p.background_prep = (v.text == "true" || v.text == "1") ? 1 : (v.text == "fast" || v.text == "2") ? 2 : 0;
      else if (f == "sampler") p.sampler = (v.text == "counter") ? OFDG_SAMPLER_COUNTER : OFDG_SAMPLER_REF;
      else throw std::runtime_error("prototxt: unknown data_generation_param field '" + f + "'");
    }
    // other fields (bottom, include, data_param.verbose ...) are accepted and ignored
  }
}
}  // namespace

LayerConfig parse_layer_prototxt(const std::string& text) {
  LayerConfig cfg;
  ofdg_default_params(&cfg.params);
  // the reference always runs getRandomizedCrop(2W, 2H, rot, zoom, shift) on the background (DataGenerator.cpp:1186-1192):
  // the layer does too unless the prototxt says `background_prep: false` (extension key)
  cfg.params.background_prep = 1;
  Lexer lx(text);
  parse_message(lx, "", &cfg, true);
  return cfg;
}

// ---------------------------------------------------------------------------
// texture collection
// ---------------------------------------------------------------------------
namespace {
bool read_ppm(const std::string& path, std::vector<uint8_t>* planar_bgr, int* w, int* h) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return false;
  std::string magic;
  f >> magic;
  if (magic != "P6") return false;
  auto next_int = [&](int* out) {
    for (;;) {
      int c = f.peek();
      if (c == '#') { std::string line; std::getline(f, line); continue; }
      if (std::isspace(c)) { f.get(); continue; }
      break;
    }
    f >> *out;
    return !f.fail();
  };
  int maxv = 0;
  if (!next_int(w) || !next_int(h) || !next_int(&maxv) || maxv != 255 || *w <= 0 || *h <= 0) return false;
  f.get();  // single whitespace after maxval
  std::vector<uint8_t> rgb((size_t)*w * *h * 3);
  f.read((char*)rgb.data(), (std::streamsize)rgb.size());
  if ((size_t)f.gcount() != rgb.size()) return false;
  const size_t n = (size_t)*w * *h;
  planar_bgr->resize(3 * n);
  for (size_t i = 0; i < n; ++i) {  // CImg planar R,G,B then swap(c0, c2) (DataGenerator.cpp:129-131)
    (*planar_bgr)[i] = rgb[3 * i + 2];
    (*planar_bgr)[n + i] = rgb[3 * i + 1];
    (*planar_bgr)[2 * n + i] = rgb[3 * i + 0];
  }
  return true;
}

// PNG through the system's libpng 1.6, bound at run time (dlopen: the library is part of the image, a build dependency on
// it is not wanted).  Its "simplified API" (png.h 1.6: png_image_begin_read_from_memory / png_image_finish_read /
// png_image_free over a caller-owned png_image) is a stable C ABI; the struct below restates png_image field by field.
// 8-bit R, G, B, A come back as stored (alpha is read and dropped: CImg's load keeps it as a fourth channel the
// reference never looks at, DataGenerator.cpp:128-131); palette, grey and 16-bit files are expanded by libpng.
struct PngImage {
  void* opaque;
  uint32_t version, width, height, format, flags, colormap_entries, warning_or_error;
  char message[64];
};
struct PngApi {
  int (*begin_read_from_memory)(PngImage*, const void*, size_t) = nullptr;
  int (*finish_read)(PngImage*, const void* background, void* buffer, int32_t row_stride, void* colormap) = nullptr;
  void (*image_free)(PngImage*) = nullptr;
  bool ok = false;
  PngApi() {
    void* h = nullptr;
    for (const char* name : {"libpng16.so.16", "libpng16.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return;
    begin_read_from_memory = (decltype(begin_read_from_memory))dlsym(h, "png_image_begin_read_from_memory");
    finish_read = (decltype(finish_read))dlsym(h, "png_image_finish_read");
    image_free = (decltype(image_free))dlsym(h, "png_image_free");
    ok = begin_read_from_memory && finish_read && image_free;
  }
};
const PngApi& png_api() { static const PngApi api; return api; }
constexpr uint32_t kPngImageVersion = 1, kPngFormatRgba = 0x03;  // PNG_IMAGE_VERSION; PNG_FORMAT_FLAG_ALPHA | PNG_FORMAT_FLAG_COLOR

bool is_png(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  unsigned char sig[8] = {0};
  f.read((char*)sig, 8);
  static const unsigned char want[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  return f.gcount() == 8 && std::memcmp(sig, want, 8) == 0;
}
// libpng's simplified API hands out 8-bit sRGB samples: it honours the file's colour-management chunks (a gAMA that is not
// sRGB's re-encodes every sample), while the reference's CImg::load (DataGenerator.cpp:128) keeps the raw sample values with
// no gamma handling.  So the file is decoded from MEMORY with those chunks - gAMA, cHRM, sRGB, iCCP: ancillary, each chunk
// carries its own CRC - left out: libpng then takes 8-bit samples as what they are, and the pool holds the bytes the
// reference's holds, whatever the file says about its gamma.  16 bits per sample stay refused, with the way out in the
// message: CImg would hand the reference's `unsigned char` image the truncated 16-bit values, libpng a conversion from linear
// light - neither is a texture anybody meant.
bool png_without_colour_chunks(const std::string& path, std::vector<unsigned char>* out, std::string* why) {
  std::ifstream f(path, std::ios::binary);
  std::vector<unsigned char> in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  auto be32 = [](const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; };
  if (in.size() < 8) { *why = "truncated PNG"; return false; }
  out->assign(in.begin(), in.begin() + 8);
  size_t i = 8;
  while (i + 12 <= in.size()) {
    const uint32_t len = be32(&in[i]);
    const std::string type((const char*)&in[i + 4], 4);
    if ((size_t)len + 12 > in.size() - i) break;
    if (type == "IHDR" && len >= 9 && in[i + 8 + 8] == 16) {
      *why = "16-bit PNG: the reference's 8-bit texture would hold its truncated samples; convert the texture to 8 bit (tools/convert_textures.py)";
      return false;
    }
    if (type != "gAMA" && type != "cHRM" && type != "sRGB" && type != "iCCP") out->insert(out->end(), in.begin() + i, in.begin() + i + 12 + len);
    i += 12 + (size_t)len;
    if (type == "IEND") return true;
  }
  *why = "truncated PNG";
  return false;
}
// planar_bgr == nullptr: the size only
bool read_png(const std::string& path, std::vector<uint8_t>* planar_bgr, int* w, int* h, std::string* why) {
  const PngApi& api = png_api();
  if (!api.ok) { *why = "libpng16 is not available on this system"; return false; }
  std::vector<unsigned char> file;
  if (!png_without_colour_chunks(path, &file, why)) return false;
  PngImage img;
  std::memset(&img, 0, sizeof(img));
  img.version = kPngImageVersion;
  if (!api.begin_read_from_memory(&img, file.data(), file.size())) { *why = img.message; return false; }
  *w = (int)img.width; *h = (int)img.height;
  if (!planar_bgr) { api.image_free(&img); return true; }
  img.format = kPngFormatRgba;
  const size_t n = (size_t)img.width * img.height;
  std::vector<uint8_t> rgba(n * 4);
  if (!api.finish_read(&img, nullptr, rgba.data(), 0, nullptr)) { *why = img.message; api.image_free(&img); return false; }
  planar_bgr->resize(3 * n);
  for (size_t i = 0; i < n; ++i) {  // planar, R <-> B swapped like the PPM path (DataGenerator.cpp:129-131)
    (*planar_bgr)[i] = rgba[4 * i + 2];
    (*planar_bgr)[n + i] = rgba[4 * i + 1];
    (*planar_bgr)[2 * n + i] = rgba[4 * i + 0];
  }
  return true;
}
// An image file of a texture list: binary PPM or PNG, by its first bytes.  planar_bgr == nullptr: the size only.
bool read_image(const std::string& path, std::vector<uint8_t>* planar_bgr, int* w, int* h, std::string* why) {
  if (is_png(path)) return read_png(path, planar_bgr, w, h, why);
  std::vector<uint8_t> scratch;
  if (read_ppm(path, planar_bgr ? planar_bgr : &scratch, w, h)) return true;
  *why = "neither a binary PPM (P6, maxval 255) nor a PNG";
  return false;
}
}  // namespace

void load_texture_collection(ofdg_ctx* ctx, const std::string& spec) {
  if (spec.compare(0, 10, "synthetic:") == 0) {
    int n = 0, w = 0, h = 0;
    unsigned seed = 0;
    if (std::sscanf(spec.c_str(), "synthetic:%d:%d:%d:%u", &n, &w, &h, &seed) < 3)
      throw std::runtime_error("Could not open texture collection (bad synthetic spec)");
    if (ofdg_pool_synthetic(ctx, n, w, h, seed) != OFDG_OK)
      throw std::runtime_error(std::string("Could not open texture collection: ") + ofdg_last_error(ctx));
    return;
  }
  std::ifstream infile(spec);
  if (infile.bad() || !infile.is_open()) throw std::runtime_error("Could not open texture collection");  // DataGenerator.cpp:121
  std::vector<std::string> paths;
  std::string imagepath;
  while (!infile.eof()) {  // reference loop: a last line without '\n' is dropped (DataGenerator.cpp:124-126)
    std::getline(infile, imagepath);
    if (infile.eof()) break;
    paths.push_back(imagepath);
  }
  if (paths.empty()) throw std::runtime_error("Could not open texture collection (no images listed)");
  // images of one size: the pool keeps them whole; of different sizes: every image is reduced to the two
  // textures the path reads (ofdg_pool_alloc_mixed)
  std::vector<std::vector<uint8_t>> first(1);
  int pw = 0, ph = 0;
  bool mixed = false;
  // headers decide; every file that cannot be used is named in ONE error (a collection with a few 16-bit PNGs is fixed in one go)
  std::string unreadable;
  int n_unreadable = 0;
  for (size_t i = 0; i < paths.size(); ++i) {
    int w = 0, h = 0;
    std::string why;
    bool ok = true;
    if (is_png(paths[i])) {
      ok = read_png(paths[i], nullptr, &w, &h, &why);
    } else {
      std::ifstream f(paths[i], std::ios::binary);
      std::string magic;
      if (!f.is_open() || !(f >> magic) || magic != "P6") { ok = false; why = "neither a binary PPM nor a PNG"; }
      for (int k = 0; ok && k < 2; ++k) {
        for (;;) { const int ch = f.peek(); if (ch == '#') { std::string line; std::getline(f, line); } else if (std::isspace(ch)) f.get(); else break; }
        f >> (k == 0 ? w : h);
      }
    }
    if (!ok) {
      if (++n_unreadable <= 16) unreadable += (unreadable.empty() ? "" : "; ") + paths[i] + ": " + why;
      continue;
    }
    if (pw == 0 && ph == 0) { pw = w; ph = h; } else if (w != pw || h != ph) mixed = true;
  }
  if (n_unreadable)
    throw std::runtime_error("Could not open texture collection (cannot read " + std::string(n_unreadable == 1 ? "" : std::to_string(n_unreadable) + " files: ") + unreadable +
                             (n_unreadable > 16 ? "; ..." : "") + ")");
  const int rc_alloc = mixed ? ofdg_pool_alloc_mixed(ctx, (int)paths.size()) : ofdg_pool_alloc(ctx, (int)paths.size(), pw, ph);
  if (rc_alloc != OFDG_OK) throw std::runtime_error(std::string("Could not open texture collection: ") + ofdg_last_error(ctx));
  for (size_t i = 0; i < paths.size(); ++i) {
    std::vector<uint8_t> img;
    int w = 0, h = 0;
    std::string why;
    if (!read_image(paths[i], &img, &w, &h, &why)) throw std::runtime_error("Could not open texture collection (cannot read " + paths[i] + ": " + why + ")");
    const int rc = mixed ? ofdg_pool_upload_mixed(ctx, (int)i, img.data(), w, h) : ofdg_pool_upload(ctx, (int)i, img.data(), w, h);
    if (rc != OFDG_OK) throw std::runtime_error(std::string("Could not open texture collection: ") + ofdg_last_error(ctx));
  }
}

// ---------------------------------------------------------------------------
// DataGenerationLayer
// ---------------------------------------------------------------------------
DataGenerationLayer::DataGenerationLayer(const std::string& layer_prototxt, ofdg_comm* comm) : cfg_(parse_layer_prototxt(layer_prototxt)) {
  if (!cfg_.type.empty() && cfg_.type != "DataGeneration") throw std::runtime_error("layer type is not \"DataGeneration\"");
  constexpr int kTableCap = 65536;
  std::vector<ofdg_tex_entry> table;
  ofdg_setup su;
  std::memset(&su, 0, sizeof(su));
  const int rank = comm ? ofdg_comm_rank(comm) : 0;
  std::exception_ptr local_failure;
  auto create = [&]() {
    int rc = ofdg_create(&cfg_.params, &ctx_);
    if (rc == OFDG_EBADMODE) throw std::runtime_error("BAD MODE");  // DataGenerator.cpp:2004
    if (rc != OFDG_OK) throw std::runtime_error(std::string("DataGenerationLayer: ") + ofdg_last_error(nullptr));
  };
  auto bcast = [&]() {  // the one start-up collective: rank 0's stream + pool description
    table.resize(kTableCap);
    if (ofdg_comm_bcast_setup(comm, 0, &su, table.data(), kTableCap) != OFDG_OK)
      throw std::runtime_error(std::string("DataGenerationLayer: ") + ofdg_comm_last_error(comm));
  };
  try {
    if (comm) {
      ofdg_params mine;
      ofdg_setup_params(&su, comm, &mine);  // (rank, world_size, device of the communicator)
      cfg_.params.rank = mine.rank; cfg_.params.world_size = mine.world_size; cfg_.params.device = mine.device;
    }
    if (!comm || rank == 0) {
      try {
        create();
        load_texture_collection(ctx_, cfg_.texture_dbases);  // DataGenerator ctor -> TextureCollection (DataGenerator.cpp:992)
      } catch (...) {
        // the other ranks are waiting in the start-up broadcast: tell them that it failed, then fail here
        if (comm) (void)ofdg_comm_bcast_abort(comm, 0, OFDG_ETEXTURES, kTableCap);
        throw;
      }
      if (comm) {
        table.resize(kTableCap);
        int rcs = ofdg_setup_of(ctx_, &su, table.data(), kTableCap);
        if (rcs != OFDG_OK) su.status = rcs;  // (travels with the broadcast: every rank fails together)
        bcast();
      }
    } else {
      bcast();
      // From here to the next collective this rank works alone (context, pool allocation): if that fails the others
      // must not be left waiting in the collective - the failure is kept and every rank learns of it in the agreement.
      try {
        ofdg_params p;
        ofdg_setup_params(&su, comm, &p);
        p.prefetch = cfg_.params.prefetch;
        p.first_level_threads = cfg_.params.first_level_threads; p.second_level_threads = cfg_.params.second_level_threads;
        cfg_.params = p;
        create();
        if (ofdg_setup_alloc_pool(ctx_, &su, table.data()) != OFDG_OK)
          throw std::runtime_error(std::string("Could not open texture collection: ") + ofdg_last_error(ctx_));
      } catch (...) {
        local_failure = std::current_exception();
      }
    }
    if (comm) {  // every rank that came through the broadcast: did everybody set itself up?
      const int rca = ofdg_comm_agree(comm, local_failure ? 0 : 1);
      if (local_failure) std::rethrow_exception(local_failure);
      if (rca != OFDG_OK) throw std::runtime_error(std::string("DataGenerationLayer: ") + ofdg_comm_last_error(comm));
    }
    // a texture collection read from disk lives on rank 0 only until here: replicate it over xGMI
    if (comm && su.pool_kind != OFDG_POOL_SYNTHETIC && ofdg_comm_bcast_pool(comm, 0, ctx_) != OFDG_OK)
      throw std::runtime_error(std::string("DataGenerationLayer: ") + ofdg_comm_last_error(comm));
    // DataGenerator::Start launches the CropGenerator for MODE == 9 (DataGenerator.cpp:1016-1020)
    if (cfg_.params.mode == 9 && ofdg_warp_generate(ctx_, 2, (uint32_t)cfg_.params.seed) != OFDG_OK)
      throw std::runtime_error(std::string("warp field generation: ") + ofdg_last_error(ctx_));
  } catch (...) {
    if (ctx_) ofdg_destroy(ctx_);
    ctx_ = nullptr;
    throw;
  }
}

DataGenerationLayer::~DataGenerationLayer() {
  if (ctx_) (void)ofdg_synchronize(ctx_, nullptr);
  for (float* p : ring_) if (p) (void)hipFree(p);
  for (void* e : ring_done_) if (e) (void)hipEventDestroy((hipEvent_t)e);
  ofdg_destroy(ctx_);
}

// render the next batch of the stream into its buffer set, on the context's next internal stream
void DataGenerationLayer::enqueue_next() {
  const int P = (int)ring_.size() / 3;
  float** set = &ring_[(size_t)(produced_ % P) * 3];
  void* chain = ofdg_stream(ctx_);  // the whole batch runs in order on this internal stream
  if (ofdg_forward(ctx_, set[0], set[1], set[2], chain) != OFDG_OK)
    throw std::runtime_error(std::string("DataGenerationLayer::Forward: ") + ofdg_last_error(ctx_));
  if (hipEventRecord((hipEvent_t)ring_done_[(size_t)(produced_ % P)], (hipStream_t)chain) != hipSuccess)
    throw std::runtime_error("DataGenerationLayer::Forward: hipEventRecord failed");
  ring_ticket_[(size_t)(produced_ % P)] = ofdg_last_ticket(ctx_);
  ++produced_;
}

void DataGenerationLayer::LayerSetUp(const std::vector<Blob*>& bottom, const std::vector<Blob*>& top) {
  if (!bottom.empty()) throw std::runtime_error("DataGeneration takes no bottom blobs");  // ExactNumBottomBlobs() == 0
  if (top.size() != 3) throw std::runtime_error("DataGeneration produces exactly 3 top blobs");  // load_batch indexes output[0..2]
  const int N = cfg_.params.batch_size, H = cfg_.params.height, W = cfg_.params.width;
  // StartInternalThread (data_generation_layer.cpp:132): the first prefetch - 1 batches start rendering now
  const int P = cfg_.params.prefetch;
  if (P > 1 && ring_.empty()) {
    ring_.assign((size_t)P * 3, nullptr);
    for (int k = 0; k < P * 3; ++k)
      if (hipMalloc((void**)&ring_[k], (size_t)N * (k % 3 == 2 ? 2 : 3) * H * W * sizeof(float)) != hipSuccess)
        throw std::runtime_error("DataGenerationLayer: hipMalloc of the prefetch buffers failed");
    ring_done_.assign((size_t)P, nullptr);
    ring_ticket_.assign((size_t)P, -1);
    for (int k = 0; k < P; ++k)
      if (hipEventCreateWithFlags((hipEvent_t*)&ring_done_[k], hipEventDisableTiming) != hipSuccess)
        throw std::runtime_error("DataGenerationLayer: hipEventCreate failed");
    while (produced_ < P - 1) enqueue_next();
  }
  if (!ring_.empty())
    for (int k = 0; k < 3; ++k) top[k]->set_gpu_data(ring_[k]);  // (the tops never own memory in this mode)
  top[0]->Reshape({N, 3, H, W});  // data_generation_layer.cpp:128-130
  top[1]->Reshape({N, 3, H, W});
  top[2]->Reshape({N, 2, H, W});
}

void DataGenerationLayer::Forward_gpu(const std::vector<Blob*>& bottom, const std::vector<Blob*>& top) {
  (void)bottom;
  if (top.size() != 3) throw std::runtime_error("DataGeneration produces exactly 3 top blobs");
  const int N = cfg_.params.batch_size, H = cfg_.params.height, W = cfg_.params.width;
  top[0]->Reshape({N, 3, H, W});
  top[1]->Reshape({N, 3, H, W});
  top[2]->Reshape({N, 2, H, W});
  if (!ring_.empty()) {
    // prefetch_full_.pop (data_generation_layer.cpp:269): the oldest batch in flight - finished long ago when the
    // caller's own work takes longer than a render - becomes the tops; its successor starts rendering at once
    const int P = (int)ring_.size() / 3;
    if (produced_ == consumed_) enqueue_next();
    // wait for THIS set's event only: the batches behind it keep rendering
    if (hipEventSynchronize((hipEvent_t)ring_done_[(size_t)(consumed_ % P)]) != hipSuccess)
      throw std::runtime_error("DataGenerationLayer::Forward: hipEventSynchronize failed");
    // THIS batch's device error flags (a flag raised by a younger batch still rendering is reported at that batch's own turn).
    // The word is cleared by the read, so the error is reported once.  A truncated batch is RETIRED like a good one before
    // the exception leaves: the tops point at ITS buffer set - the one set no batch in flight renders into until the next
    // Forward, so what a caller that catches the exception still reads there is stable (and marked bad by the exception:
    // the reference drops a bad sample and leaves stale data in its batch slot, DG:1285-1292) - and its successor starts
    // rendering; the next Forward hands out the NEXT batch.
    std::string bad;
    if (ofdg_poll_errors_of(ctx_, ring_ticket_[(size_t)(consumed_ % P)]) != OFDG_OK) bad = ofdg_last_error(ctx_);
    float** set = &ring_[(size_t)(consumed_ % P) * 3];
    for (int k = 0; k < 3; ++k) top[k]->set_gpu_data(set[k]);
    ++consumed_;
    in_flight_ = 0;
    for (long long b = consumed_; b < produced_; ++b)
      if (hipEventQuery((hipEvent_t)ring_done_[(size_t)(b % P)]) == hipErrorNotReady) ++in_flight_;
    try {
      while (produced_ < consumed_ + P - 1) enqueue_next();
    } catch (const std::exception& e) {  // (the batch's own error comes first: a failed enqueue shows again at the next Forward)
      if (bad.empty()) throw;
      bad += std::string(" (and the next batch could not be enqueued: ") + e.what() + ")";
    }
    if (!bad.empty()) throw std::runtime_error("DataGenerationLayer::Forward: " + bad);
    return;
  }
  int rc = ofdg_forward(ctx_, top[0]->mutable_gpu_data(), top[1]->mutable_gpu_data(), top[2]->mutable_gpu_data(), nullptr);
  if (rc == OFDG_OK) rc = ofdg_synchronize(ctx_, nullptr);
  if (rc != OFDG_OK) throw std::runtime_error(std::string("DataGenerationLayer::Forward: ") + ofdg_last_error(ctx_));
}

void DataGenerationLayer::Forward_cpu(const std::vector<Blob*>& bottom, const std::vector<Blob*>& top) {
  Forward_gpu(bottom, top);
}

}  // namespace ofdg

// ---------------------------------------------------------------------------
// C-ABI wrappers of the host-side pieces (declared in include/ofdg.h)
// ---------------------------------------------------------------------------
using namespace ofdg;

struct ofdg_host_sampler {
  RefSampler s;
  ofdg_host_sampler(int m, int w, int h, int n) : s(m, w, h, n) {}
};
struct ofdg_layer {
  std::unique_ptr<DataGenerationLayer> layer;
  Blob top[3];
  std::string err;
};
static thread_local std::string g_host_error;

extern "C" {

const char* ofdg_host_last_error(void) { return g_host_error.c_str(); }

int ofdg_host_decode_image(const char* path, uint8_t* planar_bgr, size_t capacity, int* width, int* height) {
  if (!path || !width || !height) return OFDG_EINVAL;
  std::vector<uint8_t> img;
  std::string why;
  if (!read_image(path, planar_bgr ? &img : nullptr, width, height, &why)) { g_host_error = std::string("cannot read ") + path + ": " + why; return OFDG_ETEXTURES; }
  if (planar_bgr) {
    if (img.size() > capacity) { g_host_error = "image buffer too small"; return OFDG_ECAPACITY; }
    std::memcpy(planar_bgr, img.data(), img.size());
  }
  return OFDG_OK;
}

int ofdg_host_sampler_create(int mode, int width, int height, int num_objects, ofdg_host_sampler** out) {
  if (!out) return OFDG_EINVAL;
  *out = nullptr;
  std::unique_ptr<ofdg_host_sampler> s(new ofdg_host_sampler(mode, width, height, num_objects));
  if (!s->s.ok()) { g_host_error = "BAD MODE"; return OFDG_EBADMODE; }
  *out = s.release();
  return OFDG_OK;
}
void ofdg_host_sampler_destroy(ofdg_host_sampler* s) { delete s; }
int ofdg_host_sampler_next(ofdg_host_sampler* s, int n_tasks, ofdg_task* tasks, ofdg_blueprint* bps, int cap, int* n_bps) {
  if (!s || !tasks || !bps || !n_bps) return OFDG_EINVAL;
  std::vector<ofdg_blueprint> pool;
  for (int i = 0; i < n_tasks; ++i) {
    int rc = s->s.next_task(&pool, &tasks[i], &g_host_error);
    if (rc != OFDG_OK) return rc;
  }
  *n_bps = (int)pool.size();
  if ((int)pool.size() > cap) { g_host_error = "blueprint capacity exceeded"; return OFDG_ECAPACITY; }
  std::memcpy(bps, pool.data(), pool.size() * sizeof(ofdg_blueprint));
  return OFDG_OK;
}

int ofdg_host_realize(const ofdg_params* prm, int pool_n, int pool_w, int pool_h, const ofdg_task* tasks, int n_tasks,
                      const ofdg_blueprint* bps, int n_bps, double* shape_mats, int shape_cap, int* n_shapes,
                      double* object_mats, int object_cap, int* n_objects) {
  if (!prm || !tasks || !bps || !n_shapes || !n_objects) return OFDG_EINVAL;
  RealizeConfig cfg{prm->width, prm->height, prm->mode, pool_n, pool_w, pool_h};
  RealizedBatch b;
  int rc = realize_batch(cfg, tasks, n_tasks, bps, n_bps, &b, &g_host_error);
  if (rc != OFDG_OK) return rc;
  *n_shapes = (int)b.shapes.size();
  *n_objects = (int)b.objects.size();
  if ((int)b.shapes.size() > shape_cap || (int)b.objects.size() > object_cap) { g_host_error = "capacity"; return OFDG_ECAPACITY; }
  for (size_t i = 0; i < b.shapes.size() && shape_mats; ++i) std::memcpy(shape_mats + 12 * i, b.shapes[i].m, sizeof(double) * 12);
  for (size_t i = 0; i < b.objects.size() && object_mats; ++i) {
    std::memcpy(object_mats + 12 * i, &b.objects[i].motion, sizeof(double) * 6);
    std::memcpy(object_mats + 12 * i + 6, &b.objects[i].tex_inv, sizeof(double) * 6);
  }
  return OFDG_OK;
}

int ofdg_parse_prototxt(const char* text, ofdg_params* out, char* texture_dbases, int cap, int* n_top) {
  if (!text || !out) return OFDG_EINVAL;
  try {
    LayerConfig cfg = parse_layer_prototxt(text);
    *out = cfg.params;
    if (texture_dbases && cap > 0) {
      std::strncpy(texture_dbases, cfg.texture_dbases.c_str(), (size_t)cap - 1);
      texture_dbases[cap - 1] = 0;
    }
    if (n_top) *n_top = (int)cfg.top.size();
    return OFDG_OK;
  } catch (const std::exception& e) {
    g_host_error = e.what();
    return OFDG_EINVAL;
  }
}

int ofdg_layer_create(const char* prototxt, ofdg_layer** out) { return ofdg_layer_create_dist(prototxt, nullptr, out); }
int ofdg_layer_create_dist(const char* prototxt, ofdg_comm* comm, ofdg_layer** out) {
  if (!prototxt || !out) return OFDG_EINVAL;
  *out = nullptr;
  std::unique_ptr<ofdg_layer> L(new ofdg_layer());
  try {
    L->layer.reset(new DataGenerationLayer(prototxt, comm));
    std::vector<Blob*> top = {&L->top[0], &L->top[1], &L->top[2]};
    L->layer->LayerSetUp({}, top);
  } catch (const std::exception& e) {
    g_host_error = e.what();
    if (g_host_error == "BAD MODE") return OFDG_EBADMODE;
    if (g_host_error.find("texture collection") != std::string::npos) return OFDG_ETEXTURES;
    return OFDG_EINVAL;
  }
  *out = L.release();
  return OFDG_OK;
}
void ofdg_layer_destroy(ofdg_layer* L) { delete L; }
int ofdg_layer_in_flight(const ofdg_layer* L) { return L ? L->layer->in_flight_after_last_forward() : OFDG_EINVAL; }
// Forward(): fills the three top blobs and returns their device pointers.
int ofdg_layer_forward(ofdg_layer* L, float** image0, float** image1, float** flow, int* shape4) {
  if (!L) return OFDG_EINVAL;
  try {
    std::vector<Blob*> top = {&L->top[0], &L->top[1], &L->top[2]};
    L->layer->Forward_gpu({}, top);
  } catch (const std::exception& e) {
    g_host_error = e.what();
    return OFDG_EHIP;
  }
  if (image0) *image0 = L->top[0].mutable_gpu_data();
  if (image1) *image1 = L->top[1].mutable_gpu_data();
  if (flow) *flow = L->top[2].mutable_gpu_data();
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = L->top[0].shape()[i];
  return OFDG_OK;
}

// The background preparation record of getRandomizedCrop(2W, 2H, angle, zoom, shift) on a pool_w x pool_h image
// (host logic, no GPU): f[8] = ca, sa, w2, h2, rw2, rh2, fx, fy; i[6] = x0, y0, cw, ch, shift_x, shift_y.
int ofdg_host_bg_prep(int pool_w, int pool_h, int width, int height, float angle, float zoom, int shift_x, int shift_y, float* f,
                      int* i) {
  if (!f || !i || pool_w < 2 * width || pool_h < 2 * height || !(zoom > 0)) return OFDG_EINVAL;
  const ofdg::DevBgPrep p = ofdg::make_bg_prep(pool_w, pool_h, width, height, angle, zoom, shift_x, shift_y, 0);
  f[0] = p.ca; f[1] = p.sa; f[2] = p.w2; f[3] = p.h2; f[4] = p.rw2; f[5] = p.rh2; f[6] = p.fx; f[7] = p.fy;
  i[0] = p.x0; i[1] = p.y0; i[2] = p.cw; i[3] = p.ch; i[4] = p.shx; i[5] = p.shy;
  return OFDG_OK;
}

}  // extern "C"
