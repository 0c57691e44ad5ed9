// C ABI of libdrin_hip.so (include/drin_hip.h): argument validation, workspace layout and the launch
// sequence of Model.forward (drin/model.py:164-209).  Host code only; kernels live in the sibling files.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <stdio.h>
#include <string.h>

#include "fused.h"
#include "internal.h"
#include "layout.h"

namespace drin {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return DRIN_E_HIP;
}

// ---- the device of a call (internal.h: DeviceScope) -----------------------------------------------
static thread_local int g_call_device = -1;

int call_device() {
  if (g_call_device >= 0) return g_call_device;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return dev;
}

DeviceScope::DeviceScope(void* stream, const void* device_pointer, const char* entry_point)
    : status(DRIN_OK), prev(-1), dev(-1), outer(g_call_device), switched(false) {
  hipError_t e = hipGetDevice(&prev);
  if (e != hipSuccess) {
    // no usable device on this host: nothing to bind; argument validation still answers, and a launch would report the HIP error
    (void)hipGetLastError();
    prev = -1;
    return;
  }
  dev = prev;
  if (stream != nullptr) {
    hipDevice_t d = 0;
    e = hipStreamGetDevice((hipStream_t)stream, &d);
    if (e != hipSuccess) {
      set_error("%s: hipStreamGetDevice: %s (is `stream` a hipStream_t?)", entry_point, hipGetErrorString(e));
      status = DRIN_E_HIP;
      return;
    }
    dev = (int)d;
  } else if (device_pointer != nullptr) {
    // the NULL stream is "the default stream of the current device": take the device that owns the call's memory instead, so
    // that a NULL-stream call on another device's tensors runs on that device's default stream
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, device_pointer) == hipSuccess && attr.type == hipMemoryTypeDevice)
      dev = attr.device;
    else
      (void)hipGetLastError();   // an unregistered pointer: keep the current device (validation will say what is wrong)
  }
  if (dev != prev) {
    e = hipSetDevice(dev);
    if (e != hipSuccess) {
      set_error("%s: hipSetDevice(%d): %s", entry_point, dev, hipGetErrorString(e));
      status = DRIN_E_HIP;
      return;
    }
    switched = true;
  }
  g_call_device = dev;
}

DeviceScope::~DeviceScope() {
  if (status != DRIN_OK || prev < 0) return;
  g_call_device = outer;
  if (switched) (void)hipSetDevice(prev);
}

// ---- roctx ranges (opt-in: DRIN_ROCTX) ----------------------------------------------------------
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (!getenv("DRIN_ROCTX")) return;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
static const Roctx& roctx() {
  static const Roctx r;  // thread-safe one-time initialisation
  return r;
}
RoctxRange::RoctxRange(const char* name) : pushed(false) {
  const Roctx& r = roctx();
  if (r.push) {
    r.push(name);
    pushed = true;
  }
}
RoctxRange::~RoctxRange() {
  if (pushed) roctx().pop();
}

// ---- process-wide kernel profile ---------------------------------------------------------------
// One profile at a time per process; launches from ANY thread are attributed while it is open (autograd
// runs drin_backward on its own thread).  Slots are claimed with an atomic counter; begin / end are
// serialised by a mutex and must not race with launches they are meant to bracket.
struct Profile {
  std::atomic<bool> open{false};
  int capacity = 0;
  std::atomic<int> used{0};
  hipEvent_t* start = nullptr;
  hipEvent_t* stop = nullptr;
  int* cls = nullptr;
};
static Profile g_prof;
static std::mutex g_prof_mutex;

KernelTimer::KernelTimer(int kernel_class, hipStream_t st) : slot(-1), stream(st) {
  Profile& p = g_prof;
  if (!p.open.load(std::memory_order_acquire)) return;
  const int s = p.used.fetch_add(1, std::memory_order_relaxed);
  if (s >= p.capacity) return;
  slot = s;
  p.cls[slot] = kernel_class;
  (void)hipEventRecord(p.start[slot], st);
}

KernelTimer::~KernelTimer() {
  if (slot >= 0) (void)hipEventRecord(g_prof.stop[slot], stream);
}

static void profile_free(Profile& p) {
  for (int i = 0; i < p.capacity; ++i) {
    if (p.start) (void)hipEventDestroy(p.start[i]);
    if (p.stop) (void)hipEventDestroy(p.stop[i]);
  }
  delete[] p.start;
  delete[] p.stop;
  delete[] p.cls;
  p.start = p.stop = nullptr;
  p.cls = nullptr;
  p.capacity = 0;
  p.used.store(0);
  p.open.store(false);
}

int vertex_act(const drin_config* c) { return c->vertex_activation == DRIN_ACT_DEFAULT ? DRIN_ACT_GELU : c->vertex_activation; }
int edge_act(const drin_config* c) { return c->edge_activation == DRIN_ACT_DEFAULT ? DRIN_ACT_SIGMOID : c->edge_activation; }

int validate_config(const drin_config* c) {
  if (!c) {
    set_error("config is NULL");
    return DRIN_E_NULL;
  }
  if (c->batch < 0 || c->num_candidates <= 0 || c->embed_dim <= 0 || c->image_dim <= 0 || c->mention_tokens <= 0 ||
      c->image_regions <= 0 || c->mention_objects < 0 || c->entity_objects < 0 || c->entity_tokens < 0 ||
      c->mention_object_inner < 0 || c->entity_image_inner < 0 || c->entity_object_inner < 0) {
    set_error("config: negative or zero dimension");
    return DRIN_E_SHAPE;
  }
  if (c->embed_dim % 4 || c->image_dim % 4) {
    set_error("config: embed_dim=%d and image_dim=%d must be multiples of 4 (16-byte lane accesses)", c->embed_dim,
              c->image_dim);
    return DRIN_E_SHAPE;
  }
  if (c->embed_dim > 1024) {
    set_error("config: embed_dim=%d > 1024 is not built (LayerNorm row kept in registers)", c->embed_dim);
    return DRIN_E_UNSUPPORTED;
  }
  if (c->vector_edges && (c->embed_dim % 8)) {
    set_error("config: vector edges need embed_dim %% 8 == 0 (got %d)", c->embed_dim);
    return DRIN_E_SHAPE;
  }
  if (c->num_layers < 0 || c->num_layers > DRIN_MAX_LAYERS) {
    set_error("config: num_layers=%d outside [0, %d]", c->num_layers, DRIN_MAX_LAYERS);
    return DRIN_E_SHAPE;
  }
  if ((int64_t)c->batch * c->num_candidates > (int64_t)1 << 30) {
    set_error("config: batch * num_candidates too large for one call; split the batch");
    return DRIN_E_SHAPE;
  }
  if (c->feature_dtype != DRIN_FEAT_F32 && c->feature_dtype != DRIN_FEAT_BF16) {
    set_error("config: feature_dtype %d is not DRIN_FEAT_F32 / DRIN_FEAT_BF16", c->feature_dtype);
    return DRIN_E_UNSUPPORTED;
  }
  if (c->vertex_activation < DRIN_ACT_DEFAULT || c->vertex_activation > DRIN_ACT_SILU) {
    set_error("config: vertex_activation %d is not a drin_activation", c->vertex_activation);
    return DRIN_E_UNSUPPORTED;
  }
  // (sigmoid, tanh, relu: the backward takes the derivative from the stored edge; gelu, silu: the forward keeps the
  //  pre-activation for it, Layout::edge_z)
  if (c->edge_activation < DRIN_ACT_DEFAULT || c->edge_activation > DRIN_ACT_SILU) {
    set_error("config: edge_activation %d is not a drin_activation", c->edge_activation);
    return DRIN_E_UNSUPPORTED;
  }
  if (c->cache_format != DRIN_CACHE_F32 && c->cache_format != DRIN_CACHE_MIXED_F16) {
    set_error("config: cache_format %d is not a drin_cache_format", c->cache_format);
    return DRIN_E_UNSUPPORTED;
  }
  if (c->precision != DRIN_PREC_F32 && c->precision != DRIN_PREC_BF16X3 && c->precision != DRIN_PREC_BF16X3_ALL &&
      c->precision != DRIN_PREC_BF16X3_IF16) {
    set_error("config: precision %d is not a drin_precision", c->precision);
    return DRIN_E_UNSUPPORTED;
  }
  return DRIN_OK;
}

static int validate_batch(const drin_config* c, const drin_batch* b) {
  if (!b) {
    set_error("batch is NULL");
    return DRIN_E_NULL;
  }
  if (c->feature_dtype != DRIN_FEAT_F32) {
    set_error("bf16 feature storage is read by drin_forward_prepared only; widen the features to fp32 for this entry point");
    return DRIN_E_UNSUPPORTED;
  }
  if (c->precision == DRIN_PREC_BF16X3_IF16) {
    set_error("DRIN_PREC_BF16X3_IF16 (the entity-image contraction in one fp16 pass) is an inference mode of drin_forward_prepared; use DRIN_PREC_BF16X3 here");
    return DRIN_E_UNSUPPORTED;
  }
  const void* req[] = {b->mention_text,  b->mention_start,        b->mention_end,     b->mention_image,
                       b->mention_object, b->mention_object_score, b->entity_text,     b->entity_image,
                       b->entity_object,  b->entity_object_score,  b->miet_similarity, b->mtei_similarity};
  const char* names[] = {"mention_text",   "mention_start",        "mention_end",     "mention_image",
                         "mention_object", "mention_object_score", "entity_text",     "entity_image",
                         "entity_object",  "entity_object_score",  "miet_similarity", "mtei_similarity"};
  for (int i = 0; i < 12; ++i) {
    if (!req[i]) {
      set_error("batch.%s is NULL", names[i]);
      return DRIN_E_NULL;
    }
  }
  if (c->entity_tokens > 0 && !b->entity_text_mask) {
    set_error("batch.entity_text_mask is NULL but entity_tokens=%d", c->entity_tokens);
    return DRIN_E_NULL;
  }
  if (b->entity_text_cls && c->entity_tokens > 0) {
    set_error("batch.entity_text_cls goes with pooled entity text (entity_tokens = 0), not with a token block");
    return DRIN_E_UNSUPPORTED;
  }
  const void* al[] = {b->mention_text, b->mention_image, b->mention_object, b->entity_text, b->entity_image,
                      b->entity_object, b->entity_text_cls};
  for (const void* p : al)
    if (!aligned16(p)) {
      set_error("batch: feature tensors must be 16-byte aligned");
      return DRIN_E_ALIGN;
    }
  return DRIN_OK;
}

static int validate_params(const drin_config* c, const drin_params* p) {
  if (!p) {
    set_error("params is NULL");
    return DRIN_E_NULL;
  }
  const void* req[] = {p->w_mention_text,  p->b_mention_text,  p->w_entity_text,  p->b_entity_text,
                       p->w_mention_image, p->b_mention_image, p->w_entity_image, p->b_entity_image};
  for (const void* q : req)
    if (!q || !aligned16(q)) {
      set_error("params: vertex encoder tensor NULL or not 16-byte aligned");
      return q ? DRIN_E_ALIGN : DRIN_E_NULL;
    }
  for (int l = 0; l < c->num_layers; ++l) {
    const drin_layer_params& L = p->layer[l];
    const void* lr[] = {L.w_h, L.b_h, L.w_u, L.b_u, L.w_v, L.b_v, L.ln_weight, L.ln_bias};
    for (const void* q : lr)
      if (!q || !aligned16(q)) {
        set_error("params: layer %d tensor NULL or not 16-byte aligned", l);
        return q ? DRIN_E_ALIGN : DRIN_E_NULL;
      }
    if (c->vector_edges && (!L.w_m || !L.b_m || !aligned16(L.w_m) || !aligned16(L.b_m))) {
      set_error("params: layer %d w_m / b_m (vector edges, model.py:112) NULL or not 16-byte aligned", l);
      return DRIN_E_NULL;
    }
  }
  return DRIN_OK;
}

// Operand pointers of the vertex / edge encoders: pooled copies in the workspace where the reference
// averages an axis, the caller's tensors where it does not.
void resolve_pooled(const drin_config* c, const drin_batch* b, const Layout& L, float* ws, Pooled* out) {
  const int D = c->embed_dim;
  out->span_mean = ws + L.span_mean;
  out->mention_image = ws + L.mimg_pool;
  out->mention_object = c->mention_object_inner > 1 ? ws + L.mobj_pool : b->mention_object;
  out->entity_object = c->entity_object_inner > 1 ? ws + L.eobj_pool : b->entity_object;
  out->entity_image = c->entity_image_inner > 1 ? ws + L.eimg_pool : b->entity_image;
  out->entity_text = c->entity_tokens > 0 ? ws + L.xet_pool : b->entity_text;
  out->entity_text_raw_stride = c->entity_tokens > 0 ? (int64_t)c->entity_tokens * D : D;
}

// Resolves the pooled / raw operand pointers of the vertex and edge encoders and runs the pooling.
int run_pooling(const drin_config* c, const drin_batch* b, const Layout& L, float* ws, Pooled* out, hipStream_t st) {
  const int B = c->batch, N = c->num_candidates, D = c->embed_dim, R = c->image_dim;
  const int64_t M = (int64_t)B * N;
  out->span_mean = ws + L.span_mean;
  DRIN_TRY(launch_span_mean(b->mention_text, b->mention_start, b->mention_end, ws + L.span_mean, B, c->mention_tokens,
                            D, st));
  out->mention_image = ws + L.mimg_pool;
  DRIN_TRY(launch_axis_mean(b->mention_image, ws + L.mimg_pool, B, c->image_regions, R, st));
  if (c->mention_object_inner > 1) {
    DRIN_TRY(launch_axis_mean(b->mention_object, ws + L.mobj_pool, (int64_t)B * c->mention_objects,
                              c->mention_object_inner, R, st));
    out->mention_object = ws + L.mobj_pool;
  } else {
    out->mention_object = b->mention_object;
  }
  if (c->entity_object_inner > 1) {
    DRIN_TRY(launch_axis_mean(b->entity_object, ws + L.eobj_pool, M * c->entity_objects, c->entity_object_inner, R, st));
    out->entity_object = ws + L.eobj_pool;
  } else {
    out->entity_object = b->entity_object;
  }
  if (c->entity_image_inner > 1) {
    DRIN_TRY(launch_axis_mean(b->entity_image, ws + L.eimg_pool, M, c->entity_image_inner, R, st));
    out->entity_image = ws + L.eimg_pool;
  } else {
    out->entity_image = b->entity_image;
  }
  if (c->entity_tokens > 0) {
    DRIN_TRY(launch_entity_token_mean(b->entity_text, b->entity_text_mask, ws + L.xet_pool, M, c->entity_tokens, D, st));
    out->entity_text = ws + L.xet_pool;
    out->entity_text_raw_stride = (int64_t)c->entity_tokens * D;  // token 0 = CLS (model.py:73-75)
  } else {
    out->entity_text = b->entity_text;
    out->entity_text_raw_stride = D;
  }
  return DRIN_OK;
}

int run_static_edges(const drin_config* c, const drin_batch* b, const Pooled& P, float* edges, hipStream_t st) {
  const int B = c->batch, N = c->num_candidates;
  const int64_t M = (int64_t)B * N;
  // tt (model.py:71-76), ti = mtei / 100, it = miet / 100 (model.py:203), ii (model.py:78-92)
  // (entity_text_cls: the raw rows of a batch whose token means were pooled ahead of time)
  const float* raw = b->entity_text_cls ? b->entity_text_cls : b->entity_text;
  const int64_t raw_stride = b->entity_text_cls ? (int64_t)c->embed_dim : P.entity_text_raw_stride;
  // (the two CLIP edges are two loads and two divisions per pair: they ride in the cosine launch)
  DRIN_TRY(launch_cosine_rows(P.span_mean, raw, raw_stride, edges + 0 * M, B, N, c->embed_dim, c->cosine_eps, 1.0f, st,
                              b->entity_index, b->mtei_similarity, b->miet_similarity, edges + 1 * M, edges + 2 * M,
                              c->clip_scale));
  DRIN_TRY(launch_miei(P.mention_object, b->mention_object_score, P.entity_object, b->entity_object_score,
                       edges + 3 * M, B, N, c->mention_objects, c->entity_objects, c->image_dim, c->cosine_eps,
                       c->miei_eps, 1.0f, st, b->entity_index));
  return DRIN_OK;
}

static int copy_out(float* dst, const float* src, size_t n, hipStream_t st) {
  if (!dst) return DRIN_OK;
  hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(trace)");
  return DRIN_OK;
}

static int tap(const drin_trace* t, int l, const Layout& L, const float* ws, int B, int64_t M, int D, hipStream_t st,
               size_t edge_width = 1) {
  if (!t) return DRIN_OK;
  DRIN_TRY(copy_out(t->mention_text_vertex[l], ws + L.vm[l], (size_t)B * D, st));
  DRIN_TRY(copy_out(t->mention_image_vertex[l], ws + L.vm[l] + (size_t)B * D, (size_t)B * D, st));
  DRIN_TRY(copy_out(t->entity_text_vertex[l], ws + L.ve[l], (size_t)M * D, st));
  DRIN_TRY(copy_out(t->entity_image_vertex[l], ws + L.ve[l] + (size_t)M * D, (size_t)M * D, st));
  DRIN_TRY(copy_out(t->edges[l], ws + L.edges[l], (size_t)4 * M * edge_width, st));
  return DRIN_OK;
}

// Table form of the layer-by-layer entry points (training over device-resident entity tables): the entity tensors of the
// batch are TABLES of cfg.num_entities rows and pair p reads row entity_index[p] - inside the static-edge kernels and in
// the row addressing of the vertex-encoder GEMMs (forward x W^T, backward dY^T x), never as gathered copies.
static int indexed_supported(const drin_config* c, const drin_batch* b, const char* who) {
  const int64_t M = (int64_t)c->batch * c->num_candidates;
  const int D = c->embed_dim, R = c->image_dim;
  const bool x3 = c->precision == DRIN_PREC_BF16X3 || c->precision == DRIN_PREC_BF16X3_ALL;
  if (c->num_entities <= 0 || c->entity_tokens != 0 || c->entity_image_inner > 1 || c->entity_object_inner > 1 ||
      c->entity_objects != 1 || c->vector_edges || !x3 || M < 1024 || (D % 32) || (R % 32) || D < 128 || R < 128 ||
      R > 2048 || c->batch > 65535) {
    set_error("%s: entity_index needs pooled entity text tables (entity_tokens = 0, num_entities > 0), one object per entity, "
              "inner dims <= 1, scalar edges, split-bf16 precision, D and R multiples of 32 (>= 128, R <= 2048) and at least "
              "1024 pairs; gather on the caller side otherwise", who);
    return DRIN_E_UNSUPPORTED;
  }
  (void)b;
  return DRIN_OK;
}

}  // namespace drin

using namespace drin;

extern "C" {

int drin_version(void) { return DRIN_ABI_VERSION; }

const char* drin_last_error(void) { return g_err; }

const char* drin_build_info(void) { return "gfx950 hipcc " __VERSION__ " " __DATE__; }

int drin_default_config(drin_config* cfg) {
  if (!cfg) {
    set_error("config is NULL");
    return DRIN_E_NULL;
  }
  memset(cfg, 0, sizeof(*cfg));
  cfg->batch = 0;
  cfg->num_candidates = 11;
  cfg->embed_dim = 768;
  cfg->image_dim = 2048;
  cfg->mention_tokens = 128;
  cfg->image_regions = 49;
  cfg->mention_objects = 3;
  cfg->entity_objects = 1;
  cfg->entity_tokens = 0;
  cfg->mention_object_inner = 1;
  cfg->entity_image_inner = 0;
  cfg->entity_object_inner = 0;
  cfg->num_layers = 2;
  cfg->dynamic_edges = 1;
  for (int k = 0; k < 4; ++k) cfg->edge_enabled[k] = 1.0f;
  cfg->layer_norm_eps = 1e-5f;
  cfg->cosine_eps = 1e-8f;
  cfg->miei_eps = 1e-9f;
  cfg->clip_scale = 100.0f;
  cfg->precision = DRIN_PREC_F32;
  return DRIN_OK;
}

size_t drin_workspace_bytes(const drin_config* cfg, int for_training) {
  if (validate_config(cfg) != DRIN_OK) return 0;
  Layout L;
  L.build(*cfg, for_training != 0);
  return L.total_floats * sizeof(float);
}

int drin_edges_fwd(const drin_config* cfg, const drin_batch* batch, float* edges, float* span_mean, void* stream) {
  DRIN_BIND_DEVICE(stream, edges, "drin_edges_fwd");
  DRIN_TRY(validate_config(cfg));
  DRIN_TRY(validate_batch(cfg, batch));
  if (!edges) {
    set_error("edges is NULL");
    return DRIN_E_NULL;
  }
  if (cfg->mention_object_inner > 1 || cfg->entity_object_inner > 1) {
    set_error("drin_edges_fwd: inner object dims > 1 need the pooled path of drin_forward");
    return DRIN_E_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const int B = cfg->batch, D = cfg->embed_dim;
  if (!span_mean) {
    set_error("drin_edges_fwd: span_mean scratch [B, D] is required");
    return DRIN_E_NULL;
  }
  DRIN_TRY(launch_span_mean(batch->mention_text, batch->mention_start, batch->mention_end, span_mean, B,
                            cfg->mention_tokens, D, st));
  Pooled P;
  P.span_mean = span_mean;
  P.mention_object = batch->mention_object;
  P.entity_object = batch->entity_object;
  P.entity_text_raw_stride = cfg->entity_tokens > 0 ? (int64_t)cfg->entity_tokens * D : D;
  return run_static_edges(cfg, batch, P, edges, st);
}

int drin_pool_fwd(const drin_config* cfg, const drin_batch* batch, float* pooled_entity_text,
                  float* pooled_mention_image, float* pooled_entity_image, void* stream) {
  DRIN_BIND_DEVICE(stream, batch ? (const void*)batch->mention_image : nullptr, "drin_pool_fwd");
  DRIN_TRY(validate_config(cfg));
  if (!batch) {
    set_error("batch is NULL");
    return DRIN_E_NULL;
  }
  const bool bf16 = cfg->feature_dtype == DRIN_FEAT_BF16;
  if (bf16 && (pooled_mention_image || pooled_entity_image)) {
    set_error("drin_pool_fwd: with bf16 feature storage only the entity token pooling is built here (the image means of the "
              "fused path are taken inside drin_forward_prepared)");
    return DRIN_E_UNSUPPORTED;
  }
  // only the inputs of the requested outputs are needed (pooling an entity TABLE once passes the text alone)
  if ((pooled_entity_text && (!batch->entity_text || !batch->entity_text_mask)) ||
      (pooled_mention_image && !batch->mention_image) || (pooled_entity_image && !batch->entity_image)) {
    set_error("drin_pool_fwd: a requested output's input tensor is NULL");
    return DRIN_E_NULL;
  }
  if ((pooled_entity_text && !aligned16(batch->entity_text)) || (pooled_mention_image && !aligned16(batch->mention_image)) ||
      (pooled_entity_image && !aligned16(batch->entity_image))) {
    set_error("drin_pool_fwd: feature tensors must be 16-byte aligned");
    return DRIN_E_ALIGN;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t M = (int64_t)cfg->batch * cfg->num_candidates;
  if (pooled_entity_text) {
    if (cfg->entity_tokens <= 0) {
      set_error("drin_pool_fwd: entity text is already pooled when entity_tokens == 0");
      return DRIN_E_SHAPE;
    }
    if (bf16)  // tokens stored as bf16: pooled in place (half the bytes of the HBM-bound pass), fp32 sums
      DRIN_TRY(launch_entity_token_mean_bf16(batch->entity_text, batch->entity_text_mask, pooled_entity_text, M,
                                             cfg->entity_tokens, cfg->embed_dim, st));
    else
      DRIN_TRY(launch_entity_token_mean(batch->entity_text, batch->entity_text_mask, pooled_entity_text, M,
                                        cfg->entity_tokens, cfg->embed_dim, st));
  }
  if (pooled_mention_image)
    DRIN_TRY(launch_axis_mean(batch->mention_image, pooled_mention_image, cfg->batch, cfg->image_regions,
                              cfg->image_dim, st));
  if (pooled_entity_image) {
    const int inner = cfg->entity_image_inner > 0 ? cfg->entity_image_inner : 1;
    DRIN_TRY(launch_axis_mean(batch->entity_image, pooled_entity_image, M, inner, cfg->image_dim, st));
  }
  return DRIN_OK;
}

int drin_linear_fwd(const float* x, const float* w, const float* bias, float* y, int64_t rows, int32_t n_out, int32_t k,
                    int32_t precision, void* stream) {
  DRIN_BIND_DEVICE(stream, y, "drin_linear_fwd");
  if (!x || !w || !y) {
    set_error("drin_linear_fwd: NULL operand");
    return DRIN_E_NULL;
  }
  if (rows < 0 || n_out <= 0 || k <= 0) {
    set_error("drin_linear_fwd: bad shape rows=%lld n_out=%d k=%d", (long long)rows, n_out, k);
    return DRIN_E_SHAPE;
  }
  return launch_gemm_nt(x, k, w, k, bias, y, n_out, rows, n_out, k, false, precision, (hipStream_t)stream);
}

int drin_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int64_t rows,
                    int32_t n_out, int32_t k, int32_t precision, float* scratch, size_t scratch_floats, void* stream) {
  DRIN_BIND_DEVICE(stream, dy, "drin_linear_bwd");
  if (!dy || (dx && !w) || (dw && !x)) {
    set_error("drin_linear_bwd: NULL operand");
    return DRIN_E_NULL;
  }
  if (rows < 0 || n_out <= 0 || k <= 0) {
    set_error("drin_linear_bwd: bad shape rows=%lld n_out=%d k=%d", (long long)rows, n_out, k);
    return DRIN_E_SHAPE;
  }
  if (rows == 0) return DRIN_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool x3 = precision == DRIN_PREC_BF16X3 || precision == DRIN_PREC_BF16X3_ALL;
  if (dx) {
    // the contraction index of dx = dy W is n: the NT kernel needs W^T [k][n_out]
    if (x3 && scratch && scratch_floats >= (size_t)n_out * k && rows >= 1024 && (n_out % 32) == 0 && (k % 4) == 0) {
      DRIN_TRY(launch_transpose(w, scratch, n_out, k, st));
      // what the transposed weight leaves of the scratch lets a partly filled last round of tiles split along K
      const size_t used = ((size_t)n_out * k + 63) & ~(size_t)63;
      float* tail = scratch_floats > used ? scratch + used : nullptr;
      DRIN_TRY(launch_gemm_nt_bf16x3(dy, n_out, scratch, n_out, nullptr, dx, k, rows, k, n_out, st, nullptr, nullptr, false,
                                     tail, tail ? scratch_floats - used : 0));
    } else {
      DRIN_TRY(launch_gemm_nn(dy, n_out, w, k, dx, k, rows, k, n_out, false, precision, st));
    }
  }
  // (stream order makes the scratch reusable product after product; every split reduction goes through it and is added in
  //  order - without scratch one workgroup per output tile / per 256 columns walks the whole reduction: no atomics either way)
  if (dw) DRIN_TRY(launch_gemm_tn(dy, n_out, x, k, dw, k, rows, n_out, k, precision, st, scratch, scratch ? scratch_floats : 0));
  if (db) DRIN_TRY(launch_colsum(dy, db, rows, n_out, st, scratch, scratch ? scratch_floats : 0));
  return DRIN_OK;
}

int drin_forward(const drin_config* cfg, const drin_batch* batch, const drin_params* params, void* workspace,
                 size_t workspace_bytes, float* scores, int keep_for_backward, const drin_trace* trace, void* stream) {
  return drin_forward_staged(cfg, batch, params, workspace, workspace_bytes, scores, keep_for_backward, trace, nullptr, stream);
}

int drin_forward_staged(const drin_config* cfg, const drin_batch* batch, const drin_params* params, void* workspace,
                        size_t workspace_bytes, float* scores, int keep_for_backward, const drin_trace* trace,
                        void* params_ready_event, void* stream) {
  DRIN_BIND_DEVICE(stream, workspace, "drin_forward_staged");
  RoctxRange range("drin_forward");
  DRIN_TRY(validate_config(cfg));
  DRIN_TRY(validate_batch(cfg, batch));
  DRIN_TRY(validate_params(cfg, params));
  const int64_t* eidx = batch->entity_index;
  if (eidx) DRIN_TRY(indexed_supported(cfg, batch, "drin_forward"));
  if (!scores) {
    set_error("scores is NULL");
    return DRIN_E_NULL;
  }
  Layout L;
  L.build(*cfg, keep_for_backward != 0);
  if (!workspace || !aligned16(workspace)) {
    set_error("workspace is NULL or not 16-byte aligned");
    return workspace ? DRIN_E_ALIGN : DRIN_E_NULL;
  }
  if (workspace_bytes < L.total_floats * sizeof(float)) {
    set_error("workspace has %zu bytes, needs %zu", workspace_bytes, L.total_floats * sizeof(float));
    return DRIN_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int B = cfg->batch, N = cfg->num_candidates, D = cfg->embed_dim, R = cfg->image_dim;
  const int64_t M = (int64_t)B * N;
  const int nl = cfg->num_layers;
  const int prec = cfg->precision;
  const int act_v = vertex_act(cfg), act_e = edge_act(cfg);
  if (B == 0) return DRIN_OK;
  // dead work of the last layer (its edge update and its image-vertex updates never reach the score,
  // SURVEY.md 3.2) is only computed when a trace asks to see it
  const bool full = trace != nullptr;

  Pooled P;
  DRIN_TRY(run_pooling(cfg, batch, L, ws, &P, st));
  const bool vec = cfg->vector_edges != 0;
  const size_t EW = vec ? (size_t)D : 1;   // floats per edge per pair
  const size_t ES = (size_t)M * EW;        // stride between the four edge types
  if (vec) {  // model.py:202: the scalar edges are broadcast over the feature dimension
    DRIN_TRY(run_static_edges(cfg, batch, P, ws + L.edges_scalar, st));
    DRIN_TRY(launch_expand_edges(ws + L.edges_scalar, ws + L.edges[0], 4 * M, D, st));
  } else {
    DRIN_TRY(run_static_edges(cfg, batch, P, ws + L.edges[0], st));
  }

  // Everything above - the pooling passes and the static edges - reads the batch alone; from here on the parameters are
  // read.  drin_forward_staged: they are final once the caller's event has fired (the previous step's gradient all-reduce
  // and optimiser update on another stream, which the parameter-free head above has just run under).
  if (params_ready_event != nullptr) {
    hipError_t we = hipStreamWaitEvent(st, (hipEvent_t)params_ready_event, 0);
    if (we != hipSuccess) return hip_fail(we, "hipStreamWaitEvent(params_ready)");
  }
  // VertexEncoder (model.py:26-46): four Linears
  float* vm0 = ws + L.vm[0];
  float* ve0 = ws + L.ve[0];
  float* const sk = L.splitk_floats ? ws + L.splitk : nullptr;  // split-K scratch of the mention-sized products
  const size_t skf = L.splitk_floats;
  // tail-split scratch of the pair-sized split-bf16 products (training layouts only: the weight-gradient partial
  // buffer is idle during the forward pass and between the weight-gradient products of the backward pass)
  float* const tl = L.tn_part_floats ? ws + L.tn_part : nullptr;
  const size_t tlf = L.tn_part_floats;
  // mention-sized products beyond the exact-fp32 split-K range (2 B > 512 rows) run split-bf16 on small tiles and split K
  // into the same scratch
  float* const msk = sk ? sk : tl;
  const size_t mskf = sk ? skf : tlf;
  // split-bf16 precision: every weight an NT product runs against is split into bf16 (hi, lo) planes ONCE per call (one
  // batched launch) and then streamed by LDS-DMA, instead of being split by every workgroup in every K-block (the register
  // stager of the 64 x 128 tile spent as many VALU cycles on it as the tile has MFMA cycles)
  auto wp = [&](size_t off) -> const float* { return L.weight_planes ? ws + off : nullptr; };
  if (L.weight_planes) {
    SplitBatch sb;
    DRIN_TRY(sb.add(params->w_mention_text, ws + L.wp_enc[0], (int64_t)D * D));
    DRIN_TRY(sb.add(params->w_mention_image, ws + L.wp_enc[1], (int64_t)D * R));
    DRIN_TRY(sb.add(params->w_entity_text, ws + L.wp_enc[2], (int64_t)D * D));
    DRIN_TRY(sb.add(params->w_entity_image, ws + L.wp_enc[3], (int64_t)D * R));
    for (int l = 0; l < nl; ++l) {
      DRIN_TRY(sb.add(params->layer[l].w_h, ws + L.wp_h[l], (int64_t)D * D));
      if (cfg->dynamic_edges && (l < nl - 1 || full)) {
        DRIN_TRY(sb.add(params->layer[l].w_u, ws + L.wp_u[l], (int64_t)D * D));
        DRIN_TRY(sb.add(params->layer[l].w_v, ws + L.wp_v[l], (int64_t)D * D));
      }
    }
    DRIN_TRY(launch_split_planes_batch(sb, st));
  }
  auto nt_pair = [&](const NtProduct& a, const NtProduct& b) -> int { return launch_gemm_nt_pair(a, b, prec, st, msk, mskf); };
  DRIN_TRY(nt_pair({P.span_mean, params->w_mention_text, params->b_mention_text, vm0, D, D, D, B, D, D, wp(L.wp_enc[0])},
                   {P.mention_image, params->w_mention_image, params->b_mention_image, vm0 + (size_t)B * D, R, R, D, B, D, R,
                    wp(L.wp_enc[1])}));
  if (eidx) {  // rows of the entity tables, addressed through the candidate index by the GEMM's stager
    const __bf16* pt = reinterpret_cast<const __bf16*>(wp(L.wp_enc[2]));
    const __bf16* pi = reinterpret_cast<const __bf16*>(wp(L.wp_enc[3]));
    DRIN_TRY(launch_gemm_nt_bf16x3(P.entity_text, D, params->w_entity_text, D, params->b_entity_text, ve0, D, M, D, D, st,
                                   pt, pt ? pt + (size_t)D * D : nullptr, false, tl, tlf, eidx));
    DRIN_TRY(launch_gemm_nt_bf16x3(P.entity_image, R, params->w_entity_image, R, params->b_entity_image, ve0 + (size_t)M * D,
                                   D, M, D, R, st, pi, pi ? pi + (size_t)D * R : nullptr, false, tl, tlf, eidx));
  } else {
    DRIN_TRY(launch_gemm_nt(P.entity_text, D, params->w_entity_text, D, params->b_entity_text, ve0, D, M, D, D, false,
                            prec, st, tl, tlf, wp(L.wp_enc[2])));
    DRIN_TRY(launch_gemm_nt(P.entity_image, R, params->w_entity_image, R, params->b_entity_image, ve0 + (size_t)M * D, D,
                            M, D, R, false, prec, st, tl, tlf, wp(L.wp_enc[3])));
  }
  DRIN_TRY(tap(trace, 0, L, ws, B, M, D, st, EW));

  bool all_enabled = true;
  for (int k = 0; k < 4; ++k) all_enabled = all_enabled && cfg->edge_enabled[k] == 1.0f;
  // gelu / silu edges (model.py:118 takes any F.* name): their derivative needs the pre-activation, kept when training
  const bool keep_z = L.training && (act_e == DRIN_ACT_GELU || act_e == DRIN_ACT_SILU);

  for (int l = 0; l < nl; ++l) {
    const drin_layer_params& W = params->layer[l];
    const bool last = l == nl - 1;
    const bool live_image = !last || full;                       // mi / ei updates
    const bool live_edges = cfg->dynamic_edges && (!last || full);
    const float* e = ws + L.edges[l];
    if (!all_enabled) {  // edges = [e * m ...] (model.py:122)
      float* me = ws + L.masked[l];
      for (int k = 0; k < 4; ++k) DRIN_TRY(launch_scale_div(e + k * ES, me + k * ES, (int64_t)ES, cfg->edge_enabled[k], 1.0f, st));
      e = me;
    }
    const float *e_tt = e, *e_ti = e + ES, *e_it = e + 2 * ES, *e_ii = e + 3 * ES;
    const float* mt = ws + L.vm[l];
    const float* mi = mt + (size_t)B * D;
    const float* et = ws + L.ve[l];
    const float* ei = et + (size_t)M * D;
    float* agg_m = ws + L.agg_m[l];
    float* agg_e = ws + L.agg_e[l];
    // neighbour aggregation, vertex_graph of model.py:105
    if (vec) {
      const float inv_n = 1.0f / (float)N;
      DRIN_TRY(launch_mention_reduce_vec(e_tt, et, e_ti, ei, mt, agg_m, B, N, D, inv_n, true, st));
      DRIN_TRY(launch_entity_aggregate_vec(e_tt, mt, e_it, mi, et, agg_e, B, N, D, st));
      if (live_image) {
        DRIN_TRY(launch_mention_reduce_vec(e_it, et, e_ii, ei, mi, agg_m + (size_t)B * D, B, N, D, inv_n, true, st));
        DRIN_TRY(launch_entity_aggregate_vec(e_ti, mt, e_ii, mi, ei, agg_e + (size_t)M * D, B, N, D, st));
      }
    } else {
      if (B <= 65535) {
        DRIN_TRY(launch_layer_aggregate(e, (int64_t)ES, mt, et, agg_m, agg_e, B, N, D, live_image, st));
      } else {
        DRIN_TRY(launch_mention_aggregate(e_tt, et, e_ti, ei, mt, agg_m, B, N, D, st));
        DRIN_TRY(launch_entity_aggregate(e_tt, mt, e_it, mi, et, agg_e, B, N, D, st));
        if (live_image) {
          DRIN_TRY(launch_mention_aggregate(e_it, et, e_ii, ei, mi, agg_m + (size_t)B * D, B, N, D, st));
          DRIN_TRY(launch_entity_aggregate(e_ti, mt, e_ii, mi, ei, agg_e + (size_t)M * D, B, N, D, st));
        }
      }
    }
    const int types = live_image ? 2 : 1;
    // shared W_h + LayerNorm + GELU for all vertex types of the layer (model.py:128)
    float* h_m = ws + L.h_m[l];
    float* h_e = ws + L.h_e[l];
    const bool scalar_update = live_edges && !vec;   // then W_u(mt, mi) is due too and shares the launch of W_h(agg_m)
    if (scalar_update) {
      DRIN_TRY(nt_pair({agg_m, W.w_h, W.b_h, h_m, D, D, D, (int64_t)types * B, D, D, wp(L.wp_h[l])},
                       {mt, W.w_u, W.b_u, ws + L.fu[l], D, D, D, 2 * (int64_t)B, D, D, wp(L.wp_u[l])}));
    } else {
      DRIN_TRY(launch_gemm_nt(agg_m, D, W.w_h, D, W.b_h, h_m, D, (int64_t)types * B, D, D, false, prec, st, msk, mskf, wp(L.wp_h[l])));
    }
    DRIN_TRY(launch_gemm_nt(agg_e, D, W.w_h, D, W.b_h, h_e, D, (int64_t)types * M, D, D, false, prec, st, tl, tlf, wp(L.wp_h[l])));
    float* st_m = L.training ? ws + L.ln_stat_m[l] : nullptr;
    float* st_e = L.training ? ws + L.ln_stat_e[l] : nullptr;
    // one LayerNorm + GELU launch for the mention and the entity vertices (they share it, model.py:128)
    DRIN_TRY(launch_layernorm_gelu2(h_m, ws + L.vm[l + 1], st_m, st_m ? st_m + 2 * (size_t)B : nullptr, (int64_t)types * B, h_e,
                                    ws + L.ve[l + 1], st_e, st_e ? st_e + 2 * (size_t)M : nullptr, (int64_t)types * M,
                                    W.ln_weight, W.ln_bias, D, cfg->layer_norm_eps, st, act_v));
    // dynamic edges, edge_graph of model.py:107: (mt,et) (mt,ei) (mi,et) (mi,ei)
    float* e_next = ws + L.edges[l + 1];
    if (live_edges && vec) {
      // model.py:150-152,133: cat(W_u(u), W_v(v)) + e -> W_m -> sigmoid, W_u / W_v: D -> D/2
      float* fu = ws + L.fu[l];
      float* fv = ws + L.fv[l];
      float* pre = ws + L.pre[l];
      const int H = D / 2;
      DRIN_TRY(launch_gemm_nt(mt, D, W.w_u, D, W.b_u, fu, H, 2 * (int64_t)B, H, D, false, prec, st, msk, mskf));
      DRIN_TRY(launch_gemm_nt(et, D, W.w_v, D, W.b_v, fv, H, 2 * M, H, D, false, prec, st));
      DRIN_TRY(launch_edge_pre_vec(fu, fv, e, pre, B, N, D, st));
      DRIN_TRY(launch_gemm_nt(pre, D, W.w_m, D, W.b_m, e_next, D, 4 * M, D, D, false, prec, st));
      DRIN_TRY(launch_sigmoid_inplace(e_next, 4 * M * D, st, act_e, keep_z ? ws + L.edge_z[l] : nullptr));
    } else if (live_edges) {
      float* fu = ws + L.fu[l];
      float* fv = ws + L.fv[l];
      DRIN_TRY(launch_gemm_nt(et, D, W.w_v, D, W.b_v, fv, D, 2 * M, D, D, false, prec, st, tl, tlf, wp(L.wp_v[l])));   // fu: with W_h above
      DRIN_TRY(launch_edge_update4(fu, fv, e, e_next, B, N, D, st, act_e, keep_z ? ws + L.edge_z[l] : nullptr));
    } else if (!cfg->dynamic_edges) {
      hipError_t err = hipMemcpyAsync(e_next, e, 4 * ES * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (err != hipSuccess) return hip_fail(err, "hipMemcpyAsync(static edges)");
    }
    if (full) DRIN_TRY(tap(trace, l + 1, L, ws, B, M, D, st, EW));
  }
  // score (model.py:207-209)
  return launch_cosine_rows(ws + L.vm[nl], ws + L.ve[nl], D, scores, B, N, D, cfg->cosine_eps, 1.0f, st);
}

int drin_profile_begin(int max_launches) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  Profile& p = g_prof;
  if (p.open.load()) profile_free(p);
  if (max_launches <= 0 || max_launches > (1 << 20)) {
    set_error("drin_profile_begin: max_launches=%d outside (0, 2^20]", max_launches);
    return DRIN_E_SHAPE;
  }
  p.start = new hipEvent_t[max_launches];
  p.stop = new hipEvent_t[max_launches];
  p.cls = new int[max_launches];
  for (int i = 0; i < max_launches; ++i) {
    hipError_t e = hipEventCreate(&p.start[i]);
    if (e == hipSuccess) e = hipEventCreate(&p.stop[i]);
    if (e != hipSuccess) {
      p.capacity = i;
      profile_free(p);
      return hip_fail(e, "hipEventCreate");
    }
  }
  p.capacity = max_launches;
  p.used.store(0);
  p.open.store(true, std::memory_order_release);
  return DRIN_OK;
}

int drin_profile_end(double* ms_by_class, int64_t* launches_by_class) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  Profile& p = g_prof;
  if (!p.open.load()) {
    set_error("drin_profile_end: no profile open");
    return DRIN_E_SHAPE;
  }
  for (int k = 0; k < DRIN_KC_COUNT; ++k) {
    if (ms_by_class) ms_by_class[k] = 0.0;
    if (launches_by_class) launches_by_class[k] = 0;
  }
  int status = DRIN_OK;
  p.open.store(false);  // launches from other threads stop claiming slots from here on
  const int claimed = p.used.load();
  const int used = claimed < p.capacity ? claimed : p.capacity;
  for (int i = 0; i < used; ++i) {
    hipError_t e = hipEventSynchronize(p.stop[i]);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, p.start[i], p.stop[i]);
    if (e != hipSuccess) {
      status = hip_fail(e, "hipEventElapsedTime");
      break;
    }
    const int k = p.cls[i];
    if (k >= 0 && k < DRIN_KC_COUNT) {
      if (ms_by_class) ms_by_class[k] += ms;
      if (launches_by_class) launches_by_class[k] += 1;
    }
  }
  if (status == DRIN_OK && claimed >= p.capacity) {
    set_error("drin_profile_end: profile overflowed its %d launch slots", p.capacity);
    status = DRIN_E_SHAPE;
  }
  profile_free(p);
  return status;
}

const char* drin_kernel_class_name(int kernel_class) {
  static const char* names[DRIN_KC_COUNT] = {"gemm", "pool", "edge", "gcn", "stream", "gemm_x3", "gemm_planes", "optim"};
  return (kernel_class >= 0 && kernel_class < DRIN_KC_COUNT) ? names[kernel_class] : "?";
}

int drin_backward(const drin_config* cfg, const drin_batch* batch, const drin_params* params, void* workspace,
                  size_t workspace_bytes, const float* grad_scores, const drin_param_grads* grads, void* stream) {
  return drin_backward_staged(cfg, batch, params, workspace, workspace_bytes, grad_scores, grads, nullptr, stream);
}

int drin_backward_staged(const drin_config* cfg, const drin_batch* batch, const drin_params* params, void* workspace,
                         size_t workspace_bytes, const float* grad_scores, const drin_param_grads* grads,
                         void* layers_ready_event, void* stream) {
  DRIN_BIND_DEVICE(stream, workspace, "drin_backward_staged");
  RoctxRange range("drin_backward");
  DRIN_TRY(validate_config(cfg));
  DRIN_TRY(validate_batch(cfg, batch));
  DRIN_TRY(validate_params(cfg, params));
  if (!grad_scores || !grads) {
    set_error("drin_backward: grad_scores / grads is NULL");
    return DRIN_E_NULL;
  }
  const int64_t* eidx = batch->entity_index;
  if (eidx) DRIN_TRY(indexed_supported(cfg, batch, "drin_backward"));
  Layout L;
  L.build(*cfg, true);
  if (!workspace || !aligned16(workspace)) {
    set_error("workspace is NULL or not 16-byte aligned");
    return workspace ? DRIN_E_ALIGN : DRIN_E_NULL;
  }
  if (workspace_bytes < L.total_floats * sizeof(float)) {
    set_error("workspace has %zu bytes, needs %zu (was drin_forward run with keep_for_backward?)", workspace_bytes,
              L.total_floats * sizeof(float));
    return DRIN_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int B = cfg->batch, N = cfg->num_candidates, D = cfg->embed_dim, R = cfg->image_dim;
  const size_t M = (size_t)B * N, BD = (size_t)B * D, MD = M * D;
  const int nl = cfg->num_layers;
  const int prec = cfg->precision;
  const int act_v = vertex_act(cfg), act_e = edge_act(cfg);
  if (B == 0) return DRIN_OK;
  Pooled P;
  resolve_pooled(cfg, batch, L, ws, &P);

  // carve the backward scratch (Layout::bwd_scratch): two generations of vertex / edge gradients,
  // W_h-input gradients, edge-update gradients
  float* sp = ws + L.bwd_scratch;
  auto carve = [&sp](size_t n) {
    float* p = sp;
    sp += (n + 63) & ~(size_t)63;
    return p;
  };
  // vertex gradients: one generation per level (g_v?[l + 1] = dL/dH of layer l, g_v?[0] = the vertex encoders' output
  // gradient) - they are operands of the weight-gradient products, which run as grouped launches at the end of the pass
  float* g_vm[DRIN_MAX_LAYERS + 1];
  for (int l = 0; l <= nl; ++l) g_vm[l] = carve(2 * BD);
  float* g_ve[DRIN_MAX_LAYERS + 1];
  for (int l = 0; l <= nl; ++l) g_ve[l] = carve(2 * MD);
  const bool vec = cfg->vector_edges != 0;
  const size_t EW = vec ? (size_t)D : 1, ES = M * EW;  // floats per edge per pair, stride between edge types
  float* g_e[2] = {carve(4 * ES), carve(4 * ES)};
  float* dA_m = carve(2 * BD);
  float* dA_e = carve(2 * MD);
  float* dfu_of[DRIN_MAX_LAYERS];
  for (int l = 0; l < (nl > 1 ? nl - 1 : 1); ++l) dfu_of[l] = carve(2 * BD);
  float* dfv_of[DRIN_MAX_LAYERS];   // per layer, for the same reason (the last layer has no live edge update)
  for (int l = 0; l < (nl > 1 ? nl - 1 : 1); ++l) dfv_of[l] = carve(2 * MD);
  float* dpre = carve(4 * ES);
  float* cos_scratch = carve(3 * M);
  if ((size_t)(sp - (ws + L.bwd_scratch)) > L.bwd_scratch_floats) {
    set_error("internal: backward scratch overflow");
    return DRIN_E_WORKSPACE;
  }

  bool all_enabled = true;
  for (int k = 0; k < 4; ++k) all_enabled = all_enabled && cfg->edge_enabled[k] == 1.0f;
  float* const tnp = L.tn_part_floats ? ws + L.tn_part : nullptr;  // partial tiles of the split-bf16 dW products
  const size_t tnf = L.tn_part_floats;
  // gelu / silu edges: the derivative is taken at the pre-activation the forward kept (Layout::edge_z)
  const bool from_z = act_e == DRIN_ACT_GELU || act_e == DRIN_ACT_SILU;
  const int act_eb = from_z ? (act_e | 0x100) : act_e;   // device_utils.h: kActFromPre
  float* const smp = ws + L.small_part;     // slices of the mention-sized exact-fp32 dW products
  float* const csp = ws + L.colsum_part;    // partial rows of the bias column sums
  // dX (+)= dY W.  Pair-sized products in split-bf16 precision run on the NT kernel against W^T, transposed into
  // workspace scratch right before use (a D x D transpose is ~3 us; the product it feeds is 2.5x faster than
  // the exact-fp32 MFMA one); everything else takes the exact fp32 NN kernel.
  const bool x3 = prec == DRIN_PREC_BF16X3 || prec == DRIN_PREC_BF16X3_ALL;
  // scalar edges: W_h^T and W_v^T of every layer in ONE batched transpose up front (slots 2 l and 2 l + 1 of L.wt);
  // anything else (W_u at 512+ mentions; the half-width W_u / W_v and W_m of vector edges) is transposed per product into the last slot
  const bool pre_t = x3 && !cfg->vector_edges && M >= 1024 && (D % 32) == 0;
  auto wt_slot = [&](int l, int which) { return ws + L.wt + ((size_t)2 * l + which) * D * D; };
  if (pre_t) {  // ... and split into bf16 (hi, lo) planes in the same pass: the NT kernel streams them by LDS-DMA
    SplitBatch tb;
    for (int l = 0; l < nl; ++l) {
      DRIN_TRY(tb.add(params->layer[l].w_h, wt_slot(l, 0), (int64_t)D * D));
      if (cfg->dynamic_edges && l < nl - 1) DRIN_TRY(tb.add(params->layer[l].w_v, wt_slot(l, 1), (int64_t)D * D));
    }
    DRIN_TRY(launch_transpose_split_batch(tb, D, D, st));
  }
  auto gemm_nn = [&](const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx, int64_t rows, int n_out,
                     int k_red, bool accumulate, const float* w_t = nullptr) -> int {
    // w is [k_red][n_out] contiguous; w_t: bf16 (hi, lo) planes of its transpose, already in the workspace
    if (x3 && rows >= 1024 && (k_red % 32) == 0 && (n_out % 4) == 0 && (size_t)k_red * n_out <= (size_t)D * D) {
      if (w_t != nullptr) {
        const __bf16* hi = reinterpret_cast<const __bf16*>(w_t);
        return launch_gemm_nt_bf16x3(dy, lddy, nullptr, k_red, nullptr, dx, lddx, rows, n_out, k_red, st, hi, hi + (size_t)k_red * n_out,
                                     accumulate, tnp, tnf);
      }
      if (w_t == nullptr) {
        float* wt = ws + L.wt + (size_t)2 * nl * D * D;   // the scratch slot: never one of the pre-transposed weights
        DRIN_TRY(launch_transpose(w, wt, k_red, n_out, st));
        w_t = wt;
      }
      return launch_gemm_nt_bf16x3(dy, lddy, w_t, k_red, nullptr, dx, lddx, rows, n_out, k_red, st, nullptr, nullptr, accumulate,
                                   tnp, tnf);
    }
    return launch_gemm_nn(dy, lddy, w, n_out, dx, lddx, rows, n_out, k_red, accumulate, prec, st,
                          L.splitk_floats ? ws + L.splitk : nullptr, L.splitk_floats);
  };
  SliceSum ln_sums;        // second level of the LayerNorm backward's column sums (dgamma, dbeta, db_h of every layer)
  ColsumBatch bias_sums;   // the bias gradients of W_u / W_v and of the vertex encoders: one launch
  // dW (+)= dY^T X.  The pair-sized products of the whole pass (dW_h, dW_v of every layer, the two entity encoders) are
  // collected and run as ONE launch at the end: alone, each deals its ~15-stage slices over the chip between a pipeline
  // fill and a partial tile per workgroup; together the workgroups walk ~4x longer slices.  Mention-sized products
  // (exact fp32) and the vector-edge ones (their operands are overwritten layer by layer) run where they arise.
  TnGroup dw_group;
  F32GemmGroup dw_small;   // the mention-sized ones (exact fp32): one launch as well
  // drin_backward_staged: how many entries of the three collections belong to the GCN layers (-1: still collecting them).
  // An instalment flushed while the vertex encoders' products are being added takes every layer entry with it: 0 remain.
  int layer_sums = -1, layer_small = -1, layer_group = -1;
  const bool defer_dw = x3 && !vec && tnp != nullptr;
  // (db: the bias gradient that goes with it = the column sums of dy; the group takes them from the rows it stages)
  auto dw_product = [&](const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw, int64_t lddw, int64_t rows,
                        int n_out, int k_red, const int64_t* x_index = nullptr, float* db = nullptr) -> int {
    const bool grouped = dw != nullptr && defer_dw && gemm_tn_bf16x3_fits(lddy, ldx, rows, n_out, k_red, dy, x) &&
                         gemm_tn_bf16x3_scratch_ok(dw, lddw, n_out, k_red, tnp, tnf);
    if (db != nullptr && !(grouped && lddy == n_out)) {
      DRIN_TRY(bias_sums.add(dy, db, rows, n_out));
      db = nullptr;
    }
    if (dw == nullptr) return DRIN_OK;
    if (grouped) {
      if (dw_group.n == TnGroup::MAX) {   // deeper than three layers: the group goes in instalments
        DRIN_TRY(launch_gemm_tn_group(dw_group, st, tnp, tnf));
        dw_group = TnGroup();
        if (layer_group >= 0) layer_group = 0;
      }
      return dw_group.add(dy, lddy, x, ldx, dw, lddw, rows, n_out, k_red, x_index, db);
    }
    if (x_index != nullptr) return launch_gemm_tn_bf16x3(dy, lddy, x, ldx, dw, lddw, rows, n_out, k_red, st, tnp, tnf, x_index);
    const bool takes_x3 = x3 && gemm_tn_bf16x3_fits(lddy, ldx, rows, n_out, k_red, dy, x);
    if (!vec && !takes_x3 && rows <= 2048 && (n_out % 4) == 0 && (k_red % 4) == 0 && (prec == DRIN_PREC_F32 || x3) &&
        aligned16(dw) && (lddw % 4) == 0) {
      if (dw_small.n == F32GemmGroup::MAX) {
        DRIN_TRY(launch_gemm_tn_f32_group(dw_small, st, smp, L.small_part_floats));
        dw_small = F32GemmGroup();
        if (layer_small >= 0) layer_small = 0;
      }
      return dw_small.add_tn(dy, lddy, x, ldx, dw, lddw, rows, n_out, k_red);
    }
    return launch_gemm_tn(dy, lddy, x, ldx, dw, lddw, rows, n_out, k_red, prec, st, tnp, tnf);
  };

  // score = cos(mt_L, et_L) (model.py:207-209)
  DRIN_TRY(launch_cosine_bwd(ws + L.vm[nl], ws + L.ve[nl], grad_scores, g_vm[nl], g_ve[nl], cos_scratch, B, N, D,
                             cfg->cosine_eps, st));
  bool have_image = false;  // gradients w.r.t. the image vertices of the current level exist
  bool have_edge = false;   // gradients w.r.t. the current level's edges exist

  for (int l = nl - 1; l >= 0; --l) {
    const drin_layer_params& W = params->layer[l];
    const auto& G = grads->layer[l];
    const int types = have_image ? 2 : 1;
    float* gm = g_vm[l + 1];  // dL/d(new mention vertices) -> dL/dH in place
    float* ge = g_ve[l + 1];
    float* gm_next = g_vm[l];
    float* ge_next = g_ve[l];
    float* dfv = dfv_of[l < nl - 1 ? l : 0];
    float* dfu = dfu_of[l < nl - 1 ? l : 0];
    const int cur = (nl - 1 - l) & 1, nxt = cur ^ 1;  // the edge gradients keep two generations
    const float* st_m = ws + L.ln_stat_m[l];
    const float* st_e = ws + L.ln_stat_e[l];
    // (a) LayerNorm + GELU backward; column sums give dgamma, dbeta and db_h
    // (mention and entity rows in one launch)
    // (its column sums' second level rides in the slice sum at the end of the pass: per-layer level-1 rows behind the block rows)
    DRIN_TRY(launch_layernorm_gelu_bwd2(ws + L.h_m[l], st_m, st_m + 2 * (size_t)B, gm, (int64_t)types * B, ws + L.h_e[l], st_e,
                                        st_e + 2 * M, ge, (int64_t)types * M, W.ln_weight, W.ln_bias, G.ln_weight, G.ln_bias,
                                        G.b_h, ws + L.ln_part, D, st, act_v, nl <= 2 ? &ln_sums : nullptr,
                                        ws + L.ln_part + ((size_t)(1024 + 16) + (size_t)16 * l) * 3 * D));
    // (b) dW_h += dH^T A
    if (G.w_h) {
      DRIN_TRY(dw_product(gm, D, ws + L.agg_m[l], D, G.w_h, D, (int64_t)types * B, D, D));
      DRIN_TRY(dw_product(ge, D, ws + L.agg_e[l], D, G.w_h, D, (int64_t)types * M, D, D));
    }
    // (c) dA = dH W_h
    DRIN_TRY(gemm_nn(gm, D, W.w_h, dA_m, D, (int64_t)types * B, D, D, false));
    DRIN_TRY(gemm_nn(ge, D, W.w_h, dA_e, D, (int64_t)types * M, D, D, false, pre_t ? wt_slot(l, 0) : nullptr));
    const float* dA_mt = dA_m;
    const float* dA_mi = types == 2 ? dA_m + BD : nullptr;
    const float* dA_et = dA_e;
    const float* dA_ei = types == 2 ? dA_e + MD : nullptr;

    const float* e = all_enabled ? ws + L.edges[l] : ws + L.masked[l];
    const float* mt = ws + L.vm[l];
    const float* mi = mt + BD;
    const float* et = ws + L.ve[l];
    const float* ei = et + MD;
    const bool edge_update = cfg->dynamic_edges && have_edge;  // this layer's edge update is live
    const float* de_extra = nullptr;
    if (edge_update && vec) {
      // (d, vector edges) e' = sigmoid(W_m(cat(W_u(u), W_v(v)) + e) + b_m)  (model.py:150-152,133)
      const int H = D / 2;
      DRIN_TRY(launch_sigmoid_bwd(g_e[cur], from_z ? ws + L.edge_z[l] : ws + L.edges[l + 1], dpre, 4 * (int64_t)ES, st, act_eb));  // dz
      if (G.w_m) DRIN_TRY(launch_gemm_tn(dpre, D, ws + L.pre[l], D, G.w_m, D, 4 * (int64_t)M, D, D, prec, st, tnp, tnf));
      DRIN_TRY(launch_colsum(dpre, G.b_m, 4 * (int64_t)M, D, st, csp, L.colsum_part_floats));
      DRIN_TRY(gemm_nn(dpre, D, W.w_m, g_e[cur], D, 4 * (int64_t)M, D, D, false));  // d(cat + e)
      DRIN_TRY(launch_edge_pre_vec_bwd(g_e[cur], dfu, dfv, B, N, D, st));
      if (G.w_v) DRIN_TRY(launch_gemm_tn(dfv, H, et, D, G.w_v, D, 2 * (int64_t)M, H, D, prec, st, tnp, tnf));
      DRIN_TRY(launch_colsum(dfv, G.b_v, 2 * (int64_t)M, H, st, csp, L.colsum_part_floats));
      if (G.w_u) DRIN_TRY(launch_gemm_tn(dfu, H, mt, D, G.w_u, D, 2 * (int64_t)B, H, D, prec, st, tnp, tnf));
      DRIN_TRY(launch_colsum(dfu, G.b_u, 2 * (int64_t)B, H, st, csp, L.colsum_part_floats));
      de_extra = g_e[cur];
    } else if (edge_update) {
      // (d) e'_k = sigmoid(mean_d(fu fv) + e_k)  (model.py:148-153,133)
      const float* fu = ws + L.fu[l];
      const float* fv = ws + L.fv[l];
      const float inv_d = 1.0f / (float)D;
      // dpre = g e' (1 - e');  dfv_t = (dpre_tt fu_t + dpre_it fu_i) / D ; dfv_i = (dpre_ti fu_t + dpre_ii fu_i) / D: one pass
      DRIN_TRY(launch_edge_update_bwd(g_e[cur], from_z ? ws + L.edge_z[l] : ws + L.edges[l + 1], fu, dpre, dfv, B, N, D, inv_d, st, act_eb));
      // dfu_t = sum_n (dpre_tt fv_t + dpre_ti fv_i) / D ; dfu_i = sum_n (dpre_it fv_t + dpre_ii fv_i) / D
      // (B <= 65535: nothing reads dfu before the mention side of the aggregation backward below, (f): the two reductions
      //  share ONE launch there - and dW_u, which reads dfu, is collected behind it)
      if (B > 65535) {
        DRIN_TRY(launch_mention_reduce(dpre, fv, dpre + M, fv + MD, nullptr, dfu, B, N, D, inv_d, st));
        DRIN_TRY(launch_mention_reduce(dpre + 2 * M, fv, dpre + 3 * M, fv + MD, nullptr, dfu + BD, B, N, D, inv_d, st));
        DRIN_TRY(dw_product(dfu, D, mt, D, G.w_u, D, 2 * (int64_t)B, D, D, nullptr, G.b_u));
      }
      DRIN_TRY(dw_product(dfv, D, et, D, G.w_v, D, 2 * (int64_t)M, D, D, nullptr, G.b_v));
      de_extra = dpre;
    } else if (!cfg->dynamic_edges && have_edge) {
      de_extra = g_e[cur];  // static edges pass through (model.py:136)
    }
    if (vec) {
      const int H = D / 2;
      // (e) entity side of the aggregation backward + edge gradients, element-wise with vector edges
      DRIN_TRY(launch_entity_side_bwd_vec(dA_mt, dA_mi, dA_et, dA_ei, mt, mi, et, ei, e, de_extra, ge_next,
                                          ge_next + MD, g_e[nxt], B, N, D, cfg->edge_enabled, st));
      if (edge_update) DRIN_TRY(gemm_nn(dfv, H, W.w_v, ge_next, D, 2 * (int64_t)M, D, H, true));
      // (f) mention side
      DRIN_TRY(launch_mention_reduce_vec(e, dA_et, e + ES, dA_ei, dA_mt, gm_next, B, N, D, 1.0f, false, st));
      DRIN_TRY(launch_mention_reduce_vec(e + 2 * ES, dA_et, e + 3 * ES, dA_ei, dA_mi, gm_next + BD, B, N, D, 1.0f, false, st));
      if (edge_update) DRIN_TRY(gemm_nn(dfu, H, W.w_u, gm_next, D, 2 * (int64_t)B, D, H, true));
    } else {
      // (e) entity side of the aggregation backward + edge gradients
      // (the edge update's dfv W_v goes first and the row kernel adds onto it)
      if (edge_update) DRIN_TRY(gemm_nn(dfv, D, W.w_v, ge_next, D, 2 * (int64_t)M, D, D, false, pre_t ? wt_slot(l, 1) : nullptr));
      DRIN_TRY(launch_entity_side_bwd(dA_mt, dA_mi, dA_et, dA_ei, mt, mi, et, ei, e, de_extra, ge_next, ge_next + MD,
                                      g_e[nxt], B, N, D, cfg->edge_enabled, edge_update, st));
      // (f) mention side
      if (B <= 65535 && edge_update) {
        const float* fv = ws + L.fv[l];
        DRIN_TRY(launch_mention_reduce2_pair(dpre, fv, fv + MD, nullptr, nullptr, dfu, dfu + BD, 1.0f / (float)D,          // (d): dfu
                                             e, dA_et, dA_ei, dA_mt, dA_mi, gm_next, gm_next + BD, 1.0f, B, N, D, st));   // (f)
        DRIN_TRY(dw_product(dfu, D, mt, D, G.w_u, D, 2 * (int64_t)B, D, D, nullptr, G.b_u));
      } else if (B <= 65535) {
        DRIN_TRY(launch_mention_reduce2(e, dA_et, dA_ei, dA_mt, dA_mi, gm_next, gm_next + BD, B, N, D, 1.0f, st));
      } else {
        DRIN_TRY(launch_mention_reduce(e, dA_et, e + M, dA_ei, dA_mt, gm_next, B, N, D, 1.0f, st));
        DRIN_TRY(launch_mention_reduce(e + 2 * M, dA_et, e + 3 * M, dA_ei, dA_mi, gm_next + BD, B, N, D, 1.0f, st));
      }
      if (edge_update) DRIN_TRY(gemm_nn(dfu, D, W.w_u, gm_next, D, 2 * (int64_t)B, D, D, true));
    }
    have_image = true;
    have_edge = true;
    // dfv / dfu are overwritten by the next layer down: their column sums go now - except layer 0's, which share the
    // launch of the vertex encoders' bias gradients below
    if (l > 0 && bias_sums.n > 0) {
      DRIN_TRY(launch_colsum_batch(bias_sums, st, csp, L.colsum_part_floats));
      bias_sums = ColsumBatch();
    }
  }

  // what is collected up to here belongs to the GCN layers, what follows to the vertex encoders (drin_backward_staged)
  layer_sums = bias_sums.n, layer_small = dw_small.n, layer_group = dw_group.n;
  // VertexEncoder (model.py:26-46): four Linears over the pooled inputs
  const float* g_mt = g_vm[0];
  const float* g_mi = g_vm[0] + BD;
  const float* g_et = g_ve[0];
  const float* g_ei = g_ve[0] + MD;
  DRIN_TRY(dw_product(g_mt, D, P.span_mean, D, grads->w_mention_text, D, B, D, D, nullptr, grads->b_mention_text));
  DRIN_TRY(dw_product(g_et, D, P.entity_text, D, grads->w_entity_text, D, (int64_t)M, D, D, eidx, grads->b_entity_text));
  if (have_image) {
    DRIN_TRY(dw_product(g_mi, D, P.mention_image, R, grads->w_mention_image, R, B, D, R, nullptr, grads->b_mention_image));
    DRIN_TRY(dw_product(g_ei, D, P.entity_image, R, grads->w_entity_image, R, (int64_t)M, D, R, eidx, grads->b_entity_image));
  }
  // The split reductions left at the end of the pass - the bias column sums, the mention-sized and the pair-sized weight
  // gradients - each store their slices, and ONE slice-sum launch adds them all to the gradients in a fixed order (two
  // products of one destination, dW_h's mention and entity rows, as two segments of one entry).
  // Staged (layers_ready_event): the same launches in two parts - first everything that lands in a GCN layer's gradients,
  // then the event, then the vertex encoders' part; the scratch regions are reused in stream order.  Each part deals the
  // chip's workgroups over ITS products (handing both parts the whole group's slice length keeps every bit of the one-part
  // launch, but leaves the chip half empty twice: measured 0.73 -> 0.95 ms of split-bf16 GEMM time per B = 64 step), so the
  // pair-sized weight gradients of the staged pass differ from drin_backward's in the last bits - by the summation
  // split only; each is reproducible.
  const int64_t target = 0;
  auto flush = [&](int s0, int s1, int f0, int f1, int g0, int g1, bool with_layernorm) -> int {
    ColsumBatch cs;
    for (int i = s0; i < s1; ++i) {
      cs.x[cs.n] = bias_sums.x[i], cs.out[cs.n] = bias_sums.out[i], cs.rows[cs.n] = bias_sums.rows[i];
      cs.c4[cs.n] = bias_sums.c4[i], cs.by[cs.n] = bias_sums.by[i];
      ++cs.n;
    }
    F32GemmGroup fg;
    for (int i = f0; i < f1; ++i) fg.item[fg.n] = dw_small.item[i], fg.bias_of[fg.n] = nullptr, ++fg.n;
    TnGroup tg;
    for (int i = g0; i < g1; ++i) tg.item[tg.n++] = dw_group.item[i];
    SliceSum sums;
    if (with_layernorm) sums = ln_sums;   // (layer gradients: they belong to the first part of a staged pass)
    DRIN_TRY(launch_colsum_batch(cs, st, csp, L.colsum_part_floats, &sums));
    DRIN_TRY(launch_gemm_tn_f32_group(fg, st, smp, L.small_part_floats, &sums));
    DRIN_TRY(launch_gemm_tn_group(tg, st, tnp, tnf, &sums, target));
    return launch_slice_sum(sums, st);
  };
  if (layers_ready_event == nullptr) return flush(0, bias_sums.n, 0, dw_small.n, 0, dw_group.n, true);
  DRIN_TRY(flush(0, layer_sums, 0, layer_small, 0, layer_group, true));
  hipError_t ev = hipEventRecord((hipEvent_t)layers_ready_event, st);
  if (ev != hipSuccess) return hip_fail(ev, "hipEventRecord(layers_ready)");
  return flush(layer_sums, bias_sums.n, layer_small, dw_small.n, layer_group, dw_group.n, false);
}

// ---- host-side self-checks (sanitizer build / CI; no launch is made, no GPU needed) -----------------------------------
// What the mention-sized exact-fp32 weight-gradient group of drin_backward may have to hold at once, at worst: every
// weight-gradient product of the pass whose reduction has at most 2048 rows (dw_product above: in exact-fp32 precision all
// of them, in split-bf16 precision the ones gemm_tn_bf16x3_fits refuses), each with small_tn_slices(rows) stored slices.
// Layout::small_part_floats must cover it for every batch size (it did not for 1024 < B N <= 2048).
static size_t small_group_worst_case_floats(const drin_config& c) {
  const size_t B = c.batch, M = B * c.num_candidates, D = c.embed_dim, R = c.image_dim;
  const int nl = c.num_layers;
  size_t need = 0;
  auto product = [&need](size_t rows, size_t n_out, size_t k_red) {
    if (rows >= 1 && rows <= 2048) need += (size_t)small_tn_slices((int64_t)rows) * n_out * k_red;
  };
  for (int l = nl - 1; l >= 0; --l) {
    const size_t types = l == nl - 1 ? 1 : 2;       // the top layer's image vertices are dead
    product(types * B, D, D), product(types * M, D, D);                          // dW_h: mention rows, entity rows
    if (c.dynamic_edges && l < nl - 1) product(2 * M, D, D), product(2 * B, D, D);   // dW_v, dW_u
  }
  product(B, D, D), product(M, D, D), product(B, D, R), product(M, D, R);        // the four vertex encoders
  return need;
}

int drin_host_selftest(void) {
  // (1) the slice scratch of the mention-sized weight-gradient group covers its worst case at every batch size
  for (int precision : {(int)DRIN_PREC_F32, (int)DRIN_PREC_BF16X3})
    for (int nl : {1, 2, 3})
      for (int n : {1, 11, 101})
        for (int dims = 0; dims < 2; ++dims)
          for (int b = 1; b <= 4300; b += (b < 48 ? 1 : 7)) {
            drin_config c;
            drin_default_config(&c);
            c.batch = b, c.num_candidates = n, c.num_layers = nl, c.precision = precision;
            if (dims) c.embed_dim = 64, c.image_dim = 128;
            Layout L;
            L.build(c, true);
            const size_t need = small_group_worst_case_floats(c);
            if (need > L.small_part_floats) {
              set_error("selftest: B=%d N=%d layers=%d D=%d precision=%d: the mention-sized weight-gradient group may store %zu "
                        "floats of slices, Layout::small_part_floats = %zu", b, n, nl, c.embed_dim, precision, need, L.small_part_floats);
              return DRIN_E_WORKSPACE;
            }
          }
  // (2) a grouped launch refuses an item with an empty reduction instead of deriving a slice length of 0 from it and
  //     dividing by that (the SIGFPE of round 3's staged backward: value-initialised items past a flushed group)
  alignas(16) static float dummy[16];
  {
    F32GemmGroup g;
    g.n = 2;
    g.item[0] = {dummy, 4, dummy, 4, dummy, 4, 4, 4, 4};
    g.item[1] = F32GemmGroup::Item();   // M = N = K = 0, NULL operands
    const int rc = launch_gemm_tn_f32_group(g, nullptr, nullptr, 0, nullptr);
    if (rc != DRIN_E_SHAPE) {
      set_error("selftest: launch_gemm_tn_f32_group accepted an item with an empty reduction (status %d)", rc);
      return rc == DRIN_OK ? DRIN_E_SHAPE : rc;
    }
    F32GemmGroup h;
    h.n = 1;
    h.item[0] = F32GemmGroup::Item();
    h.bias_of[0] = nullptr;
    const int rn = launch_gemm_nt_f32_group(h, nullptr, dummy, 16);
    if (rn != DRIN_E_SHAPE) {
      set_error("selftest: launch_gemm_nt_f32_group accepted an empty item (status %d)", rn);
      return rn == DRIN_OK ? DRIN_E_SHAPE : rn;
    }
  }
  // (3) an empty reduction through the single-product entry: nothing to add, nothing launched, nothing divided
  {
    const int rc = launch_gemm_tn(dummy, 4, dummy, 4, dummy, 4, /*M=*/0, 4, 4, DRIN_PREC_F32, nullptr, dummy, 16);
    if (rc != DRIN_OK) return rc;
    if (launch_gemm_tn(dummy, 4, dummy, 4, dummy, 4, /*M=*/-1, 4, 4, DRIN_PREC_F32, nullptr, dummy, 16) != DRIN_E_SHAPE) {
      set_error("selftest: launch_gemm_tn accepted a negative reduction length");
      return DRIN_E_SHAPE;
    }
  }
  set_error("");
  return DRIN_OK;
}

}  // extern "C"
