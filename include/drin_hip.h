/*
 * drin_hip.h - C ABI of libdrin_hip.so: the MI355X (gfx950) DRIN scoring path.
 *
 * The reference (starreeze/drin, /root/reference) has no FFI: its boundary for this path is the
 * Python nn.Module `drin/model.py:156-209` (`Model()` / `Model.forward(batch)`), called from
 * `train.py:27-33`.  This library sits directly underneath a drop-in replacement of that Module
 * (`drin_amd/model.py`) and is what a maintainer of the reference would bind with ctypes
 * (INTEGRATION.md).  Each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions across the boundary.
 *   - every function returns DRIN_OK (0) or a negative drin_status; drin_last_error() gives a
 *     thread-local message for the last failure on the calling thread.
 *   - all data pointers are DEVICE pointers unless a parameter says "host".  The caller owns every
 *     buffer, including the workspace (size from drin_workspace_bytes); the library allocates
 *     nothing and keeps no mutable global state (the opt-in profile of drin_profile_begin apart; environment
 *     probes are read once), so calls are re-entrant.
 *   - every kernel is launched on the caller's `stream` (a hipStream_t passed as void*); the
 *     library never synchronises (drin_index_status apart, which exists to be the caller's synchronisation point).
 *   - THE DEVICE OF A CALL IS THE DEVICE OF ITS STREAM, not the calling thread's current device: every launching entry
 *     point asks the stream (hipStreamGetDevice), makes that device current for the duration of the call - kernel
 *     attributes, launches, memsets and events all land there - and restores the caller's device before it returns.  With
 *     the NULL stream the device is the one that owns the call's workspace / output pointer (hipPointerGetAttributes), and
 *     THAT device's default stream is used.  A single process may therefore drive several GPUs from one thread without
 *     hipSetDevice between calls; pointers of a call must all live on the stream's device.
 *   - tensors are dense row-major fp32 (`float`), index/mask tensors int64 exactly as
 *     `drin/data.py:110-126` collates them.
 */
#ifndef DRIN_HIP_H_
#define DRIN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRIN_ABI_VERSION 7
#define DRIN_API __attribute__((visibility("default")))

typedef enum {
  DRIN_OK = 0,
  DRIN_E_SHAPE = -1,     /* a dimension is out of the supported range / inconsistent            */
  DRIN_E_NULL = -2,      /* a required pointer is NULL                                           */
  DRIN_E_ALIGN = -3,     /* a pointer or leading dimension misses the 16-byte alignment contract */
  DRIN_E_WORKSPACE = -4, /* workspace smaller than drin_workspace_bytes()                        */
  DRIN_E_HIP = -5,       /* a HIP runtime call failed (launch error, bad stream, ...)            */
  DRIN_E_UNSUPPORTED = -6, /* configuration not built (e.g. vector edge features)               */
  DRIN_E_INDEX = -7      /* drin_index_status: a candidate row index was outside [0, num_entities - 1]
                            (the reference's fancy index, drin/data.py:87-93, raises IndexError)  */
} drin_status;

/* Arithmetic of the contractions.  All other arithmetic is fp32.  Every value here is INSIDE the 1e-4 bar of the path on freshly
 * initialised and on trained weights (tests/test_gpu_round4.py, tests/test_gpu_round5.py); the values 2 and 4 (plain one-pass bf16: 5e-4;
 * one bf16 pass for the image contraction: 1.6e-4 on trained weights) were outside it and were removed with ABI 6. */
typedef enum {
  DRIN_PREC_F32 = 0,    /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32): k-ordered fmaf chain        */
  DRIN_PREC_BF16X3 = 1, /* operands split hi+lo bf16, 3 bf16 MFMAs, fp32 accumulate (~fp32): <= 2e-6 on the scores at
                           initialisation, <= 6e-6 on trained weights.  The default.                */
  DRIN_PREC_BF16X3_ALL = 3, /* BF16X3 also for the mention-sized contractions that BF16X3 leaves on
                              the fp32 kernel for latency reasons (used by the parity tests)      */
  DRIN_PREC_BF16X3_IF16 = 5 /* drin_forward_prepared only: precision BY CONTRACTION.  BF16X3 everywhere except the folded
                              entity-image contraction x_i (W_h1 W_ei)^T - 57 % of the path's FLOPs - which runs ONE pass of the
                              FP16 matrix instruction (11-bit operands).  Its result feeds only the layer-1 entity IMAGE vertex,
                              which reaches the score through mean_n(ti' ei') in the layer-2 mention vertex alone
                              (model.py:124-129,143-144; vertex graph :105): the rounding noise is averaged over the N candidates
                              before it meets the score.  fp16's range is made a non-issue by scaling: k_entity_stream writes
                              every image row as fp16(x / 2^ceil(log2 max|x|)) (exact) with the scale beside it, drin_prepare
                              the folded weight as one fp16 plane under one power-of-two scale, and the contraction multiplies
                              both back in its epilogue.  Measured at N = 101: <= 4e-6 on the scores with freshly initialised
                              weights, <= 2e-5 with trained ones (BF16X3 itself: 2e-6 / 6e-6).  Taken only for num_candidates >= 64
                              (at N = 11 the pass costs 8e-6 at initialisation but 1.2e-4 on trained weights), D = 768, R = 2048,
                              per-pair (not table-form) fp32-stored image rows and calls of at least 128 tiles of 256 x 256
                              (~11 000 pairs); every other call - bf16-stored features included: their two-pass contraction reads
                              the rows in place and writing the plane buys nothing - runs BF16X3 bit for bit.  Other entry points return
                              DRIN_E_UNSUPPORTED for it.  */
} drin_precision;

/* Geometry + switches of one forward.  Names follow common/args.py. */
typedef struct {
  int32_t batch;            /* B mentions in this call                                            */
  int32_t num_candidates;   /* N = num_candidates_model (args.py:101), answer slot included       */
  int32_t embed_dim;        /* D = bert_embed_dim = gcn_embed_dim (args.py:25,45)                 */
  int32_t image_dim;        /* R = resnet_embed_dim (args.py:52)                                  */
  int32_t mention_tokens;   /* L = tokens per mention sentence (args.py:72)                       */
  int32_t image_regions;    /* P = resnet_num_region (args.py:53)                                 */
  int32_t mention_objects;  /* Km (args.py:57)                                                    */
  int32_t entity_objects;   /* Ke (args.py:57)                                                    */
  int32_t entity_tokens;    /* T > 0: WikiMEL token-level entity text [B,N,T,D] + mask [B,N,T];
                               0: WikiDiverse pooled entity text [B,N,D] (ghmfc.py:239-249)       */
  int32_t mention_object_inner; /* size of the dim averaged at model.py:78-79 (1 in both datasets)*/
  int32_t entity_image_inner;   /* size of the dim averaged at model.py:43-44 (0 = tensor is 3-D) */
  int32_t entity_object_inner;  /* size of the dim averaged at model.py:82-83 (0 = tensor is 4-D) */
  int32_t num_layers;       /* num_gcn_layers (args.py:26)                                        */
  int32_t dynamic_edges;    /* gcn_edge_type == "dynamic" (args.py:32)                            */
  float edge_enabled[4];    /* gcn_edge_enabled for (tt, ti, it, ii) (args.py:34, model.py:122)   */
  float layer_norm_eps;     /* 1e-5  (nn.LayerNorm default, model.py:119)                         */
  float cosine_eps;         /* 1e-8  (nn.CosineSimilarity default, model.py:57,162)               */
  float miei_eps;           /* 1e-9  (model.py:92)                                                */
  float clip_scale;         /* 100   (model.py:203)                                               */
  int32_t precision;        /* drin_precision                                                     */
  int32_t num_entities;     /* rows of the entity TABLES when drin_batch.entity_index is set, else 0    */
  int32_t vector_edges;     /* gcn_edge_feature == "vector" (args.py:33): edges are [B, N, D], w_m is a Linear,
                               w_u / w_v map D -> D/2 (model.py:112-116,151-152).  Layer-by-layer path only. */
  int32_t feature_dtype;    /* drin_feature_dtype: storage type of the six FEATURE tensors of drin_batch (mention_text,
                               mention_image, mention_object, entity_text, entity_image, entity_object).  With
                               DRIN_FEAT_BF16 those pointers address bf16 arrays of the same shapes - half the bytes
                               of the HBM-bound pass; scores, similarities, masks, weights and ALL arithmetic stay
                               fp32 (a bf16 value is exact in fp32, so the result equals the reference forward on
                               the same features widened to fp32).  drin_forward_prepared only; the other paths
                               return DRIN_E_UNSUPPORTED (widen the features on the caller side). */
  int32_t vertex_activation; /* drin_activation: gcn_vertex_activation (args.py:35; model.py:117,128), default gelu    */
  int32_t edge_activation;   /* drin_activation: gcn_edge_activation   (args.py:36; model.py:118,133), default sigmoid */
  int32_t cache_format;      /* drin_cache_format: row format of the per-entity cache (drin_build_entity_cache /
                                drin_forward_cached only; every other entry point ignores it)                           */
} drin_config;

/* Row format of the per-entity precompute cache.  DRIN_CACHE_F32: every field fp32 ([5 D + R + 4] floats per entity).
 * DRIN_CACHE_MIXED_F16 ("precision by storage"): the fields that carry a vertex or the text-text edge - the two layer-1
 * entity contractions h_t, h_i and the normalised CLS row (model.py:71-76,128,146) - stay fp32; the operands of per-pair
 * SCALARS - W_v1(et0), W_v1(ei0) (the edge update: a mean over D inside a sigmoid, model.py:148-153) and the score-weighted
 * normalised object row (the image-image edge, model.py:84-92, which feeds mi' and ei' alone) - are stored as fp16 with one
 * power-of-two scale per field and row (exact to apply; the row may have any magnitude): [4 D + R / 2 + 4] floats per
 * entity, 16 400 B instead of 23 568 at D = 768, R = 2 048.  Effect on the scores: 2e-7 at N = 101, 6e-7 at N = 11
 * (oracle/precision_emulation.py; measured: tests/test_gpu_round4.py) - below the split-bf16 contractions' own 1.3e-6; with
 * TRAINED weights <= 3e-6 at N = 101 and <= 1.7e-5 at N = 11 against the fp32 rows (profiles/r4_precision_on_trained_weights.txt).
 * The limit of the format: an fp16 field holds its per-pair scalar (an edge logit, the static image-image edge) to ~1e-5, which the
 * mention aggregates mean_n(edge x vertex) (model.py:143-144) average over the N candidates.  A candidate whose image row is 1e6 times
 * the others' DOMINATES that mean, and the aggregate then inherits the error of its ONE edge: measured 1e-5 .. 7e-5 on the scores of
 * such mentions against the fp64 oracle (fp32 rows: 9e-7), reproduced to 1e-5 by the fp64 oracle with the three fields passed through
 * the format's rounding (tests/test_gpu_round5.py).  Inside the 1e-4 bar, outside the 1e-5 guard: tables with rows that far off
 * scale belong in DRIN_CACHE_F32.
 * Needs embed_dim % 8 == 0 and image_dim % 8 == 0. */
typedef enum {
  DRIN_CACHE_F32 = 0,
  DRIN_CACHE_MIXED_F16 = 1
} drin_cache_format;

/* The reference resolves `getattr(torch.nn.functional, name)` (model.py:117-118) - any function name.  Built here: the
 * five below, for the vertices AND for the edges (sigmoid, tanh and relu take their derivative from the stored output; gelu and
 * silu edges keep the pre-activation for the backward pass - Layout::edge_z; goldens tiny_wd_gelu_edge,
 * tiny_wm_silu_edge_vector).  Every entry point takes them: the layer-by-layer forward / backward and the folded
 * inference paths (whose default-activation kernels are separate instantiations, unchanged by the switch). */
typedef enum {
  DRIN_ACT_DEFAULT = 0, /* gelu for vertex_activation, sigmoid for edge_activation (a zero-initialised config is the reference's) */
  DRIN_ACT_GELU = 1,    /* exact-erf gelu (F.gelu default)                                              */
  DRIN_ACT_SIGMOID = 2,
  DRIN_ACT_RELU = 3,
  DRIN_ACT_TANH = 4,
  DRIN_ACT_SILU = 5
} drin_activation;

typedef enum {
  DRIN_FEAT_F32 = 0,
  DRIN_FEAT_BF16 = 1
} drin_feature_dtype;

/* The 14 tensors `Model.forward` unpacks (drin/model.py:165-180), device pointers.
 * mention_text_mask is carried by the reference but never read by the DRIN compute, so it is absent. */
typedef struct {
  const float* mention_text;          /* [B, L, D]                                               */
  const int64_t* mention_start;       /* [B]   already +1 for CLS (data.py:113)                  */
  const int64_t* mention_end;         /* [B]                                                     */
  const float* mention_image;         /* [B, P, R]                                               */
  const float* mention_object;        /* [B, Km, inner, R]                                       */
  const float* mention_object_score;  /* [B, Km]                                                 */
  const float* entity_text;           /* T>0: [B, N, T, D]   T==0: [B, N, D]                     */
  const int64_t* entity_text_mask;    /* T>0: [B, N, T]      T==0: ignored (may be NULL)         */
  const float* entity_image;          /* [B, N, (inner,) R]                                      */
  const float* entity_object;         /* [B, N, Ke, (inner,) R]                                  */
  const float* entity_object_score;   /* [B, N, Ke]                                              */
  const float* miet_similarity;       /* [B, N]  CLIP logits (model.py:178,203)                  */
  const float* mtei_similarity;       /* [B, N]                                                  */
  /* Optional on-device form of the WikiMEL entity-table gather (drin/data.py:87-93).  When non-NULL,
   * entity_text / entity_text_mask / entity_image / entity_object / entity_object_score are TABLES with
   * cfg.num_entities rows ([E, T, D], [E, T], [E, (inner,) R], ...) and candidate (b, n) reads row
   * entity_index[b, n] (clamped to [0, E-1] by drin_forward_prepared) instead of row b*N + n.  Taken by
   * drin_forward_prepared (token-level or pooled tables) and, for training over tables pooled ahead of time
   * (entity_tokens = 0, entity_text_cls set or not, one object per entity, inner dims <= 1, scalar edges, split-bf16
   * precision, >= 1024 pairs), by drin_forward / drin_backward: the static-edge kernels and the vertex-encoder GEMMs
   * (forward x W^T, backward dY^T x) address the table rows through the index - indices must lie in [0, E-1]. */
  const int64_t* entity_index;        /* [B, N] or NULL                                          */
  /* Optional, T == 0 only: the rows the text-text edge compares the mention span with (model.py:73-75: token 0 of
   * the WikiMEL token block) when entity_text carries token means POOLED AHEAD OF TIME - the pooling of
   * ghmfc.py:245-249 has no weights, so a training loop over an entity table can pool every entity once instead of
   * every candidate every step.  NULL: the edge reads entity_text itself (the WikiDiverse layout).
   * drin_forward / drin_backward / drin_edges_fwd; the fused inference entry points return DRIN_E_UNSUPPORTED. */
  const float* entity_text_cls;       /* [B, N, D] or NULL                                       */
  /* Optional, with entity_index: int32[4] in DEVICE memory where the kernels of drin_forward_prepared / drin_forward_cached
   * REPORT an index outside [0, num_entities - 1].  Such an index is always clamped (no out-of-bounds read whatever the
   * caller sends) - but the clamped row is another entity's, where the reference's fancy index (drin/data.py:87-93) raises.
   * The first kernel to meet one sets word 0 to 1 and leaves the pair b * N + n in word 1 and the offending value in words
   * 2 (low half) and 3 (high half); nothing else ever writes the words: STICKY - the caller zeroes them once and reads them
   * wherever it synchronises anyway, or through drin_index_status.  NULL: clamp silently.  (The table form of drin_forward /
   * drin_backward reads rows through the index without clamping: indices must be valid there, see entity_index.) */
  int32_t* index_status;              /* [4] or NULL                                             */
} drin_batch;

/* One GCNLayer's parameters (drin/model.py:109-119); nn.Linear layout weight[out][in]. */
typedef struct {
  const float *w_h, *b_h, *w_u, *b_u, *w_v, *b_v, *ln_weight, *ln_bias;
  const float *w_m, *b_m; /* vector edges only (model.py:112): [D, D], [D]; then w_u / w_v are [D/2, D] */
} drin_layer_params;

#define DRIN_MAX_LAYERS 8

/* The 24 state_dict tensors. */
typedef struct {
  const float *w_mention_text, *b_mention_text;   /* vertex_encoder.mention_text_encoder.final_layer.linear */
  const float *w_entity_text, *b_entity_text;     /* vertex_encoder.entity_text_encoder.final_layer         */
  const float *w_mention_image, *b_mention_image; /* vertex_encoder.mention_image_linear  [D, R]            */
  const float *w_entity_image, *b_entity_image;   /* vertex_encoder.entity_image_linear   [D, R]            */
  drin_layer_params layer[DRIN_MAX_LAYERS];
} drin_params;

/* Gradients w.r.t. the same tensors; every non-NULL pointer is ACCUMULATED into (+=), so the caller
 * zeroes them (torch .grad semantics).  Parameters the output does not depend on (last layer's
 * w_u / w_v, model.py:130-134) are left untouched. */
typedef struct {
  float *w_mention_text, *b_mention_text, *w_entity_text, *b_entity_text;
  float *w_mention_image, *b_mention_image, *w_entity_image, *b_entity_image;
  struct { float *w_h, *b_h, *w_u, *b_u, *w_v, *b_v, *ln_weight, *ln_bias, *w_m, *b_m; } layer[DRIN_MAX_LAYERS];
} drin_param_grads;

/* Optional taps of intermediate values for tests/debugging (any pointer may be NULL).
 * Index l = 0 is the VertexEncoder/EdgeEncoder output, l >= 1 the output of GCN layer l. */
typedef struct {
  float* mention_text_vertex[DRIN_MAX_LAYERS + 1];  /* [B, D]    */
  float* mention_image_vertex[DRIN_MAX_LAYERS + 1]; /* [B, D]    */
  float* entity_text_vertex[DRIN_MAX_LAYERS + 1];   /* [B, N, D] */
  float* entity_image_vertex[DRIN_MAX_LAYERS + 1];  /* [B, N, D] */
  float* edges[DRIN_MAX_LAYERS + 1];                /* [4, B, N] ([4, B, N, D] with vector edges) order tt, ti, it, ii */
} drin_trace;

/* ---- housekeeping ------------------------------------------------------------------------- */
DRIN_API int drin_version(void);                 /* DRIN_ABI_VERSION the library was built with           */
DRIN_API const char* drin_last_error(void);      /* thread-local, never NULL                              */
DRIN_API const char* drin_build_info(void);      /* "gfx950 <compiler> <date>"                            */

/* Fills `cfg` with the reference defaults (eps values, scale, all edges enabled, 2 dynamic layers,
 * WikiDiverse geometry).  Replaces the import-time globals of common/args.py. */
DRIN_API int drin_default_config(drin_config* cfg);

/* Host-side self-checks of the library's launch-sequencing code, for the sanitizer build and CI: no kernel is launched,
 * no GPU is needed.  Checks that the workspace layout gives the mention-sized weight-gradient group of drin_backward
 * (train.py:33-34 through model.py:164-209) its worst-case slice scratch at every batch size, and that the grouped GEMM
 * launchers refuse an item with an empty reduction (DRIN_E_SHAPE) instead of dividing by a slice length derived from it.
 * DRIN_OK, or the first failing check's status with its message in drin_last_error().  No reference counterpart. */
DRIN_API int drin_host_selftest(void);

/* Bytes of device workspace drin_forward / drin_backward need for `cfg` (0 on error). */
DRIN_API size_t drin_workspace_bytes(const drin_config* cfg, int for_training);

/* ---- the path ----------------------------------------------------------------------------- */

/* Parameter-free edge builder: EdgeEncoder.forward (drin/model.py:60-94) + edge assembly
 * (model.py:201-204).  Writes edges[4][B][N] in order (tt, ti, it, ii) and, if non-NULL, the
 * span mean of the mention text [B, D] (ghmfc.py:54-60), which the vertex encoder reuses. */
DRIN_API int drin_edges_fwd(const drin_config* cfg, const drin_batch* batch, float* edges, float* span_mean,
                   void* stream);

/* Input pooling of VertexEncoder.forward: entity token mean (ghmfc.py:245-249, only T>0),
 * mention region mean (model.py:41), entity image inner mean (model.py:43-44).  Any output may be
 * NULL to skip it.  pooled_entity_text [B,N,D], pooled_mention_image [B,R], pooled_entity_image [B,N,R].
 * Only the inputs of the requested outputs are read (pooling an entity TABLE once passes the text and its mask alone).
 * With cfg.feature_dtype == DRIN_FEAT_BF16 the entity token pooling reads bf16 tokens in place (fp32 sums of the
 * exactly widened values); the two image means are then not available here (DRIN_E_UNSUPPORTED). */
DRIN_API int drin_pool_fwd(const drin_config* cfg, const drin_batch* batch, float* pooled_entity_text,
                  float* pooled_mention_image, float* pooled_entity_image, void* stream);

/* y[m, n] = sum_k x[m, k] * w[n, k] + bias[n]   (nn.Linear, weight [n_out][k]); bias may be NULL.
 * The building block behind every Linear of the path (ghmfc.py:69,250; model.py:42,45,128,150-153). */
DRIN_API int drin_linear_fwd(const float* x, const float* w, const float* bias, float* y, int64_t rows,
                    int32_t n_out, int32_t k, int32_t precision, void* stream);

/* Backward of the same Linear (what autograd does for every nn.Linear of the path under train.py:33-34):
 *   dx[m, k]  = sum_n dy[m, n] w[n, k]      (written; dx may be NULL)
 *   dw[n, k] += sum_m dy[m, n] x[m, k]      (accumulated; dw may be NULL)
 *   db[n]    += sum_m dy[m, n]              (accumulated; db may be NULL)
 * `scratch` (`scratch_floats` floats, may be NULL / 0): n_out * k floats hold the transposed weight the split-bf16 dx
 * product runs against (without them dx stays on the exact fp32 kernel); with at least 28 * n_out * k floats the
 * split-bf16 dw product stores its reduction slices there and adds them in order (bit-reproducible) instead of
 * using fp32 atomics, and a partly filled last round of 256 x 256 output tiles of the dx product is split along the
 * reduction over the idle CUs (partial tiles in the scratch behind the transposed weight, added in order).
 * Small problems take the exact fp32 kernels in every precision. */
DRIN_API int drin_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                             int64_t rows, int32_t n_out, int32_t k, int32_t precision, float* scratch,
                             size_t scratch_floats, void* stream);

/* Model.forward (drin/model.py:164-209): scores[B, N] = cos(mt'', et'').
 * `keep_for_backward` != 0 lays the intermediates backward needs out in `workspace`, which must
 * then stay untouched until drin_backward returns.  `trace` may be NULL. */
DRIN_API int drin_forward(const drin_config* cfg, const drin_batch* batch, const drin_params* params,
                 void* workspace, size_t workspace_bytes, float* scores, int keep_for_backward,
                 const drin_trace* trace, void* stream);

/* drin_forward for a training loop that runs the previous step's gradient all-reduce and optimiser update on ANOTHER
 * stream (the reference runs one device, train.py:117-118, and has no counterpart).  `params_ready_event`: a
 * hipEvent_t recorded on that stream behind the update, or NULL (= drin_forward).  The parameter-free head of the pass - the
 * pooling kernels and the static edges (ghmfc.py:54-60,245-249; model.py:41-45,60-94), which read the batch alone: 0.2 of
 * the 1.35 ms of a WikiMEL step at batch 64 - is enqueued first; `stream` then waits for the event (hipStreamWaitEvent)
 * before the first launch that reads `params`.  Same kernels, same results as drin_forward. */
DRIN_API int drin_forward_staged(const drin_config* cfg, const drin_batch* batch, const drin_params* params,
                 void* workspace, size_t workspace_bytes, float* scores, int keep_for_backward,
                 const drin_trace* trace, void* params_ready_event, void* stream);

/* Backward of drin_forward: given grad_scores[B, N], accumulates parameter gradients into `grads`
 * (loss.backward() of train.py:33-34 through model.py:164-209).  Inputs carry no gradient in the
 * reference (precomputed features), so none is produced. */
DRIN_API int drin_backward(const drin_config* cfg, const drin_batch* batch, const drin_params* params,
                  void* workspace, size_t workspace_bytes, const float* grad_scores,
                  const drin_param_grads* grads, void* stream);

/* drin_backward in two stages for a data-parallel caller (the reference runs one device, train.py:117-118,
 * and has no counterpart).  `layers_ready_event`: a hipEvent_t the caller created, or NULL (= drin_backward).  The event
 * is recorded on `stream` once every gradient of the GCN layers (the `layer[l]` tensors of `grads`) is complete; only
 * the four vertex encoders' gradients (w_/b_ mention_text, entity_text, mention_image, entity_image) are written after
 * it.  A caller whose gradient bucket keeps the two groups apart starts the all-reduce of the layers' part behind the
 * event, under the vertex encoders' weight-gradient products.  The two parts deal the chip's workgroups separately, so the
 * pair-sized weight gradients differ from drin_backward's in the last bits (another split of the same sums; no atomics:
 * each entry point reproduces its own bits run after run) and the pass costs two part-filled launches more - measure
 * before preferring it to drin_forward_staged, which hides the collective without touching the backward pass. */
DRIN_API int drin_backward_staged(const drin_config* cfg, const drin_batch* batch, const drin_params* params,
                  void* workspace, size_t workspace_bytes, const float* grad_scores,
                  const drin_param_grads* grads, void* layers_ready_event, void* stream);

/* ---- fused two-layer inference path ------------------------------------------------------------
 * Same result as drin_forward (fp32 re-association only) for the default geometry num_layers == 2,
 * with the weight-only products folded once per weight version (csrc/fused_forward.hip):
 *   drin_prepare           folds the weights into `prepared` (drin_prepared_bytes; caller-owned, reusable
 *                          for every batch until a parameter changes)
 *   drin_forward_prepared  Model.forward (drin/model.py:164-209) without autograd state
 * drin_fused_supported returns DRIN_OK when `cfg` can take this path (else use drin_forward). */
DRIN_API int drin_fused_supported(const drin_config* cfg);
DRIN_API size_t drin_prepared_bytes(const drin_config* cfg);
DRIN_API size_t drin_fused_workspace_bytes(const drin_config* cfg);
DRIN_API int drin_prepare(const drin_config* cfg, const drin_params* params, void* prepared, size_t prepared_bytes,
                          void* stream);
DRIN_API int drin_forward_prepared(const drin_config* cfg, const drin_batch* batch, const drin_params* params,
                                   const void* prepared, void* workspace, size_t workspace_bytes, float* scores,
                                   void* stream);
/* How the row-streaming kernels of the two folded paths divide a mention's candidate list for THIS call size: the number of
 * workgroups (candidate chunks) per mention of k_entity_stream / k_pair_layer1 / k_pair_final (cached == 0:
 * drin_forward_prepared - 16 candidates per workgroup, 48 from 1 024 mentions, the whole list up to 128 from 2 048) or of
 * k_cached_pairs (cached != 0: drin_forward_cached - 64 candidates, 128 from 2 048 mentions; 16 off the exact widths).  The
 * per-mention sums are grouped by these chunks, so a mention's scores in calls of different sizes agree to fp32
 * re-association only (<= 5e-6 measured), and are the same bits every run within one call size.  Introspection for
 * tests and benchmarks (which instantiation did this call take?); 0 when cfg is off the folded paths.  No reference
 * counterpart. */
DRIN_API int32_t drin_workgroups_per_mention(const drin_config* cfg, int32_t cached);

/* How drin_forward_prepared evaluates the folded entity-image contraction x_i (W_h1 W_ei)^T - 57 % of the path's FLOPs - for this
 * configuration (`indexed`: drin_batch.entity_index is set): 0 = the exact fp32 kernel (DRIN_PREC_F32), else the number of
 * bf16 / fp16 MFMA passes: 3 (split-bf16), 2 (bf16-stored image rows are exact in their hi plane), 1 (DRIN_PREC_BF16X3_IF16 where its
 * gate holds: see drin_precision).  The library's own gate, for callers that account executed FLOPs (bench.py). */
DRIN_API int32_t drin_image_contraction_passes(const drin_config* cfg, int32_t indexed);

/* The caller-side check of drin_batch.index_status: waits for `stream` (the ONE entry point that synchronises: it exists to be
 * the point where the caller would synchronise anyway), copies the four words to the host and returns DRIN_OK, or
 * DRIN_E_INDEX with the pair and the value in drin_last_error() - what drin/data.py:87-93 reports as an IndexError - after
 * zeroing the words again.  `index_status` is the device pointer given in drin_batch. */
DRIN_API int drin_index_status(int32_t* index_status, void* stream);

/* ---- per-entity precompute cache for table-form inference ---------------------- *
 * With frozen weights, what the first GCN layer takes from an entity (its rows of the entity_* tables,
 * `drin/data.py:87-93`) does not depend on the mention: `W_h1 W_et x_t`, `W_h1 W_ei x_i`, `W_v1(et0)`,
 * `W_v1(ei0)`, the normalised CLS row and the score-weighted normalised object row are computed ONCE per
 * entity and weight version into `cache` ([num_entities][5 D + R + 4] fp32; cfg.cache_format = DRIN_CACHE_MIXED_F16:
 * [num_entities][4 D + R / 2 + 4] floats of fp32 and scaled-fp16 fields, see drin_cache_format).  drin_forward_cached then
 * scores table-form batches (drin_batch.entity_index + the mention-side tensors + the two similarity
 * matrices; the entity_* pointers are not read) with one gathered pass over cache rows and the layer-2
 * contraction - same scores as drin_forward_prepared to fp32 re-association.  `cfg.num_entities` = table
 * rows; `prepared` = the drin_prepare buffer of the same weights.  `tables`: a drin_batch whose entity_*
 * fields point at the tables (other fields unused). */
DRIN_API size_t drin_entity_cache_bytes(const drin_config* cfg);
DRIN_API size_t drin_entity_cache_build_workspace_bytes(const drin_config* cfg);
DRIN_API size_t drin_cached_workspace_bytes(const drin_config* cfg);
DRIN_API int drin_build_entity_cache(const drin_config* cfg, const drin_batch* tables, const drin_params* params,
                                     const void* prepared, void* cache, size_t cache_bytes, void* workspace,
                                     size_t workspace_bytes, void* stream);
DRIN_API int drin_forward_cached(const drin_config* cfg, const drin_batch* batch, const drin_params* params,
                                 const void* prepared, const void* cache, void* workspace, size_t workspace_bytes,
                                 float* scores, void* stream);

/* Building blocks of the split-bf16 precision: x = hi + lo with hi, lo bf16 planes (n % 4 == 0), and the
 * contraction y = x w^T (+ bias) on such planes by LDS-DMA + bf16 MFMA (k % 32 == 0).  Plane pointers are
 * device pointers to bf16 arrays with the row strides of the fp32 originals.  x_lo may be NULL when x is
 * exact in bf16 (then two MFMAs per tile pair instead of three); the weight planes are both required. */
DRIN_API int drin_split_planes(const float* x, void* hi, void* lo, int64_t n, void* stream);
DRIN_API int drin_linear_planes_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo,
                                    const float* bias, float* y, int64_t rows, int32_t n_out, int32_t k,
                                    void* stream);

/* ---- caller-side loss and metric of one batch --------------------------------- *
 * Replaces, for scores already on the device, `TripletLoss.forward` (common/utils.py:26-43, called at
 * train.py:34) and `TopkAccuracy.update` (utils.py:60-66, train.py:36-37), and the autograd backward of
 * the former:
 *   scores   [batch, num_candidates] fp32 (num_candidates includes the answer slot, dropped like
 *            utils.py:36-37);  answer [batch, num_candidates-1] uint8 one-hot (all-zero row: no gold);
 *   loss     [1] fp32 (written);  d_scores [batch, num_candidates] fp32 = d loss / d scores, or NULL;
 *   correct  [num_topk] int64, ACCUMULATED (+=) like the metric's `correct` state; `topk` is a HOST array
 *            of up to 8 values of k; the caller adds `batch` to its `total`.
 * Ties count as correct (`y_pred >= lower`), NaN scores order as in torch.topk, a NaN loss propagates. */
DRIN_API size_t drin_loss_workspace_bytes(int32_t batch);
DRIN_API int drin_triplet_topk(const float* scores, const uint8_t* answer, int32_t batch, int32_t num_candidates,
                               float margin, const int32_t* topk, int32_t num_topk, float* loss, float* d_scores,
                               int64_t* correct, void* workspace, size_t workspace_bytes, void* stream);

/* ---- caller-side optimiser step of the training configs ------------------------------------------ *
 * `torch.optim.Adam(self.parameters(), lr)` of train.py:55-56 (torch defaults: betas (0.9, 0.999), eps 1e-8, no weight
 * decay, no amsgrad), one step over `n` contiguous fp32 elements in ONE launch - a flat bucket holding every parameter
 * that receives a gradient, its gradient bucket (what drin_backward wrote / RCCL all-reduced) and the two moment buckets.
 * The op sequence and the per-op fp32 rounding are those of torch's default multi-tensor implementation, so a training
 * loop stepped with it tracks the reference's loop bit for bit (csrc/optim_kernels.hip).  Scalars, formed on the HOST in
 * double precision as torch/optim/adam.py forms them for step t (1-based) and rounded to fp32 here:
 *   lerp_weight = 1 - beta1;  one_minus_beta2 = 1 - beta2;  bias_correction2_sqrt = (1 - beta2^t)^0.5;
 *   neg_step_size = -(lr / (1 - beta1^t)).
 * Every `a + s x` of that sequence is one fma, sqrt and the divisions are correctly rounded - what torch's kernels do on
 * this GPU (tools/adam_diag.py). */
DRIN_API int drin_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                            float lerp_weight, float beta2, float one_minus_beta2, float bias_correction2_sqrt, float eps,
                            float neg_step_size, void* stream);

/* ---- in-process kernel timing (bench.py's roofline leg) ---------------------------------------- */

/* Kernel classes the launches are attributed to. */
typedef enum {
  DRIN_KC_GEMM = 0,   /* k_gemm_f32: exact fp32 MFMA contractions (all backward products too)     */
  DRIN_KC_POOL = 1,   /* input pooling: span / region / token means                               */
  DRIN_KC_EDGE = 2,   /* static edge builders: cosine rows, miei, scaling                          */
  DRIN_KC_GCN = 3,    /* aggregation, LayerNorm+GELU, edge update, their backward                  */
  DRIN_KC_STREAM = 4, /* k_entity_stream / k_cached_pairs: the single HBM-bound pass over entity bytes  */
  DRIN_KC_GEMM_X3 = 5,     /* k_gemm_bf16x3: split-bf16 contraction, fp32 operands split on the fly */
  DRIN_KC_GEMM_PLANES = 6, /* k_gemm_x3_planes: split-bf16 contraction on pre-split planes (LDS-DMA) */
  DRIN_KC_OPTIM = 7,       /* drin_adam_step                                                        */
  DRIN_KC_COUNT = 8
} drin_kernel_class;

/* While a profile is open, every launch the library makes - from any thread, e.g. drin_backward on
 * autograd's thread - is bracketed by hipEvents on its stream.  drin_profile_end synchronises those
 * events, returns per-class GPU milliseconds and launch counts (arrays of DRIN_KC_COUNT) and closes the
 * profile.  One profile per process at a time; this is the library's only process-wide state and it is
 * inert unless a profile is open.  The loss kernels (drin_triplet_topk) count under DRIN_KC_EDGE.  The events belong to the
 * device that is current when drin_profile_begin runs: a profile times the calls whose streams live on THAT device (launches on
 * another device run untimed - their results are unaffected). */
DRIN_API int drin_profile_begin(int max_launches);
DRIN_API int drin_profile_end(double* ms_by_class, int64_t* launches_by_class);
DRIN_API const char* drin_kernel_class_name(int kernel_class);

#ifdef __cplusplus
}
#endif
#endif /* DRIN_HIP_H_ */
