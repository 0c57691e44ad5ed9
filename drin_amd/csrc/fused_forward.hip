// Fused two-layer inference pipeline: drin_prepare / drin_forward_prepared.
//
// The reference evaluates, per (mention, candidate) pair, six D x D and one D x R contractions
// (10.2 MFLOP after dead-code elimination).  For inference most of them are linear maps applied to
// linear maps, so they fold (fp32 re-association only; parity is pinned by the same golden tests):
//
//   layer-1 entity vertices (model.py:128,146):   W_h(e_a mt + e_b mi + W_et x + b_et) + b_h
//        = x (W_h W_et)^T + e_a (W_h mt) + e_b (W_h mi) + (W_h b_et + b_h)        -> C_t, hm, c_t
//   layer-1 dynamic edges (model.py:148-153):     mean_d(W_u(u) * (W_v(W_et x + b_et) + b_v))
//        = ((W_v W_et)^T W_u(u)) . x / D + W_u(u) . (W_v b_et + b_v) / D           -> q, kappa
//   layer-1 mention vertices (model.py:143-144):  mean_n(e W_et x + e b_et)
//        = W_et (sum_n e x) / N + b_et (sum_n e) / N                               -> S, sigma
//   layer-2 (last): only mt'' and et'' reach the score (model.py:207-209); its edge update, mi'', ei''
//   are dead, and  W_h2(e_a mt' + e_b mi' + et') = et' W_h2^T + e_a (W_h2 mt') + e_b (W_h2 mi').
//
// Per pair this leaves three contractions - x_t C_t^T (D x D), x_i C_i^T (D x R), et' W_h2^T (D x D):
// 5.5 MFLOP - plus ONE streaming pass over the entity bytes (k_entity_stream) and two row kernels.
// The folded matrices depend only on the weights: drin_prepare computes them once per weight version
// into a caller-owned buffer.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fused.h"
#include "internal.h"
#include "layout.h"

namespace drin {

// DRIN_PREC_BF16X3_IF16: the shortest candidate list whose mean averages the one-pass image contraction's noise far enough
constexpr int kMixedMinCandidates = 64;

struct FusedLayout {  // workspace offsets in floats
  size_t span_mean, mimg, vm0, hmfu, q, e0m, e1m, xt, s_part, s_text, s_img, sig, tm, tm2, agg1, vm1, hm2, h_text, h_image,
      et1, s2_part, agg2, mt2, p_xt, p_xi, p_et1, xi_scale, splitk, splitk_floats, pair_splitk, pair_splitk_floats, total;
  int chunks;
  void build(const drin_config& c) {
    const size_t B = c.batch, N = c.num_candidates, D = c.embed_dim, R = c.image_dim, M = B * N;
    // a workgroup owns 16 consecutive candidates of one mention (4 per wave) - or, from 2 048 mentions up, where the mentions
    // alone fill the chip eight times over, up to 128 of them (WikiMEL: the whole list): one prologue of mention-side vectors
    // and one cross-wave reduction per mention instead of seven, no partial sums to write and reduce (same box, three
    // alternating runs: k_reduce_stream_partials' 0.15 ms gone, the stream kernel itself unchanged, step -0.19 ms;
    // profiles/r3_stream_chunk_ab.txt).  The grouping of the per-mention sums therefore depends on the size of the call: a
    // mention's scores in a small and in a large call differ by fp32 re-association (as through the tile choices of the
    // mention-sized products), within one call size they are the same bits every run.
    chunks = (int)(B >= 2048 ? (N + 127) / 128 : B >= 1024 ? (N + 47) / 48 : (N + 15) / 16);   // (1 024 mentions: 48 candidates, 4.56 -> 4.49 ms)
    {
      static const char* sc_env = getenv("DRIN_STREAM_CHUNK");   // probe: candidates per workgroup of the stream kernel
      const int per = sc_env ? atoi(sc_env) : 0;
      if (per >= 16) chunks = (int)((N + per - 1) / per);
    }
    size_t off = 0;
    auto take = [&off](size_t n) {
      const size_t o = off;
      off += (n + 63) & ~(size_t)63;
      return o;
    };
    span_mean = take(B * D);
    mimg = take(B * R);
    vm0 = take(2 * B * D);
    hmfu = take(2 * B * 2 * D);
    q = take(2 * B * (D + R));
    e0m = take(4 * M);
    e1m = take(4 * M);
    xt = take(c.entity_tokens > 0 ? M * D : 0);
    s_part = take(B * chunks * (2 * D + 2 * R + 4));
    s_text = take(2 * B * D);
    s_img = take(2 * B * R);
    sig = take(4 * B);
    tm = take(2 * B * D);
    tm2 = take(2 * B * D);
    agg1 = take(2 * B * D);
    vm1 = take(2 * B * D);
    hm2 = take(2 * B * D);
    h_text = take(M * D);   // re-used for et1 W_h2^T once layer 1 is done
    h_image = take(M * D);
    et1 = take(M * D);
    s2_part = take(B * chunks * 2 * D);
    agg2 = take(B * D);
    mt2 = take(B * D);
    const bool planes = c.precision == DRIN_PREC_BF16X3 || c.precision == DRIN_PREC_BF16X3_ALL || c.precision == DRIN_PREC_BF16X3_IF16;
    p_xt = take(planes ? M * D : 0);   // hi plane (M*D bf16) then lo plane
    // image planes, sized by use: the table form (cfg.num_entities > 0: entity_index gathers rows, so the stream kernel
    // writes them as (hi, lo) bf16 planes) M R floats; DRIN_PREC_BF16X3_IF16 the one fp16 plane, M R / 2; gathered split-bf16
    // batches read the image rows in place: nothing (3.4 GB less workspace at the headline batch)
    p_xi = take(!planes ? 0 : c.num_entities > 0 ? M * R : c.precision == DRIN_PREC_BF16X3_IF16 ? M * R / 2 : 0);
    p_et1 = take(planes ? M * D : 0);
    xi_scale = take(c.precision == DRIN_PREC_BF16X3_IF16 ? M : 0);   // per-pair power-of-two scale of the image row
    // split-K partials of the mention-sized exact-fp32 products (small batches: the call is a chain of ~25 launches)
    // (in split-bf16 precision at least 192 partial 256 x 256 tiles, so that a partly filled last round of tiles of the
    //  larger products can split K over the idle CUs - gemm_bf16x3.hip)
    splitk_floats = 2 * B <= 512 ? 8 * 2 * B * (D + R) : 0;
    if (planes && splitk_floats < (size_t)192 * 65536) splitk_floats = (size_t)192 * 65536;   // 64 tail tiles x 3 partials
    splitk = take(splitk_floats);
    // ... and of the pair-sized split-bf16 products when the whole batch is less than half a round of 256 x 256 tiles
    pair_splitk_floats = 0;
    {
      const size_t tiles = ((M + 255) / 256) * ((D + 255) / 256);
      if (planes && tiles <= 128)
        for (size_t s : {16, 8, 6, 4, 3, 2})
          if (tiles * s <= 256) {
            pair_splitk_floats = s * ((M + 255) / 256 * 256) * D;
            break;
          }
    }
    pair_splitk = take(pair_splitk_floats);
    total = off;
  }
};

// The configuration part of the DRIN_PREC_BF16X3_IF16 gate (what is left - 16-byte alignment of workspace slots - holds by the
// workspace contract): split-bf16 widths, per-pair fp32-stored rows, a list long enough to average the pass's rounding, the exact
// widths the stream kernel's fp16 hand-over is instantiated for, at least half a round of 256 x 256 tiles.
static bool if16_gate(const drin_config* c, bool indexed) {
  const int64_t M = (int64_t)c->batch * c->num_candidates;
  return c->precision == DRIN_PREC_BF16X3_IF16 && !indexed && c->feature_dtype == DRIN_FEAT_F32 &&
         c->num_candidates >= kMixedMinCandidates && c->embed_dim == 768 && c->image_dim == 2048 &&
         cdiv(M, 256) * cdiv(c->embed_dim, 256) >= 128;
}

int fused_supported(const drin_config* c) {
  if (c->num_layers != 2) {
    set_error("fused path: built for num_layers == 2 (got %d); use drin_forward", c->num_layers);
    return DRIN_E_UNSUPPORTED;
  }
  if (c->vector_edges) {
    set_error("fused path: vector edge features (model.py:112-116) run on the layer-by-layer path; use drin_forward");
    return DRIN_E_UNSUPPORTED;
  }
  if (c->mention_object_inner > 1 || c->entity_object_inner > 1 || c->entity_image_inner > 1) {
    set_error("fused path: inner feature dims > 1 need the pooled path of drin_forward");
    return DRIN_E_UNSUPPORTED;
  }
  const bool tiny = c->embed_dim <= 256 && c->image_dim <= 256;
  const bool full = c->embed_dim <= 768 && c->image_dim <= 2048;
  if (!tiny && !full) {
    set_error("fused path: D=%d R=%d outside the built instantiations", c->embed_dim, c->image_dim);
    return DRIN_E_UNSUPPORTED;
  }
  return DRIN_OK;
}

}  // namespace drin

using namespace drin;

extern "C" {

int drin_split_planes(const float* x, void* hi, void* lo, int64_t n, void* stream) {
  DRIN_BIND_DEVICE(stream, x, "drin_split_planes");
  if (!x || !hi || !lo) {
    set_error("drin_split_planes: NULL argument");
    return DRIN_E_NULL;
  }
  return launch_split_planes(x, hi, lo, n, (hipStream_t)stream);
}

int drin_linear_planes_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                           float* y, int64_t rows, int32_t n_out, int32_t k, void* stream) {
  DRIN_BIND_DEVICE(stream, y, "drin_linear_planes_fwd");
  if (!x_hi || !w_hi || !w_lo || !y) {
    set_error("drin_linear_planes_fwd: NULL argument");
    return DRIN_E_NULL;
  }
  return launch_gemm_x3_planes(x_hi, x_lo, k, w_hi, w_lo, k, bias, y, n_out, rows, n_out, k, (hipStream_t)stream);
}

int drin_fused_supported(const drin_config* cfg) {
  DRIN_TRY(validate_config(cfg));
  return fused_supported(cfg);
}

size_t drin_prepared_bytes(const drin_config* cfg) {
  if (validate_config(cfg) != DRIN_OK || fused_supported(cfg) != DRIN_OK) return 0;
  Prepared P;
  P.build(*cfg);
  return P.total * sizeof(float);
}

int drin_prepare(const drin_config* cfg, const drin_params* params, void* prepared, size_t prepared_bytes,
                 void* stream) {
  DRIN_BIND_DEVICE(stream, prepared, "drin_prepare");
  RoctxRange range("drin_prepare");
  DRIN_TRY(validate_config(cfg));
  DRIN_TRY(fused_supported(cfg));
  if (!params || !prepared) {
    set_error("drin_prepare: NULL argument");
    return DRIN_E_NULL;
  }
  Prepared P;
  P.build(*cfg);
  if (prepared_bytes < P.total * sizeof(float) || !aligned16(prepared)) {
    set_error("drin_prepare: buffer has %zu bytes (needs %zu) or is not 16-byte aligned", prepared_bytes,
              P.total * sizeof(float));
    return DRIN_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* pb = (float*)prepared;
  const int D = cfg->embed_dim, R = cfg->image_dim;
  const drin_layer_params& L1 = params->layer[0];
  const int F32 = DRIN_PREC_F32;  // the folds are computed once, in exact fp32
  auto copy = [&](float* dst, const float* src, size_t n) -> int {
    hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? DRIN_OK : hip_fail(e, "hipMemcpyAsync(prepare)");
  };
  DRIN_TRY(copy(pb + P.wcat1, L1.w_h, (size_t)D * D));
  DRIN_TRY(copy(pb + P.wcat1 + (size_t)D * D, L1.w_u, (size_t)D * D));
  {
    hipError_t e = hipMemsetAsync(pb + P.bcat1, 0, D * sizeof(float), st);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(prepare)");
  }
  DRIN_TRY(copy(pb + P.bcat1 + D, L1.b_u, D));
  // E = W_v1 [W_et | W_ei]   (y[m, n] = sum_k x[m, k] w[k, n])
  DRIN_TRY(launch_gemm_nn(L1.w_v, D, params->w_entity_text, D, pb + P.etmp, D + R, D, D, D, false, F32, st));
  DRIN_TRY(launch_gemm_nn(L1.w_v, D, params->w_entity_image, R, pb + P.etmp + D, D + R, D, R, D, false, F32, st));
  DRIN_TRY(launch_transpose(pb + P.etmp, pb + P.ecat, D, D + R, st));
  // k = b_e W_v1^T + b_v1 as row vectors
  DRIN_TRY(launch_gemm_nt(params->b_entity_text, D, L1.w_v, D, L1.b_v, pb + P.k_t, D, 1, D, D, false, F32, st));
  DRIN_TRY(launch_gemm_nt(params->b_entity_image, D, L1.w_v, D, L1.b_v, pb + P.k_i, D, 1, D, D, false, F32, st));
  // C = W_h1 [W_et | W_ei],  cb = b_e W_h1^T + b_h1
  DRIN_TRY(launch_gemm_nn(L1.w_h, D, params->w_entity_text, D, pb + P.c_txt, D, D, D, D, false, F32, st));
  DRIN_TRY(launch_gemm_nn(L1.w_h, D, params->w_entity_image, R, pb + P.c_img, R, D, R, D, false, F32, st));
  DRIN_TRY(launch_gemm_nt(params->b_entity_text, D, L1.w_h, D, L1.b_h, pb + P.cb_t, D, 1, D, D, false, F32, st));
  DRIN_TRY(launch_gemm_nt(params->b_entity_image, D, L1.w_h, D, L1.b_h, pb + P.cb_i, D, 1, D, D, false, F32, st));
  // bf16 hi / lo planes of the pair-sized GEMM weights (split-bf16 precision)
  {
    const size_t dd = (size_t)D * D, dr = (size_t)D * R;
    __bf16* q = reinterpret_cast<__bf16*>(pb + P.p_ctxt);
    DRIN_TRY(launch_split_planes(pb + P.c_txt, q, q + dd, dd, st));
    q = reinterpret_cast<__bf16*>(pb + P.p_cimg);
    DRIN_TRY(launch_split_planes(pb + P.c_img, q, q + dr, dr, st));
    // DRIN_PREC_BF16X3_IF16: one fp16 plane under one power-of-two scale (weights of any magnitude stay inside fp16's range)
    DRIN_TRY(launch_to_f16_scaled(pb + P.c_img, pb + P.p_cimg_f16, (int64_t)dr, pb + P.cimg_f16_scale, st));
    q = reinterpret_cast<__bf16*>(pb + P.p_wh2);
    DRIN_TRY(launch_split_planes(params->layer[1].w_h, q, q + dd, dd, st));
    // mention-sized GEMM weights (they matter for short candidate lists: at N = 11 the mention side is a third
    // of the contraction work)
    struct { size_t off; const float* w; size_t n; } extra[] = {
        {P.p_wmt, params->w_mention_text, dd},  {P.p_wmi, params->w_mention_image, dr},
        {P.p_wcat1, pb + P.wcat1, 2 * dd},      {P.p_ecat, pb + P.ecat, dd + dr},
        {P.p_wet, params->w_entity_text, dd},   {P.p_wei, params->w_entity_image, dr},
        {P.p_wh1, L1.w_h, dd}};
    for (const auto& e : extra) {
      q = reinterpret_cast<__bf16*>(pb + e.off);
      DRIN_TRY(launch_split_planes(e.w, q, q + e.n, e.n, st));
    }
  }
  return DRIN_OK;
}

}  // extern "C"

namespace drin {

// The whole forward on the caller's stream: the one pass over the entity bytes (HBM-bound), then the contractions and row kernels
// behind it (MFMA-bound).  (Running the two halves of consecutive mention chunks side by side - two streams, forced co-residency,
// disjoint CU masks - was built in round 3 and measured slower every time; removed in round 5: profiles/r3_pipeline_probe.txt.)
static int forward_prepared_on_stream(const drin_config* cfg, const drin_batch* b, const drin_params* params, const void* prepared,
                                      void* workspace, float* scores, hipStream_t st) {
  FusedLayout L;
  L.build(*cfg);
  Prepared P;
  P.build(*cfg);
  float* ws = (float*)workspace;
  const float* pb = (const float*)prepared;
  const int B = cfg->batch, N = cfg->num_candidates, D = cfg->embed_dim, R = cfg->image_dim;
  const int64_t M = (int64_t)B * N;
  if (B == 0) return DRIN_OK;
  const int prec = cfg->precision == DRIN_PREC_BF16X3_IF16 ? (int)DRIN_PREC_BF16X3 : cfg->precision;
  const bool dyn = cfg->dynamic_edges != 0;
  const bool tokens = cfg->entity_tokens > 0;
  // split-bf16 precision: the producers write bf16 hi / lo planes and the three pair-sized contractions
  // run on the LDS-DMA kernel of gemm_x3_planes.hip
  const bool planes = (prec == DRIN_PREC_BF16X3 || prec == DRIN_PREC_BF16X3_ALL) && (D % 32 == 0) && (R % 32 == 0);
  const bool indexed = b->entity_index != nullptr;
  // The image rows are read in place by their contraction (fp32, split on the fly): writing 8 KB/pair of
  // planes from the stream kernel costs it more (measured twice on one box: +1.0 ms at B = 4096) than the LDS-DMA
  // kernel gains on that GEMM (-0.35 ms).  Only the table form, which has to gather the rows anyway, writes them.
  const bool xi_planes = planes && indexed;
  // DRIN_PREC_BF16X3_IF16: x_i C_i^T in ONE pass of the fp16 matrix instruction on single planes.  k_entity_stream, which holds every
  // image row in registers anyway, writes it as fp16(x / 2^ceil(log2 max|x|)) - 4 KB per pair - with the scale beside it; the folded
  // weight is one fp16 plane under one scale (drin_prepare); the all-DMA four-phase kernel multiplies both back in its epilogue.
  // For gathered (per-pair) fp32-stored rows at the exact widths in calls of at least half a round of 256 x 256 tiles; everything
  // else runs the three passes, bit for bit.  Measured (profiles/r5_*if16*): the contraction 3.4 -> 1.29 ms, the stream kernel +0.70 ms
  // for the 1.69 GB it now writes (HBM writes inside a read-bound pass cost ~2.5 x their bytes) - the headline batch 15.8 -> 14.6 ms,
  // what round 4's register-staged fp32-A kernel reached too (1.80 ms, nothing written).  bf16-STORED rows do NOT take the pass: their
  // two-pass contraction reads them in place (2.07 ms); the plane costs their stream kernel +0.88 ms for -0.78 ms: measured, a wash.
  // (the candidate-count gate: with freshly initialised weights the fp16 pass costs 8e-6 at N = 11, but once the weights are
  //  TRAINED the vertex -> score map steepens and 11 candidates average too little - 1.2e-4 after 200 Adam steps, outside the bar;
  //  at N = 101 the same weights give 2e-5: profiles/r4_precision_on_trained_weights.txt)
  const bool if16 = if16_gate(cfg, indexed) && gemm_f16_planes_fits(ws + L.p_xi, R, pb + P.p_cimg_f16, R, ws + L.h_image, D, M, D, R);
  if (b->entity_text_cls) {
    set_error("drin_forward_prepared: entity_text_cls (text pooled ahead of time) is a form of the training entry points");
    return DRIN_E_UNSUPPORTED;
  }
  if (indexed && cfg->num_entities <= 0) {
    set_error("drin_forward_prepared: entity_index given but cfg.num_entities = %d", cfg->num_entities);
    return DRIN_E_SHAPE;
  }
  if (indexed && !planes) {
    // the fp32 contractions read the entity rows in place; with a table they need gathered copies
    set_error("drin_forward_prepared: entity_index needs the split-bf16 precision (the stream kernel gathers the "
              "GEMM operands as planes); gather on the caller side for DRIN_PREC_F32");
    return DRIN_E_UNSUPPORTED;
  }
  const bool bf16_feat = cfg->feature_dtype == DRIN_FEAT_BF16;
  if (bf16_feat && !planes) {
    set_error("drin_forward_prepared: bf16 features need the split-bf16 precision (DRIN_PREC_BF16X3) and D, R multiples of 32");
    return DRIN_E_UNSUPPORTED;
  }
  __bf16* xt_hi = reinterpret_cast<__bf16*>(ws + L.p_xt);
  __bf16* xi_hi = reinterpret_cast<__bf16*>(ws + L.p_xi);
  __bf16* e1_hi = reinterpret_cast<__bf16*>(ws + L.p_et1);
  const size_t MD = (size_t)M * D, MR = (size_t)M * R;
  const drin_layer_params& L1 = params->layer[0];
  const drin_layer_params& L2 = params->layer[1];

  // Mention-sized contractions: the configured precision; in split-bf16 precision problems of >= 256 rows stream
  // the pre-split weight planes by LDS-DMA, smaller ones (latency-bound) stay on the exact fp32 kernel
  auto lin = [&](const float* x, int64_t ldx, const float* w, int64_t ldw, size_t plane_off, size_t plane_elems,
                 const float* bias, float* y, int64_t ldy, int64_t rows, int n_out, int k) -> int {
    if (planes && (rows >= 256 || prec == DRIN_PREC_BF16X3_ALL) && (k % 32) == 0 && (ldw % 8) == 0) {
      const __bf16* hi = reinterpret_cast<const __bf16*>(pb + plane_off);
      return launch_gemm_nt_bf16x3(x, ldx, w, ldw, bias, y, ldy, rows, n_out, k, st, hi, hi + plane_elems, false,
                                   L.splitk_floats ? ws + L.splitk : nullptr, L.splitk_floats);
    }
    return launch_gemm_nt(x, ldx, w, ldw, bias, y, ldy, rows, n_out, k, false, prec, st,
                          L.splitk_floats ? ws + L.splitk : nullptr, L.splitk_floats);
  };
  // Two such products that do not depend on each other, both on the exact-fp32 split-K kernel (small batches): one launch for
  // the two kernels, one for the two slice reductions (gemm_f32.hip: F32GemmGroup; same slices, same order, same bits)
  struct Lin {
    const float* x;
    int64_t ldx;
    const float* w;
    int64_t ldw;
    size_t plane_off, plane_elems;
    const float* bias;
    float* y;
    int64_t ldy, rows;
    int n_out, k;
  };
  auto lin_pair = [&](const Lin& a, const Lin& b) -> int {
    const Lin* two[2] = {&a, &b};
    bool grouped = L.splitk_floats > 0;
    size_t need = 8;
    for (const Lin* q : two) {
      const bool x3 = planes && (q->rows >= 256 || prec == DRIN_PREC_BF16X3_ALL) && (q->k % 32) == 0 && (q->ldw % 8) == 0;
      grouped = grouped && !x3 && gemm_nt_f32_group_fits(q->x, q->ldx, q->w, q->ldw, q->y, q->ldy, q->rows, q->n_out, q->k, prec);
      need += (size_t)8 * q->rows * q->n_out;
    }
    if (grouped && need <= L.splitk_floats) {
      F32GemmGroup g;
      for (const Lin* q : two) DRIN_TRY(g.add_nt(q->x, q->ldx, q->w, q->ldw, q->bias, q->y, q->ldy, q->rows, q->n_out, q->k));
      return launch_gemm_nt_f32_group(g, st, ws + L.splitk, L.splitk_floats);
    }
    for (const Lin* q : two)
      DRIN_TRY(lin(q->x, q->ldx, q->w, q->ldw, q->plane_off, q->plane_elems, q->bias, q->y, q->ldy, q->rows, q->n_out, q->k));
    return DRIN_OK;
  };
  const size_t DD = (size_t)D * D, DR = (size_t)D * R;
  float* vm0 = ws + L.vm0;
  float* hmfu = ws + L.hmfu;
  // pooled entity text stored as bf16 is exact in its hi plane: no lo plane, two MFMAs per tile pair
  const bool xt_exact = bf16_feat && !tokens;
  // (1) mention-side pooling (ghmfc.py:54-60, model.py:41) and vertex-encoder Linears
  if (bf16_feat) {
    DRIN_TRY(launch_span_mean_bf16(b->mention_text, b->mention_start, b->mention_end, ws + L.span_mean, B,
                                   cfg->mention_tokens, D, st));
    DRIN_TRY(launch_axis_mean_bf16(b->mention_image, ws + L.mimg, B, cfg->image_regions, R, st));
  } else {
    DRIN_TRY(launch_span_mean(b->mention_text, b->mention_start, b->mention_end, ws + L.span_mean, B,
                              cfg->mention_tokens, D, st));
    DRIN_TRY(launch_axis_mean(b->mention_image, ws + L.mimg, B, cfg->image_regions, R, st));
  }
  // mention-sized contractions take the configured precision too: launch_gemm_nt keeps problems of fewer
  // than 1024 rows on the fp32 kernel (latency-bound), larger ones (WikiDiverse batches) go split-bf16
  DRIN_TRY(lin_pair({ws + L.span_mean, D, params->w_mention_text, D, P.p_wmt, DD, params->b_mention_text, vm0, D, B, D, D},
                    {ws + L.mimg, R, params->w_mention_image, R, P.p_wmi, DR, params->b_mention_image, vm0 + (size_t)B * D, D, B, D, R}));
  // (2) [hm | fu] = [mt0; mi0] [W_h1; W_u1]^T + [0; b_u1], then q = fu [W_v1 W_et | W_v1 W_ei]
  DRIN_TRY(lin(vm0, D, pb + P.wcat1, D, P.p_wcat1, 2 * DD, pb + P.bcat1, hmfu, 2 * D, 2 * (int64_t)B, 2 * D, D));
  if (dyn)
    DRIN_TRY(lin(hmfu + D, 2 * D, pb + P.ecat, D, P.p_ecat, DD + DR, nullptr, ws + L.q, D + R, 2 * (int64_t)B, D + R, D));
  // (3) one pass over the entity-side bytes
  StreamArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.entity_text = b->entity_text;
  sa.entity_mask = b->entity_text_mask;
  sa.entity_image = b->entity_image;
  sa.entity_object = b->entity_object;
  sa.entity_object_score = b->entity_object_score;
  sa.entity_index = b->entity_index;
  sa.num_entities = cfg->num_entities;
  sa.index_status = b->index_status;
  sa.miet = b->miet_similarity;
  sa.mtei = b->mtei_similarity;
  sa.span_mean = ws + L.span_mean;
  sa.mobj = b->mention_object;
  sa.mscore = b->mention_object_score;
  sa.fu = hmfu + D;
  sa.ldfu = 2 * D;
  sa.q = ws + L.q;
  sa.ldq = D + R;  // q row = [q_text (D) | q_image (R)]
  sa.k_t = pb + P.k_t;
  sa.k_i = pb + P.k_i;
  sa.xt_out = planes ? nullptr : ws + L.xt;
  if (planes) {
    sa.xt_hi = xt_hi;
    sa.xt_lo = xt_exact ? nullptr : xt_hi + MD;
    if (xi_planes) {
      sa.xi_hi = xi_hi;
      sa.xi_lo = bf16_feat ? nullptr : xi_hi + MR;  // bf16 image rows are their own hi plane: nothing left for lo
    }
  }
  sa.xi_scale = if16 ? ws + L.xi_scale : nullptr;
  sa.xi_f16 = if16 ? static_cast<void*>(xi_hi) : nullptr;
  sa.e0m = ws + L.e0m;
  sa.e1m = ws + L.e1m;
  sa.s_part = ws + L.s_part;
  sa.B = B;
  sa.N = N;
  sa.D4 = D / 4;
  sa.R4 = R / 4;
  sa.T = cfg->entity_tokens;
  sa.Km = cfg->mention_objects;
  sa.Ke = cfg->entity_objects;
  sa.chunks = L.chunks;
  sa.dynamic = dyn ? 1 : 0;
  sa.act_e = edge_act(cfg);
  sa.bf16_features = bf16_feat ? 1 : 0;
  for (int k = 0; k < 4; ++k) sa.mask[k] = cfg->edge_enabled[k];
  sa.cos_eps = cfg->cosine_eps;
  sa.miei_eps = cfg->miei_eps;
  sa.clip = cfg->clip_scale;
  sa.s_text = ws + L.s_text;
  sa.s_img = ws + L.s_img;
  sa.sig = ws + L.sig;
  DRIN_TRY(launch_entity_stream(sa, st));
  if (L.chunks > 1)
    DRIN_TRY(launch_reduce_stream_partials(ws + L.s_part, ws + L.s_text, ws + L.s_img, ws + L.sig, B, D, R, L.chunks, st));
  // (4) layer-1 mention vertices: T = S_text W_et^T + S_img W_ei^T, then the W_h input, W_h, LN, GELU
  DRIN_TRY(lin_pair({ws + L.s_text, D, params->w_entity_text, D, P.p_wet, DD, nullptr, ws + L.tm, D, 2 * (int64_t)B, D, D},
                    {ws + L.s_img, R, params->w_entity_image, R, P.p_wei, DR, nullptr, ws + L.tm2, D, 2 * (int64_t)B, D, R}));
  DRIN_TRY(launch_mention_input1(ws + L.tm, ws + L.tm2, ws + L.sig, params->b_entity_text, params->b_entity_image, vm0, ws + L.agg1, B, D, N, st));
  float* vm1 = ws + L.vm1;
  DRIN_TRY(lin(ws + L.agg1, D, L1.w_h, D, P.p_wh1, DD, L1.b_h, vm1, D, 2 * (int64_t)B, D, D));
  DRIN_TRY(launch_layernorm_gelu(vm1, L1.ln_weight, L1.ln_bias, vm1, nullptr, nullptr, 2 * (int64_t)B, D, cfg->layer_norm_eps, st,
                                 vertex_act(cfg)));
  DRIN_TRY(lin(vm1, D, L2.w_h, D, P.p_wh2, DD, nullptr, ws + L.hm2, D, 2 * (int64_t)B, D, D));
  // (5) the two pair-sized layer-1 contractions on the folded weights
  // split-K scratch: the whole-product split when the batch is a few tiles, else the tail-split scratch
  float* const psk = L.pair_splitk_floats ? ws + L.pair_splitk : (L.splitk_floats ? ws + L.splitk : nullptr);
  const size_t pskf = L.pair_splitk_floats ? L.pair_splitk_floats : L.splitk_floats;
  if (planes) {
    const __bf16* ct = reinterpret_cast<const __bf16*>(pb + P.p_ctxt);
    const __bf16* ci = reinterpret_cast<const __bf16*>(pb + P.p_cimg);
    DRIN_TRY(launch_gemm_x3_planes(xt_hi, xt_exact ? nullptr : xt_hi + MD, D, ct, ct + (size_t)D * D, D, nullptr, ws + L.h_text, D, M, D,
                                   D, st, psk, pskf));
    if (if16)        // one fp16 pass on the plane the stream kernel wrote
      DRIN_TRY(launch_gemm_f16_planes(xi_hi, R, pb + P.p_cimg_f16, R, ws + L.xi_scale, pb + P.cimg_f16_scale, ws + L.h_image, D, M, D, R,
                                      st, psk, pskf));
    else if (xi_planes)
      DRIN_TRY(launch_gemm_x3_planes(xi_hi, bf16_feat ? nullptr : xi_hi + MR, R, ci, ci + (size_t)D * R, R, nullptr, ws + L.h_image, D, M,
                                     D, R, st, psk, pskf));
    else if (bf16_feat)  // the bf16 image rows are read in place as the (only) plane of the A operand
      DRIN_TRY(launch_gemm_x3_planes(b->entity_image, nullptr, R, ci, ci + (size_t)D * R, R, nullptr, ws + L.h_image, D, M, D, R, st, psk,
                                     pskf));
    else
      DRIN_TRY(launch_gemm_nt_bf16x3(b->entity_image, R, pb + P.c_img, R, nullptr, ws + L.h_image, D, M, D, R, st, ci, ci + (size_t)D * R,
                                     false, psk, pskf));
  } else {
    const float* x_t = tokens ? ws + L.xt : b->entity_text;
    DRIN_TRY(launch_gemm_nt(x_t, D, pb + P.c_txt, D, nullptr, ws + L.h_text, D, M, D, D, false, prec, st));
    DRIN_TRY(launch_gemm_nt(b->entity_image, R, pb + P.c_img, R, nullptr, ws + L.h_image, D, M, D, R, false, prec, st));
  }
  // The two pair kernels walk the same candidate chunks as the stream kernel (FusedLayout::chunks: whole mentions from 2 048
  // mentions up - their share of that change: row kernels 1.385 -> 1.27 ms at 4 096 mentions; 64 mentions with whole-mention
  // workgroups 0.55 -> 0.63 ms, which is why small calls keep 16 candidates).  DRIN_PAIR_CHUNK = candidates per workgroup (probes).
  int pair_chunks = L.chunks;
  {
    static const char* pc_env = getenv("DRIN_PAIR_CHUNK");
    const int per = pc_env ? atoi(pc_env) : 0;
    if (per >= 16) pair_chunks = std::max(1, std::min(L.chunks, (int)cdiv(N, per)));
  }
  // (6) layer-1 entity vertices + layer-2 mention aggregates
  PairArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.h_text = ws + L.h_text;
  pa.h_image = ws + L.h_image;
  pa.hm = hmfu;
  pa.ldhm = 2 * D;
  pa.c_t = pb + P.cb_t;
  pa.c_i = pb + P.cb_i;
  pa.gamma = L1.ln_weight;
  pa.beta = L1.ln_bias;
  pa.e0m = ws + L.e0m;
  pa.e1m = ws + L.e1m;
  pa.et1 = planes ? nullptr : ws + L.et1;
  if (planes) {
    pa.et1_hi = e1_hi;
    pa.et1_lo = e1_hi + MD;
  }
  pa.s2_part = ws + L.s2_part;
  pa.B = B;
  pa.N = N;
  pa.D4 = D / 4;
  pa.chunks = pair_chunks;
  pa.ln_eps = cfg->layer_norm_eps;
  pa.act_v = vertex_act(cfg);
  DRIN_TRY(launch_pair_layer1(pa, st));
  // (7) layer-2 mention-text vertex
  DRIN_TRY(launch_mention_input2(ws + L.s2_part, vm1, ws + L.agg2, B, D, N, pair_chunks, st));
  DRIN_TRY(lin(ws + L.agg2, D, L2.w_h, D, P.p_wh2, DD, L2.b_h, ws + L.mt2, D, B, D, D));
  DRIN_TRY(launch_layernorm_gelu(ws + L.mt2, L2.ln_weight, L2.ln_bias, ws + L.mt2, nullptr, nullptr, B, D, cfg->layer_norm_eps, st,
                                 vertex_act(cfg)));
  // (8) layer-2 entity-text contraction, vertex and score
  float* h2 = ws + L.h_text;
  // (Round 6, built and dropped - profiles/r6_full_row_ab.txt, tools/probes/r6_full_row/, commit 8a5bc6b: this contraction and k_pair_final's
  //  arithmetic in ONE launch on 96-row x 768-column tiles, so that h2 never goes to HBM.  Correct, and slower: a full-row tile needs all
  //  768 weight rows per K-step for 96 activation rows - 1.56 x the LDS-DMA instructions per MFMA of the 256 x 256 tile - and its K-loop
  //  alone took what the 256 x 256 kernel takes including its store (1.20 against 1.15 ms); the epilogue's vector work came on top.)
  if (planes) {
    const __bf16* w2 = reinterpret_cast<const __bf16*>(pb + P.p_wh2);
    DRIN_TRY(launch_gemm_x3_planes(e1_hi, e1_hi + MD, D, w2, w2 + (size_t)D * D, D, nullptr, h2, D, M, D, D, st, psk, pskf));
  } else {
    DRIN_TRY(launch_gemm_nt(ws + L.et1, D, L2.w_h, D, nullptr, h2, D, M, D, D, false, prec, st));
  }
  FinalArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.h2 = h2;
  fa.hm2 = ws + L.hm2;
  fa.b_h2 = L2.b_h;
  fa.gamma = L2.ln_weight;
  fa.beta = L2.ln_bias;
  fa.e1m = ws + L.e1m;
  fa.mt2 = ws + L.mt2;
  fa.scores = scores;
  fa.B = B;
  fa.N = N;
  fa.D4 = D / 4;
  fa.chunks = pair_chunks;
  fa.ln_eps = cfg->layer_norm_eps;
  fa.act_v = vertex_act(cfg);
  fa.cos_eps = cfg->cosine_eps;
  return launch_pair_final(fa, st);
}

}  // namespace drin

extern "C" {

int32_t drin_workgroups_per_mention(const drin_config* cfg, int32_t cached) {
  if (validate_config(cfg) != DRIN_OK || fused_supported(cfg) != DRIN_OK) return 0;
  if (cached) return cached_chunks_per_mention(*cfg);
  FusedLayout L;
  L.build(*cfg);
  return L.chunks;
}

int32_t drin_image_contraction_passes(const drin_config* cfg, int32_t indexed) {
  if (validate_config(cfg) != DRIN_OK || fused_supported(cfg) != DRIN_OK) return 0;
  if (cfg->precision == DRIN_PREC_F32) return 0;
  if (if16_gate(cfg, indexed != 0)) return 1;
  return cfg->feature_dtype == DRIN_FEAT_BF16 ? 2 : 3;
}

int drin_index_status(int32_t* index_status, void* stream) {
  if (!index_status) {
    set_error("drin_index_status: NULL argument");
    return DRIN_E_NULL;
  }
  DRIN_BIND_DEVICE(stream, index_status, "drin_index_status");
  int32_t w[4] = {0, 0, 0, 0};
  hipError_t e = hipMemcpyAsync(w, index_status, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "drin_index_status");
  if (w[0] == 0) return DRIN_OK;
  e = hipMemsetAsync(index_status, 0, sizeof(w), (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "drin_index_status(reset)");
  const long long value = (long long)(((uint64_t)(uint32_t)w[3] << 32) | (uint32_t)w[2]);
  set_error("entity_index out of range: candidate row %lld at pair %d (b * N + n) is outside the entity tables - the row was clamped "
            "and a WRONG entity scored; drin/data.py:87-93 raises IndexError here", value, (int)w[1]);
  return DRIN_E_INDEX;
}

size_t drin_fused_workspace_bytes(const drin_config* cfg) {
  if (validate_config(cfg) != DRIN_OK || fused_supported(cfg) != DRIN_OK) return 0;
  FusedLayout L;
  L.build(*cfg);
  return L.total * sizeof(float);
}

int drin_forward_prepared(const drin_config* cfg, const drin_batch* b, const drin_params* params, const void* prepared,
                          void* workspace, size_t workspace_bytes, float* scores, void* stream) {
  DRIN_BIND_DEVICE(stream, workspace, "drin_forward_prepared");
  RoctxRange range("drin_forward_prepared");
  DRIN_TRY(validate_config(cfg));
  DRIN_TRY(fused_supported(cfg));
  if (!b || !params || !prepared || !workspace || !scores) {
    set_error("drin_forward_prepared: NULL argument");
    return DRIN_E_NULL;
  }
  FusedLayout L;
  L.build(*cfg);
  if (workspace_bytes < L.total * sizeof(float) || !aligned16(workspace)) {
    set_error("drin_forward_prepared: workspace has %zu bytes (needs %zu) or is not 16-byte aligned", workspace_bytes,
              L.total * sizeof(float));
    return DRIN_E_WORKSPACE;
  }
  return forward_prepared_on_stream(cfg, b, params, prepared, workspace, scores, (hipStream_t)stream);
}

}  // extern "C"
