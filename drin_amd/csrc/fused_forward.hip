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
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "fused.h"
#include "internal.h"
#include "layout.h"

namespace drin {

// side-by-side phases (below): off until measured; DRIN_PIPE / drin_set_pipeline choose
constexpr int kPipeDefaultStreamCus = 0, kPipeDefaultChunkPairs = 51712;
// DRIN_PREC_BF16X3_I1: the shortest candidate list whose mean averages the one-pass image contraction's noise far enough
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
    const bool planes = c.precision == DRIN_PREC_BF16X3 || c.precision == DRIN_PREC_BF16X3_ALL || c.precision == DRIN_PREC_BF16 ||
                        c.precision == DRIN_PREC_BF16X3_I1 || c.precision == DRIN_PREC_BF16X3_IF16;
    p_xt = take(planes ? M * D : 0);   // hi plane (M*D bf16) then lo plane
    p_xi = take(planes ? M * R : 0);   // used by the table form only
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
  if (!x || !hi || !lo) {
    set_error("drin_split_planes: NULL argument");
    return DRIN_E_NULL;
  }
  return launch_split_planes(x, hi, lo, n, (hipStream_t)stream);
}

int drin_linear_planes_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                           float* y, int64_t rows, int32_t n_out, int32_t k, void* stream) {
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
    DRIN_TRY(launch_to_f16(pb + P.c_img, pb + P.p_cimg_f16, (int64_t)dr, st));   // DRIN_PREC_BF16X3_IF16: one fp16 plane
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

// The call in two halves.  kPhaseStream: everything up to and including the one pass over the entity bytes (HBM-bound);
// kPhaseContract: the contractions and row kernels behind it (MFMA-bound).  Together: the whole forward on one stream.
enum { kPhaseStream = 1, kPhaseContract = 2, kPhaseAll = 3 };

static int forward_prepared_phases(const drin_config* cfg, const drin_batch* b, const drin_params* params, const void* prepared,
                                   void* workspace, float* scores, hipStream_t st, const int phases) {
  FusedLayout L;
  L.build(*cfg);
  Prepared P;
  P.build(*cfg);
  float* ws = (float*)workspace;
  const float* pb = (const float*)prepared;
  const int B = cfg->batch, N = cfg->num_candidates, D = cfg->embed_dim, R = cfg->image_dim;
  const int64_t M = (int64_t)B * N;
  if (B == 0) return DRIN_OK;
  // DRIN_PREC_BF16: the three pair-sized contractions run one bf16 MFMA pass (operands rounded to bf16, no lo planes);
  // everything mention-sized keeps the split-bf16 / exact-fp32 arithmetic of DRIN_PREC_BF16X3
  const bool one_pass = cfg->precision == DRIN_PREC_BF16;
  // DRIN_PREC_BF16X3_I1, precision by contraction: only x_i C_i^T runs one pass - its result, the layer-1 entity image
  // vertex, reaches the score through a mean over the N candidates alone (model.py:124-129,143-144), which averages the
  // rounding noise down: 1.7-2.5e-5 on the scores at N = 101 against 3e-4 for either D x D contraction (oracle/
  // precision_emulation.py; tests/test_gpu_round4.py).  At N = 11 the averaging is sqrt(11): 5-10e-5, no margin under the
  // 1e-4 bar - lists shorter than kMixedMinCandidates keep three passes (then the mode IS split-bf16, bit for bit).
  const bool i1 = cfg->precision == DRIN_PREC_BF16X3_I1 && cfg->num_candidates >= kMixedMinCandidates;
  const int prec = (one_pass || cfg->precision == DRIN_PREC_BF16X3_I1 || cfg->precision == DRIN_PREC_BF16X3_IF16) ? (int)DRIN_PREC_BF16X3
                                                                                                                  : cfg->precision;
  const bool dyn = cfg->dynamic_edges != 0;
  const bool tokens = cfg->entity_tokens > 0;
  // split-bf16 precision: the producers write bf16 hi / lo planes and the three pair-sized contractions
  // run on the LDS-DMA kernel of gemm_x3_planes.hip
  const bool planes = (prec == DRIN_PREC_BF16X3 || prec == DRIN_PREC_BF16X3_ALL) && (D % 32 == 0) && (R % 32 == 0);
  const bool indexed = b->entity_index != nullptr;
  // The image rows are read in place by their contraction (fp32, split on the fly): writing 8 KB/pair of
  // planes from the stream kernel costs it more (measured twice on one box: +1.0 ms at B = 4096) than the LDS-DMA
  // kernel gains on that GEMM (-0.35 ms).  Only the table form, which has to gather the rows anyway, writes them.
  static const char* xi_env = getenv("DRIN_XI_PLANES");   // probe switch: image rows as planes in the gathered form too
  const bool xi_planes = planes && (indexed || (xi_env != nullptr && xi_env[0] == '1' && cfg->feature_dtype == DRIN_FEAT_F32));
  // DRIN_PREC_BF16X3_IF16: x_i C_i^T in one FP16 pass, image rows scaled by a power of two each (k_entity_stream hands the
  // scales over).  For the per-pair fp32 image rows of a call that fills whole 256 x 256 grids; everything else: three passes.
  // (the candidate-count gate of DRIN_PREC_BF16X3_I1 holds for this mode too: with freshly initialised weights the fp16 pass costs
  //  8e-6 at N = 11, but once the weights are TRAINED the vertex -> score map steepens and 11 candidates average too little - 1.2e-4
  //  after 200 Adam steps, outside the bar; at N = 101 the same weights give 2e-5: profiles/r4_precision_on_trained_weights.txt)
  const bool if16 = cfg->precision == DRIN_PREC_BF16X3_IF16 && planes && !xi_planes && cfg->feature_dtype == DRIN_FEAT_F32 &&
                    cfg->num_candidates >= kMixedMinCandidates &&
                    D == 768 && R == 2048 &&   // (the stream kernel's row-scale hand-over is an instantiation of the exact widths)
                    gemm_nt_f16_scaled_fits(b->entity_image, R, pb + P.p_cimg_f16, R, ws + L.h_image, D, M, D, R);
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
      return launch_gemm_nt_bf16x3(x, ldx, w, ldw, bias, y, ldy, rows, n_out, k, st, hi, hi + plane_elems, false, false,
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
  if (phases & kPhaseStream) {
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
    sa.xt_lo = (xt_exact || one_pass) ? nullptr : xt_hi + MD;
    if (xi_planes) {
      sa.xi_hi = xi_hi;
      sa.xi_lo = (bf16_feat || one_pass || i1) ? nullptr : xi_hi + MR;  // bf16 image rows are their own hi plane: nothing left for lo
    }
  }
  sa.xi_scale = if16 ? ws + L.xi_scale : nullptr;
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
  }
  if (!(phases & kPhaseContract)) return DRIN_OK;
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
    DRIN_TRY(launch_gemm_x3_planes(xt_hi, (xt_exact || one_pass) ? nullptr : xt_hi + MD, D, ct,
                                   one_pass ? nullptr : ct + (size_t)D * D, D, nullptr, ws + L.h_text, D, M, D, D, st, psk, pskf));
    const bool xi_one = one_pass || i1;   // x_i C_i^T in one pass
    if (xi_planes)
      DRIN_TRY(launch_gemm_x3_planes(xi_hi, (bf16_feat || xi_one) ? nullptr : xi_hi + MR, R, ci,
                                     xi_one ? nullptr : ci + (size_t)D * R, R, nullptr, ws + L.h_image, D, M, D, R, st, psk, pskf));
    else if (bf16_feat)  // the bf16 image rows are read in place as the (only) plane of the A operand
      DRIN_TRY(launch_gemm_x3_planes(b->entity_image, nullptr, R, ci, xi_one ? nullptr : ci + (size_t)D * R, R, nullptr,
                                     ws + L.h_image, D, M, D, R, st, psk, pskf));
    else if (if16)
      DRIN_TRY(launch_gemm_nt_f16_scaled(b->entity_image, R, pb + P.p_cimg_f16, R, ws + L.xi_scale, ws + L.h_image, D, M, D, R, st));
    else
      DRIN_TRY(launch_gemm_nt_bf16x3(b->entity_image, R, pb + P.c_img, R, nullptr, ws + L.h_image, D, M, D, R, st, ci,
                                     ci + (size_t)D * R, false, xi_one && cdiv(M, 256) * cdiv(D, 256) >= 192,
                                     psk, pskf));
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
    pa.et1_lo = one_pass ? nullptr : e1_hi + MD;
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
  if (planes) {
    const __bf16* w2 = reinterpret_cast<const __bf16*>(pb + P.p_wh2);
    DRIN_TRY(launch_gemm_x3_planes(e1_hi, one_pass ? nullptr : e1_hi + MD, D, w2, one_pass ? nullptr : w2 + (size_t)D * D, D,
                                   nullptr, h2, D, M, D, D, st, psk, pskf));
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

// ---- the two phases side by side on disjoint CU sets ------------------------------------------------------------------
// k_entity_stream is HBM-bound, the contractions MFMA-bound; on the same CUs they do not overlap (the GEMM's tile loads queue
// behind the stream kernel's ~100 KB per CU in flight), on DISJOINT CU sets they do, and a pass over 3 KB rows needs far
// fewer than 256 CUs to keep HBM busy (tools/probes/partition_probe.hip: 96 CUs 4.6 TB/s, 128 CUs 5.6, 160 CUs 6.2 = what
// all 256 reach).  A large batch is therefore cut into chunks of mentions; chunk c + 1's stream phase runs on the first
// `stream_cus` CUs while chunk c's contraction phase runs on the others (two streams created with
// hipExtStreamCreateWithCUMask, forked from and joined to the caller's stream by events).  Each chunk has its own workspace,
// so no buffer is shared between chunks in flight.  A mention's score depends on its chunk's size only through the tile /
// split-K choices of the mention-sized products (all inside the parity bar, and a deterministic function of the batch size).
struct PipePlan {
  int chunks = 1, chunk_mentions = 0, stream_cus = 0;
};

static std::atomic<int> g_pipe_stream_cus{-1}, g_pipe_chunk_pairs{-1};   // -1: environment / default

static PipePlan pipe_plan(const drin_config& c) {
  PipePlan p;
  p.chunk_mentions = c.batch;
  int cus = g_pipe_stream_cus.load(std::memory_order_relaxed);
  int chunk_pairs = g_pipe_chunk_pairs.load(std::memory_order_relaxed);
  if (cus < 0 || chunk_pairs < 0) {
    // DRIN_PIPE = "off" | "<stream CUs>" | "<stream CUs>:<pairs per chunk>"
    static const char* env = getenv("DRIN_PIPE");
    int e_cus = kPipeDefaultStreamCus, e_pairs = kPipeDefaultChunkPairs;
    if (env != nullptr && env[0] != 0) {
      if (env[0] == 'o') e_cus = 0;
      else {
        e_cus = atoi(env);
        const char* colon = strchr(env, ':');
        if (colon) e_pairs = atoi(colon + 1);
      }
    }
    if (cus < 0) cus = e_cus;
    if (chunk_pairs < 0) chunk_pairs = e_pairs;
  }
  const bool planes = (c.precision == DRIN_PREC_BF16X3 || c.precision == DRIN_PREC_BF16X3_ALL || c.precision == DRIN_PREC_BF16X3_I1 ||
                       c.precision == DRIN_PREC_BF16X3_IF16) &&
                      c.embed_dim % 32 == 0 && c.image_dim % 32 == 0;
  if (cus <= 0 || cus >= 256 || chunk_pairs < 4096 || !planes) return p;
  int per = (int)std::max<int64_t>(1, chunk_pairs / c.num_candidates);
  if (per >= 256) per -= per % 256;   // whole 256-row tiles of the mention-sized products
  const int n = (int)cdiv(c.batch, per);
  if (n < 4) return p;                // fill and drain of a short pipeline cost what the overlap buys
  p.chunks = n;
  p.chunk_mentions = per;
  p.stream_cus = cus;
  return p;
}

struct PipeStreams {
  hipStream_t a = nullptr, b = nullptr;
  hipEvent_t fork = nullptr, join_a = nullptr, join_b = nullptr;
  std::vector<hipEvent_t> done;
  int cus = 0;
};
static std::mutex g_pipe_mutex;
static PipeStreams g_pipe[16][2];   // per device: the configured split, and one spare for a second setting (probes)

static int pipe_streams(int stream_cus, int chunks, PipeStreams** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev < 0 || dev >= 16) {
    set_error("pipelined forward: device %d outside the 16 per-device stream slots", dev);
    return DRIN_E_UNSUPPORTED;
  }
  std::lock_guard<std::mutex> lock(g_pipe_mutex);
  PipeStreams* ps = nullptr;
  for (PipeStreams& q : g_pipe[dev])
    if (q.cus == stream_cus || q.cus == 0) {
      ps = &q;
      break;
    }
  if (!ps) {  // a third setting in one process: rebuild the spare
    ps = &g_pipe[dev][1];
    (void)hipStreamSynchronize(ps->a);
    (void)hipStreamSynchronize(ps->b);
    (void)hipStreamDestroy(ps->a);
    (void)hipStreamDestroy(ps->b);
    ps->a = ps->b = nullptr;
    ps->cus = 0;
  }
  if (ps->cus == 0) {
    int total = 0;
    hipError_t e = hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return hip_fail(e, "hipDeviceGetAttribute(multiprocessor count)");
    if (total > 256 || stream_cus >= total) {
      set_error("pipelined forward: %d stream CUs of %d", stream_cus, total);
      return DRIN_E_UNSUPPORTED;
    }
    // mask bit i is CU i / 8 of XCD i % 8 (the driver deals the bits round-robin over the XCDs): a prefix of the bits
    // is spread evenly over the eight XCDs and their L2s
    uint32_t ma[8] = {0}, mb[8] = {0};
    for (int c = 0; c < total; ++c) (c < stream_cus ? ma : mb)[c / 32] |= 1u << (c % 32);
    e = hipExtStreamCreateWithCUMask(&ps->a, 8, ma);
    if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ps->b, 8, mb);
    if (e != hipSuccess) return hip_fail(e, "hipExtStreamCreateWithCUMask");
    for (hipEvent_t* ev : {&ps->fork, &ps->join_a, &ps->join_b})
      if (*ev == nullptr && (e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess)
        return hip_fail(e, "hipEventCreate(pipeline)");
    ps->cus = stream_cus;
  }
  while ((int)ps->done.size() < chunks) {
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return hip_fail(e, "hipEventCreate(pipeline)");
    ps->done.push_back(ev);
  }
  *out = ps;
  return DRIN_OK;
}

// mentions [m0, m0 + count) of a batch
static drin_batch slice_batch(const drin_config& c, const drin_batch& b, int64_t m0) {
  const size_t es = c.feature_dtype == DRIN_FEAT_BF16 ? 2 : 4;
  auto feat = [&](const float* p, int64_t elems) -> const float* {
    return p ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + (size_t)elems * es) : nullptr;
  };
  auto inner = [](int v) { return (int64_t)(v > 1 ? v : 1); };
  const int64_t N = c.num_candidates, D = c.embed_dim, R = c.image_dim, T = c.entity_tokens;
  drin_batch s = b;
  s.mention_text = feat(b.mention_text, m0 * c.mention_tokens * D);
  s.mention_start = b.mention_start ? b.mention_start + m0 : nullptr;
  s.mention_end = b.mention_end ? b.mention_end + m0 : nullptr;
  s.mention_image = feat(b.mention_image, m0 * c.image_regions * R);
  s.mention_object = feat(b.mention_object, m0 * c.mention_objects * inner(c.mention_object_inner) * R);
  s.mention_object_score = b.mention_object_score ? b.mention_object_score + m0 * c.mention_objects : nullptr;
  s.miet_similarity = b.miet_similarity ? b.miet_similarity + m0 * N : nullptr;
  s.mtei_similarity = b.mtei_similarity ? b.mtei_similarity + m0 * N : nullptr;
  if (b.entity_index) {   // the entity tensors are tables: only the index moves
    s.entity_index = b.entity_index + m0 * N;
    return s;
  }
  s.entity_text = feat(b.entity_text, T > 0 ? m0 * N * T * D : m0 * N * D);
  s.entity_text_mask = (b.entity_text_mask && T > 0) ? b.entity_text_mask + m0 * N * T : b.entity_text_mask;
  s.entity_image = feat(b.entity_image, m0 * N * inner(c.entity_image_inner) * R);
  s.entity_object = feat(b.entity_object, m0 * N * c.entity_objects * inner(c.entity_object_inner) * R);
  s.entity_object_score = b.entity_object_score ? b.entity_object_score + m0 * N * c.entity_objects : nullptr;
  return s;
}

static size_t chunk_workspace_floats(const drin_config& c, const PipePlan& plan) {
  drin_config cc = c;
  cc.batch = plan.chunk_mentions;
  FusedLayout L;
  L.build(cc);
  return L.total;
}

static int forward_prepared_pipelined(const drin_config* cfg, const drin_batch* b, const drin_params* params, const void* prepared,
                                      void* workspace, float* scores, hipStream_t st, const PipePlan& plan) {
  PipeStreams* ps = nullptr;
  DRIN_TRY(pipe_streams(plan.stream_cus, plan.chunks, &ps));
  const size_t chunk_floats = chunk_workspace_floats(*cfg, plan);
  hipError_t e = hipEventRecord(ps->fork, st);
  if (e == hipSuccess) e = hipStreamWaitEvent(ps->a, ps->fork, 0);
  if (e == hipSuccess) e = hipStreamWaitEvent(ps->b, ps->fork, 0);
  if (e != hipSuccess) return hip_fail(e, "pipeline fork");
  int rc = DRIN_OK;
  for (int c = 0; c < plan.chunks && rc == DRIN_OK; ++c) {
    const int64_t m0 = (int64_t)c * plan.chunk_mentions;
    drin_config cc = *cfg;
    cc.batch = (int)std::min<int64_t>(plan.chunk_mentions, cfg->batch - m0);
    const drin_batch sb = slice_batch(*cfg, *b, m0);
    float* ws = (float*)workspace + (size_t)c * chunk_floats;
    float* sc = scores + m0 * cfg->num_candidates;
    rc = forward_prepared_phases(&cc, &sb, params, prepared, ws, sc, ps->a, kPhaseStream);
    if (rc != DRIN_OK) break;
    e = hipEventRecord(ps->done[c], ps->a);
    if (e == hipSuccess) e = hipStreamWaitEvent(ps->b, ps->done[c], 0);
    if (e != hipSuccess) {
      rc = hip_fail(e, "pipeline hand-over");
      break;
    }
    rc = forward_prepared_phases(&cc, &sb, params, prepared, ws, sc, ps->b, kPhaseContract);
  }
  // join in every case: the caller's stream never runs ahead of work already queued on the side streams
  e = hipEventRecord(ps->join_a, ps->a);
  if (e == hipSuccess) e = hipEventRecord(ps->join_b, ps->b);
  if (e == hipSuccess) e = hipStreamWaitEvent(st, ps->join_a, 0);
  if (e == hipSuccess) e = hipStreamWaitEvent(st, ps->join_b, 0);
  if (e != hipSuccess && rc == DRIN_OK) rc = hip_fail(e, "pipeline join");
  return rc;
}

}  // namespace drin

extern "C" {

int drin_set_pipeline(int32_t stream_cus, int32_t chunk_pairs) {
  if (stream_cus >= 256) {
    set_error("drin_set_pipeline: stream_cus=%d (0 = one stream, -1 = default, else < 256)", stream_cus);
    return DRIN_E_SHAPE;
  }
  g_pipe_stream_cus.store(stream_cus < 0 ? -1 : stream_cus);
  g_pipe_chunk_pairs.store(chunk_pairs < 0 ? -1 : chunk_pairs);
  return DRIN_OK;
}

int drin_set_weight_gradient_passes(int32_t passes) {
  if (passes != 1 && passes != 3 && passes >= 0) {
    set_error("drin_set_weight_gradient_passes: %d (1 = one bf16 pass, 3 = the split product, -1 = default)", passes);
    return DRIN_E_SHAPE;
  }
  set_weight_gradient_passes(passes < 0 ? -1 : passes);
  return DRIN_OK;
}

int32_t drin_workgroups_per_mention(const drin_config* cfg, int32_t cached) {
  if (validate_config(cfg) != DRIN_OK || fused_supported(cfg) != DRIN_OK) return 0;
  if (cached) return cached_chunks_per_mention(*cfg);
  FusedLayout L;
  L.build(*cfg);
  return L.chunks;
}

size_t drin_fused_workspace_bytes(const drin_config* cfg) {
  if (validate_config(cfg) != DRIN_OK || fused_supported(cfg) != DRIN_OK) return 0;
  const PipePlan plan = pipe_plan(*cfg);
  if (plan.chunks > 1) return (size_t)plan.chunks * chunk_workspace_floats(*cfg, plan) * sizeof(float);
  FusedLayout L;
  L.build(*cfg);
  return L.total * sizeof(float);
}

int drin_forward_prepared(const drin_config* cfg, const drin_batch* b, const drin_params* params, const void* prepared,
                          void* workspace, size_t workspace_bytes, float* scores, void* stream) {
  RoctxRange range("drin_forward_prepared");
  DRIN_TRY(validate_config(cfg));
  DRIN_TRY(fused_supported(cfg));
  if (!b || !params || !prepared || !workspace || !scores) {
    set_error("drin_forward_prepared: NULL argument");
    return DRIN_E_NULL;
  }
  hipStream_t st = (hipStream_t)stream;
  PipePlan plan = pipe_plan(*cfg);
  if (plan.chunks > 1) {   // a stream being captured into a graph stays on the one-stream schedule
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) plan = PipePlan();
  }
  FusedLayout L;
  L.build(*cfg);
  const size_t need = plan.chunks > 1 ? (size_t)plan.chunks * chunk_workspace_floats(*cfg, plan) : L.total;
  if (workspace_bytes < need * sizeof(float) || !aligned16(workspace)) {
    set_error("drin_forward_prepared: workspace has %zu bytes (needs %zu) or is not 16-byte aligned", workspace_bytes,
              need * sizeof(float));
    return DRIN_E_WORKSPACE;
  }
  if (plan.chunks > 1) return forward_prepared_pipelined(cfg, b, params, prepared, workspace, scores, st, plan);
  return forward_prepared_phases(cfg, b, params, prepared, workspace, scores, st, kPhaseAll);
}

}  // extern "C"
