// Workspace layout of one forward (+ backward) pass.  All offsets are in floats from the 16-byte
// aligned workspace base the caller owns; every region starts on a 256-byte boundary.
//
// Inference (training == false) re-uses the per-layer scratch regions across layers and ping-pongs
// the vertex buffers; training keeps every layer's intermediates because backward reads them.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/drin_hip.h"

namespace drin {

struct Pooled {
  const float* span_mean = nullptr;      // [B, D]   mean of the mention span (ghmfc.py:54-60)
  const float* mention_image = nullptr;  // [B, R]   mean over regions (model.py:41)
  const float* mention_object = nullptr; // [B, Km, R]
  const float* entity_object = nullptr;  // [B, N, Ke, R]
  const float* entity_image = nullptr;   // [B, N, R]
  const float* entity_text = nullptr;    // [B, N, D] pooled (ghmfc.py:245-249) or as delivered
  int64_t entity_text_raw_stride = 0;    // row stride of the raw CLS / pooler vector (model.py:73-75)
};

struct Layout {
  bool training = false;
  size_t total_floats = 0;
  size_t span_mean = 0, mimg_pool = 0, mobj_pool = 0, eobj_pool = 0, eimg_pool = 0, xet_pool = 0;
  size_t edges[DRIN_MAX_LAYERS + 1] = {};   // [4][M]  tt, ti, it, ii
  size_t masked[DRIN_MAX_LAYERS] = {};      // [4][M]  edges * gcn_edge_enabled (model.py:122)
  size_t vm[DRIN_MAX_LAYERS + 1] = {};      // [2][B][D]  mention text, mention image vertex
  size_t ve[DRIN_MAX_LAYERS + 1] = {};      // [2][M][D]  entity text, entity image vertex
  size_t agg_m[DRIN_MAX_LAYERS] = {};       // [2][B][D]  W_h input of the mention vertices
  size_t agg_e[DRIN_MAX_LAYERS] = {};       // [2][M][D]  W_h input of the entity vertices
  size_t h_m[DRIN_MAX_LAYERS] = {};         // [2][B][D]  pre-LayerNorm
  size_t h_e[DRIN_MAX_LAYERS] = {};         // [2][M][D]
  size_t ln_stat_m[DRIN_MAX_LAYERS] = {};   // [2][2][B]  mean, rstd   (training only)
  size_t ln_stat_e[DRIN_MAX_LAYERS] = {};   // [2][2][M]
  size_t fu[DRIN_MAX_LAYERS] = {};          // [2][B][D]  W_u(mt), W_u(mi)
  size_t fv[DRIN_MAX_LAYERS] = {};          // [2][M][D]  W_v(et), W_v(ei)
  size_t edges_scalar = 0;                  // [4][M]     scalar static edges before model.py:202 expands them (vector edges)
  size_t pre[DRIN_MAX_LAYERS] = {};         // [4][M][D]  W_m input cat(fu, fv) + e (vector edges)
  size_t edge_z[DRIN_MAX_LAYERS] = {};      // [4][M](x D) pre-activation of the edge update, kept for backward with gelu / silu edges
  size_t splitk = 0, splitk_floats = 0;     // split-K partials of the mention-sized exact-fp32 products (small batches only)
  size_t tn_part = 0, tn_part_floats = 0;   // [slices][N][K] partial tiles of the split-bf16 weight-gradient products
  size_t small_part = 0, small_part_floats = 0;     // slices of the mention-sized exact-fp32 weight-gradient products (launch_gemm_tn_f32_group)
  size_t colsum_part = 0, colsum_part_floats = 0;   // [8][kColsumMaxSlices][D] partial rows of the bias column sums
  size_t ln_part = 0;                       // [1024 + 16][3][D] per-block column sums of the LayerNorm backward and their first reduction level
  size_t wt = 0;                            // [2 layers + 1][D][D]: bf16 (hi, lo) planes of W_h^T / W_v^T of every layer (split-bf16 dX = dY W, one batched transpose + split per backward) + one fp32 slot for products transposed on the fly
  // bf16 (hi, lo) planes of the weights the split-bf16 NT products run against (one batched split per forward call):
  // vertex encoders (mention text, mention image, entity text, entity image), per layer W_h, W_u, W_v.  numel floats each.
  size_t wp_enc[4] = {};
  size_t wp_h[DRIN_MAX_LAYERS] = {}, wp_u[DRIN_MAX_LAYERS] = {}, wp_v[DRIN_MAX_LAYERS] = {};
  bool weight_planes = false;
  size_t bwd_scratch = 0;                   // backward temporaries (training only)
  size_t bwd_scratch_floats = 0;

  size_t take(size_t n) {
    const size_t o = total_floats;
    total_floats += (n + 63) & ~(size_t)63;
    return o;
  }

  void build(const drin_config& c, bool train) {
    training = train;
    total_floats = 0;
    const size_t B = (size_t)c.batch, N = (size_t)c.num_candidates, D = (size_t)c.embed_dim, R = (size_t)c.image_dim;
    const size_t M = B * N;
    const int nl = c.num_layers;
    const size_t EW = c.vector_edges ? D : 1;  // floats per edge per pair
    edges_scalar = take(c.vector_edges ? 4 * M : 0);
    span_mean = take(B * D);
    mimg_pool = take(B * R);
    mobj_pool = take(c.mention_object_inner > 1 ? B * c.mention_objects * R : 0);
    eobj_pool = take(c.entity_object_inner > 1 ? M * c.entity_objects * R : 0);
    eimg_pool = take(c.entity_image_inner > 1 ? M * R : 0);
    xet_pool = take(c.entity_tokens > 0 ? M * D : 0);
    splitk_floats = 2 * B <= 512 ? 16 * 2 * B * D + 64 : 0;   // 8 slices of two [2 B][D] products (grouped launches, gemm_f32.hip)
    splitk = take(splitk_floats);
    // split-bf16 precision with scalar edges and at least a tile row of pairs: the weights as planes
    weight_planes = (c.precision == DRIN_PREC_BF16X3 || c.precision == DRIN_PREC_BF16X3_ALL) && !c.vector_edges && M >= 256 &&
                    (D % 32) == 0 && (R % 32) == 0;
    if (weight_planes) {
      wp_enc[0] = take(D * D), wp_enc[1] = take(D * R), wp_enc[2] = take(D * D), wp_enc[3] = take(D * R);
      for (int l = 0; l < nl; ++l) wp_h[l] = take(D * D), wp_u[l] = take(D * D), wp_v[l] = take(D * D);
    }
    if (train) {
      for (int l = 0; l <= nl; ++l) {
        edges[l] = take(4 * M * EW);
        vm[l] = take(2 * B * D);
        ve[l] = take(2 * M * D);
      }
      for (int l = 0; l < nl; ++l) {
        masked[l] = take(4 * M * EW);
        pre[l] = take(c.vector_edges ? 4 * M * D : 0);
        edge_z[l] = take((c.edge_activation == DRIN_ACT_GELU || c.edge_activation == DRIN_ACT_SILU) && c.dynamic_edges ? 4 * M * EW : 0);
        agg_m[l] = take(2 * B * D);
        agg_e[l] = take(2 * M * D);
        h_m[l] = take(2 * B * D);
        h_e[l] = take(2 * M * D);
        ln_stat_m[l] = take(4 * B);
        ln_stat_e[l] = take(4 * M);
        fu[l] = take(2 * B * D);
        fv[l] = take(2 * M * D);
      }
      // backward temporaries: gradients w.r.t. two generations of vertices/edges + GEMM operands
      wt = take(((size_t)2 * nl + 1) * D * D);   // slots 2 l, 2 l + 1: W_h^T, W_v^T of layer l; slot 2 nl: the product in flight
      ln_part = take((size_t)(1024 + 16 + 16 * (nl > 0 ? nl : 1)) * 3 * D);   // + one level-1 area per layer (deferred second level)
      {  // one workgroup per CU over the whole group of weight-gradient products (launch_gemm_tn_group): at most 256 partial
        // tiles of 256 x 256 - or, where one product alone has more tiles than that, one slice of each product
        const size_t one_slice = ((size_t)2 * nl + 1) * D * D + D * R;
        tn_part_floats = ((size_t)256 * 65536 > one_slice ? (size_t)256 * 65536 : one_slice) + (size_t)8 * 256 * D;  // + column sums
        tn_part = take(tn_part_floats);
      }
      {  // every weight-gradient product of at most 2048 reduction rows may store up to four slices (internal.h:
        // small_tn_slices); the mention side has 2 B rows at most, the entity side 2 M
        // (a side's products have `rows` or 2 x `rows` reduction rows - dW_h[top] and the vertex encoders see one vertex type,
        //  the others both - and the ones past 2048 rows leave the group: each side is sized for whichever of its two row
        //  counts stores more.  Sized at 2 x rows alone - ADVICE r3 - a side with rows <= 2048 < 2 rows got NO scratch while
        //  its single-type products still stored four slices: DRIN_E_WORKSPACE in exact fp32 at 1024 < B N <= 2048.)
        auto slices = [](size_t rows) { return rows > 2048 ? (size_t)0 : (rows + 127) / 128 > 4 ? (size_t)4 : (rows + 127) / 128 < 1 ? (size_t)1 : (rows + 127) / 128; };
        auto side = [&slices](size_t rows) { return slices(rows) > slices(2 * rows) ? slices(rows) : slices(2 * rows); };
        const size_t per_side = ((size_t)2 * nl + 1) * D * D + D * R;   // dW_h, dW_u | dW_v of every layer, a text and an image encoder
        small_part_floats = (side(B) + side(M)) * per_side + 64;
        small_part = take(small_part_floats);
        colsum_part_floats = (size_t)8 * 64 * D;
        colsum_part = take(colsum_part_floats);
      }
      // backward temporaries: vertex gradients per level (nl + 1) + dA + dfv / dfu per layer, entity and mention side; edges x3
      bwd_scratch_floats = ((size_t)2 * nl + 2) * (2 * M * D + 64 + 2 * B * D + 64) + 3 * (4 * M * EW + 64) + (3 * M + 64);
      bwd_scratch = take(bwd_scratch_floats);
    } else {
      size_t e2[2] = {take(4 * M * EW), take(4 * M * EW)};
      size_t vm2[2] = {take(2 * B * D), take(2 * B * D)};
      size_t ve2[2] = {take(2 * M * D), take(2 * M * D)};
      for (int l = 0; l <= nl; ++l) {
        edges[l] = e2[l & 1];
        vm[l] = vm2[l & 1];
        ve[l] = ve2[l & 1];
      }
      const size_t s_masked = take(4 * M * EW), s_agg_m = take(2 * B * D), s_agg_e = take(2 * M * D);
      const size_t s_fu = take(2 * B * D);
      const size_t s_pre = take(c.vector_edges ? 4 * M * D : 0);
      for (int l = 0; l < nl; ++l) {
        masked[l] = s_masked;
        pre[l] = s_pre;
        agg_m[l] = s_agg_m;
        agg_e[l] = s_agg_e;
        // pre-LN values are written straight into the next vertex buffer and normalised in place
        h_m[l] = vm[l + 1];
        h_e[l] = ve[l + 1];
        fu[l] = s_fu;
        // W_v(et), W_v(ei) are consumed by the edge update before the next layer touches agg_e
        fv[l] = s_agg_e;
      }
    }
  }
};

int validate_config(const drin_config* c);
int vertex_act(const drin_config* c);   // drin_activation with DRIN_ACT_DEFAULT resolved (gelu)
int edge_act(const drin_config* c);     // (sigmoid)
void resolve_pooled(const drin_config* c, const drin_batch* b, const Layout& L, float* ws, Pooled* out);
int run_pooling(const drin_config* c, const drin_batch* b, const Layout& L, float* ws, Pooled* out, hipStream_t st);
int run_static_edges(const drin_config* c, const drin_batch* b, const Pooled& P, float* edges, hipStream_t st);

}  // namespace drin
