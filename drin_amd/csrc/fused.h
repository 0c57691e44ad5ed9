// Argument blocks and launch wrappers of the fused two-layer inference pipeline.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/drin_hip.h"

namespace drin {

// Layout of the caller-owned buffer drin_prepare fills (weight-only products, offsets in floats).
struct Prepared {  // offsets in floats
  size_t wcat1, bcat1, ecat, etmp, k_t, k_i, c_txt, c_img, cb_t, cb_i, total;
  size_t p_ctxt, p_cimg, p_wh2;  // bf16 (hi, lo) planes of the three pair-sized GEMM weights; lo follows hi
  size_t p_wmt, p_wmi, p_wcat1, p_ecat, p_wet, p_wei, p_wh1;  // the same for the mention-sized GEMM weights
  size_t p_cimg_f16;             // W_h1 W_ei as ONE fp16 plane [D, R] under one power-of-two scale (DRIN_PREC_BF16X3_IF16)
  size_t cimg_f16_scale;         // [2]: that scale, and the scratch word of its reduction
  void build(const drin_config& c) {
    const size_t D = c.embed_dim, R = c.image_dim;
    size_t off = 0;
    auto take = [&off](size_t n) {
      const size_t o = off;
      off += (n + 63) & ~(size_t)63;
      return o;
    };
    wcat1 = take(2 * D * D);      // [W_h1; W_u1]           [2D, D]
    bcat1 = take(2 * D);          // [0; b_u1]
    ecat = take(D * (D + R));     // ([W_v1 W_et | W_v1 W_ei])^T  [D + R, D]  (nn.Linear layout: q = fu ecat^T)
    etmp = take(D * (D + R));     // scratch of drin_prepare: the un-transposed product
    k_t = take(D);                // W_v1 b_et + b_v1
    k_i = take(D);                // W_v1 b_ei + b_v1
    c_txt = take(D * D);          // W_h1 W_et              [D, D]
    c_img = take(D * R);          // W_h1 W_ei              [D, R]
    cb_t = take(D);               // W_h1 b_et + b_h1
    cb_i = take(D);               // W_h1 b_ei + b_h1
    p_ctxt = take(D * D);         // hi plane D*D bf16 (= D*D/2 floats) then lo plane
    p_cimg = take(D * R);
    p_wh2 = take(D * D);
    p_wmt = take(D * D);          // W_mt
    p_wmi = take(D * R);          // W_mi
    p_wcat1 = take(2 * D * D);    // [W_h1; W_u1]
    p_ecat = take(D * (D + R));   // ([W_v1 W_et | W_v1 W_ei])^T
    p_wet = take(D * D);          // W_et
    p_wei = take(D * R);          // W_ei
    p_wh1 = take(D * D);          // W_h1
    p_cimg_f16 = take(D * R / 2 + 2);   // D R halves
    cimg_f16_scale = take(2);
    total = off;
  }
};

// DRIN_OK when `c` can take the folded two-layer pipelines (fused_forward.hip, entity_cache.hip)
int fused_supported(const drin_config* c);
// workgroups (candidate chunks) per mention of the per-entity-cache path's kernels for this call (entity_cache.hip)
int cached_chunks_per_mention(const drin_config& c);

struct StreamArgs {
  // batch (entity side) - device pointers into the caller's tensors
  const void* entity_text;           // TOKENS: [M, T, D]   else [M, D]     (fp32, or bf16 when bf16_features)
  const int64_t* entity_mask;        // TOKENS: [M, T]
  const void* entity_image;          // [M, R]
  const void* entity_object;         // [M, Ke, R]
  const float* entity_object_score;  // [M, Ke]
  const int64_t* entity_index;       // optional [M]: entity_* above are tables, pair p reads row entity_index[p]
  int64_t num_entities;
  int32_t* index_status;             // optional int32[4]: where a clamped entity_index is reported (drin_batch.index_status)
  const float* miet;                 // [M]
  const float* mtei;                 // [M]
  // mention side
  const float* span_mean;            // [B, D]
  const void* mobj;                  // [B, Km, R]   (fp32 / bf16 like the entity features)
  const float* mscore;               // [B, Km]
  const float* fu;                   // rows b and B + b, row stride ldfu: W_u(mt0), W_u(mi0)  (dynamic edges only)
  const float* q;                    // [2 B][ldq]  fu * [W_v W_et | W_v W_ei]     (dynamic edges only)
  const float* k_t;                  // [D]  W_v b_et + b_v
  const float* k_i;                  // [D]  W_v b_ei + b_v
  // outputs
  float* xt_out;                     // TOKENS: pooled entity text [M, D] (NULL when only the planes are wanted)
  void* xt_hi;                       // optional bf16 hi / lo planes of the entity text GEMM operand [M, D]
  void* xt_lo;
  void* xi_hi;                       // optional bf16 hi / lo planes of the entity image rows [M, R]
  void* xi_lo;
  float* xi_scale;                   // optional [M]: 2^ceil(log2 max |image row|) per pair (1 for an all-zero row) - DRIN_PREC_BF16X3_IF16 -
  void* xi_f16;                      //   and with it [M, R] fp16: the image row divided by that scale (the one plane of its contraction)
  float* e0m;                        // [4][M] layer-1 edges (already multiplied by the edge switch)
  float* e1m;                        // [4][M] layer-2 edges (ditto)
  float* s_part;                     // [B][chunks][2 D + 2 R + 4]
  float* s_text;                     // chunks == 1 only: [2][B][D] (S_tt, S_it), [2][B][R] (S_ti, S_ii), [4][B] edge sums,
  float* s_img;                      //   written directly in the layout k_reduce_stream_partials would produce
  float* sig;
  int B, N, D4, R4, T, Km, Ke, chunks, ldq, ldfu, dynamic, bf16_features;
  int act_e;                         // drin_activation of the edges, resolved (DRIN_ACT_SIGMOID by default)
  float mask[4];
  float cos_eps, miei_eps, clip;
};

struct PairArgs {
  const float* h_text;   // [M, D]  X_t (W_h W_et)^T
  const float* h_image;  // [M, D]  X_i (W_h W_ei)^T
  const float* hm;       // rows b and B + b, row stride ldhm: W_h mt0, W_h mi0 (no bias)
  const float* c_t;      // [D]  W_h b_et + b_h
  const float* c_i;      // [D]  W_h b_ei + b_h
  const float* gamma;
  const float* beta;
  const float* e0m;
  const float* e1m;
  float* et1;            // [M, D]  (NULL when only the planes are wanted)
  void* et1_hi;          // optional bf16 hi / lo planes of et1
  void* et1_lo;
  float* s2_part;        // [B][chunks][2 D]
  int B, N, D4, chunks, ldhm;
  int act_v;             // drin_activation of the vertices, resolved (DRIN_ACT_GELU by default)
  float ln_eps;
};

struct FinalArgs {
  const float* h2;       // [M, D]  et1 W_h2^T
  const float* hm2;      // [2][B][D]  W_h2 mt1, W_h2 mi1 (no bias)
  const float* b_h2;
  const float* gamma;
  const float* beta;
  const float* e1m;
  const float* mt2;      // [B, D]
  float* scores;         // [M]
  int B, N, D4, chunks;
  int act_v;             // drin_activation of the vertices, resolved
  float ln_eps, cos_eps;
};

size_t entity_stream_lds_bytes(const StreamArgs& a);
int launch_entity_stream(const StreamArgs& a, hipStream_t st);
int launch_reduce_stream_partials(const float* part, float* s_text, float* s_img, float* sig, int B, int D, int R,
                                  int chunks, hipStream_t st);
int launch_mention_input1(const float* T, const float* T2, const float* sig, const float* b_et, const float* b_ei,
                          const float* v0, float* out, int B, int D, int N, hipStream_t st);
int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t st);
int launch_pair_layer1(const PairArgs& a, hipStream_t st);
int launch_mention_input2(const float* part, const float* mt1, float* out, int B, int D, int N, int chunks,
                          hipStream_t st);
int launch_pair_final(const FinalArgs& a, hipStream_t st);
}  // namespace drin
