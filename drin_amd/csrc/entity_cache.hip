// Per-entity precompute cache for table-form inference: drin_build_entity_cache /
// drin_forward_cached.
//
// With frozen weights everything the first GCN layer takes from an ENTITY is a function of that entity alone,
// not of the (mention, candidate) pair.  One row per entity of a device-resident table holds
//
//   h_t  = x_t (W_h1 W_et)^T                 [D]   layer-1 entity-text contraction   (model.py:128,146)
//   h_i  = x_i (W_h1 W_ei)^T                 [D]   layer-1 entity-image contraction
//   fv_t = W_v1 (W_et x_t + b_et) + b_v1     [D]   edge-update operand W_v(et0)      (model.py:148-153)
//   fv_i = W_v1 (W_ei x_i + b_ei) + b_v1     [D]
//   c^   = cls / max(|cls|, eps)             [D]   CLS / pooler row, normalised      (model.py:71-76)
//   o^   = sum_j es_j obj_j / max(|obj_j|, eps)   [R]   score-weighted normalised objects (model.py:84-92)
//   sg   = sum_j es_j                        [1]   (+ 3 pad)
//
// (x_t = pooled entity text, ghmfc.py:245-249; x_i = image row.)  The image-image edge factorises because the
// cosine is bilinear in the normalised rows:  sum_ij cos(m_i, o_j) ms_i es_j = (sum_i ms_i m^_i) . o^ .
// A pair then needs no W_et / W_ei / W_v contraction at all: per pair 4.33 of the 5.51 MFLOP of the folded
// path disappear, and so does the pass over the token-level entity text.  What remains per pair is ONE
// gathered read of the cache row (k_cached_pairs: edges, both layer-1 entity vertices, every cross-candidate
// sum of both layers), the layer-2 contraction et' W_h2^T and k_pair_final.
//
// The cache is a function of the weights: it is rebuilt (one pass over the table + four table-sized GEMMs)
// whenever they change, so it serves inference / evaluation with frozen weights - never training.
//
// Row formats (drin_cache_format).  DRIN_CACHE_F32, floats:  h_t 0 | h_i D | fv_t 2D | fv_i 3D | c^ 4D | o^ 5D | sg 5D+R (+3 pad).
// DRIN_CACHE_MIXED_F16 - precision by storage: the three contraction results / direction rows that carry a vertex (h_t, h_i)
// or the text-text edge (c^) stay fp32; the operands of per-pair SCALARS that only enter a sigmoid or a weight - fv_t, fv_i
// (mean_d(.) inside the edge sigmoid, model.py:148-153) and o^ (the image-image edge, model.py:84-92, which feeds mi' and ei'
// alone) - are fp16, each with ONE power-of-two scale per row (2^ceil(log2 max|.|): exact to apply, so the row may have any
// magnitude; fp16 keeps 11 bits and a D- or R-long dot averages its rounding).  Floats:
//   h_t 0 | h_i D | c^ 2D [fp32] | fv 3D [per float4 column c4 one 16-byte unit: fv_t[4 c4 .. +3], fv_i[4 c4 .. +3]] |
//   o^ 4D [R f16, natural order: read in the PAIR layout of row_ops.h] | tail 4D + R/2: sg, s_fvt, s_fvi, s_o
// 16 400 B per entity instead of 23 568 at D = 768, R = 2 048.  k_cached_pairs at config 5 is bound by the gathered
// row bytes down to ~12 ms per 4 096 x 1 001 chunk (tools/cached_floor_probe.sh: 19.2-19.7 ms on the fp32 rows, 11.7 ms
// with every row L2-resident), so the smaller row is worth its conversions.  Emulated cost on the scores
// (oracle/precision_emulation.py::scores_with_rounded_cache_fields): 2e-7 at N = 101, 6e-7 at N = 11 - below the split-bf16
// contractions' own 1.3e-6.  (h_i as fp16 too - 14 880 B - was built and measured first: 139-140 M pairs/s against 133-135,
// 2.3e-6 emulated on a homogeneous table.  Not kept: h_i is a VERTEX operand - the cached path forms the layer-1 mention
// aggregates from sum_n e h_i - so its 2^-12 rounding is averaged only as long as the candidates' rows are of one size, and
// ten times the error for 4 % of the chunk time is the wrong trade on a 1e-4 bar: profiles/r4_mixed_cache_ab.txt.)
#include <string.h>

#include <stdlib.h>

#include "fused.h"
#include "internal.h"
#include "layout.h"
#include "row_ops.h"

namespace drin {

static inline bool cache_mixed(const drin_config& c) { return c.cache_format == DRIN_CACHE_MIXED_F16; }
static inline size_t cache_row_floats(const drin_config& c) {
  const size_t D = (size_t)c.embed_dim, R = (size_t)c.image_dim;
  return cache_mixed(c) ? 4 * D + R / 2 + 4 : 5 * D + R + 4;
}
constexpr int64_t kCacheSlab = 262144;  // entities per builder slab (pooled-text scratch: slab * D floats)

struct CacheBuildArgs {
  const float* entity_text;          // TOKENS: [E, T, D] else [E, D]
  const int64_t* entity_mask;        // TOKENS: [E, T]
  const float* entity_object;        // [E, Ke, R]
  const float* entity_object_score;  // [E, Ke]
  float* xt;                         // [rows, D] pooled text of this slab (GEMM operand)
  float* cache;                      // row e0 of the slab
  int64_t ldc, rows;
  int D4, R4, T, Ke;
  float cos_eps;
  int mixed;                         // DRIN_CACHE_MIXED_F16 row (c^ at 2 D, o^ as scaled fp16 at 4 D, tail at 4 D + R / 2)
};

// one wave per entity
template <int DV, int RV, bool TOKENS>
__global__ void __launch_bounds__(256) k_entity_cache_rows(const CacheBuildArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= a.rows) return;
  const int D4 = a.D4, R4 = a.R4, D = D4 * 4, R = R4 * 4;
  float* row = a.cache + e * a.ldc;
  Row<DV> xt, cls;
  if (TOKENS) {  // ghmfc.py:245-249: mean of tokens 1 .. ntok-2; model.py:73-75: token 0 feeds the cosine
    const int T = a.T;
    int cnt = 0;
    for (int t = lane; t < T; t += 64) cnt += (int)a.entity_mask[e * T + t];
    cnt = (int)wave_sum((float)cnt);
    int stop = cnt - 1;
    if (stop < 0) stop += T;
    stop = stop < 0 ? 0 : (stop > T ? T : stop);
    const float* base = a.entity_text + e * (int64_t)T * D;
    cls = load_row<DV>(base, lane, D4);
    Row<DV> acc = zero_row<DV>();
    int t = 1;
    for (; t + 4 <= stop; t += 4) {  // same association as k_entity_stream: the two paths pool bit-identically
      const Row<DV> r0 = load_row<DV>(base + (int64_t)t * D, lane, D4);
      const Row<DV> r1 = load_row<DV>(base + (int64_t)(t + 1) * D, lane, D4);
      const Row<DV> r2 = load_row<DV>(base + (int64_t)(t + 2) * D, lane, D4);
      const Row<DV> r3 = load_row<DV>(base + (int64_t)(t + 3) * D, lane, D4);
#pragma unroll
      for (int j = 0; j < DV; ++j) acc.v[j] = (((acc.v[j] + r0.v[j]) + r1.v[j]) + r2.v[j]) + r3.v[j];
    }
    for (; t < stop; ++t) {
      const Row<DV> r0 = load_row<DV>(base + (int64_t)t * D, lane, D4);
#pragma unroll
      for (int j = 0; j < DV; ++j) acc.v[j] = acc.v[j] + r0.v[j];
    }
    const float den = stop > 1 ? (float)(stop - 1) : 0.0f;  // empty slice: 0 / 0 = NaN like the reference
#pragma unroll
    for (int j = 0; j < DV; ++j)
      xt.v[j] = make_float4(acc.v[j].x / den, acc.v[j].y / den, acc.v[j].z / den, acc.v[j].w / den);
  } else {
    xt = load_row<DV>(a.entity_text + e * D, lane, D4);
    cls = xt;
  }
  store_row<DV>(a.xt + e * D, xt, lane, D4);
  {
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(dot_rows<DV>(cls, cls))), a.cos_eps);
#pragma unroll
    for (int j = 0; j < DV; ++j) cls.v[j] = cls.v[j] * inv;
    store_row<DV>(row + (a.mixed ? 2 * D : 4 * D), cls, lane, D4);
  }
  Row<RV> o = zero_row<RV>();
  float sg = 0.f;
  for (int j = 0; j < a.Ke; ++j) {
    const Row<RV> eo = load_row<RV>(a.entity_object + (e * a.Ke + j) * R, lane, R4);
    const float es = a.entity_object_score[e * a.Ke + j];
    axpy_row<RV>(o, es / fmaxf(sqrtf(wave_sum(dot_rows<RV>(eo, eo))), a.cos_eps), eo);
    sg += es;
  }
  if (a.mixed) {  // o^ as fp16 under one power-of-two scale; tail = sg, s_fvt, s_fvi (k_cache_pack_mixed), s_o
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < RV; ++j)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(o.v[j].x), fabsf(o.v[j].y)), fmaxf(fabsf(o.v[j].z), fabsf(o.v[j].w))));
    const float s_o = cache_field_scale(wave_max(m)), inv = 1.0f / s_o;
    char* dst = reinterpret_cast<char*>(row + 4 * D);
#pragma unroll
    for (int j = 0; j < RV; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < R4) *reinterpret_cast<u32x2_t*>(dst + (size_t)c4 * 8) = float4_to_f16x4(o.v[j] * inv);
    }
    float* tail = row + 4 * D + R / 2;
    if (lane == 0) {
      tail[0] = sg;
      tail[3] = s_o;
    }
    return;
  }
  store_row<RV>(row + 5 * D, o, lane, R4);
  if (lane == 0) st4(row + 5 * D + R, make_float4(sg, 0.f, 0.f, 0.f));
}

// DRIN_CACHE_MIXED_F16: the two table-sized edge-update products fv_t, fv_i ([rows][2 D] fp32 scratch) -> scaled fp16 in the
// cache row.  One wave per entity.
template <int DV>
__global__ void __launch_bounds__(256) k_cache_pack_mixed(const float* __restrict__ tmp, float* __restrict__ cache, int64_t ldc,
                                                          int64_t rows, int D4, int R4, int have_fv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= rows) return;
  const int D = D4 * 4, R = R4 * 4;
  float* row = cache + e * ldc;
  const float* src = tmp + e * 2 * (int64_t)D;
  Row<DV> f[2];
  float sc[2], inv[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    f[k] = have_fv ? load_row<DV>(src + (int64_t)k * D, lane, D4) : zero_row<DV>();
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < DV; ++j)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(f[k].v[j].x), fabsf(f[k].v[j].y)), fmaxf(fabsf(f[k].v[j].z), fabsf(f[k].v[j].w))));
    sc[k] = cache_field_scale(wave_max(m));
    inv[k] = 1.0f / sc[k];
  }
  char* fv = reinterpret_cast<char*>(row + 3 * D);
#pragma unroll
  for (int j = 0; j < DV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < D4) {
      const u32x2_t t = float4_to_f16x4(f[0].v[j] * inv[0]), i = float4_to_f16x4(f[1].v[j] * inv[1]);
      u32x4_t u;
      u[0] = t[0], u[1] = t[1], u[2] = i[0], u[3] = i[1];
      *reinterpret_cast<u32x4_t*>(fv + (size_t)c4 * 16) = u;
    }
  }
  if (lane == 0) {
    float* tail = row + 4 * D + R / 2;
    tail[1] = sc[0];
    tail[2] = sc[1];
  }
}

// ------------------------------------------------------------------------------------------------
struct CachedArgs {
  const float* cache;            // [E][ldc]
  int64_t ldc, num_entities;
  const int64_t* entity_index;   // [M]
  const float* miet;             // [M]
  const float* mtei;             // [M]
  const float* span_mean;        // [B, D]
  const float* mobj;             // [B, Km, R]
  const float* mscore;           // [B, Km]
  const float* fu;               // rows b and B + b, row stride ldfu: W_u1(mt0), W_u1(mi0) incl. bias
  const float* hm;               // rows b and B + b, row stride ldhm: W_h1 mt0, W_h1 mi0 (no bias)
  const float* c_t;              // [D] W_h1 b_et + b_h1
  const float* c_i;              // [D] W_h1 b_ei + b_h1
  const float* gamma;            // layer-1 LayerNorm
  const float* beta;
  float* e1m;                    // [4][M] layer-2 edges (already multiplied by the edge switch)
  float* et1;                    // [M, D] or NULL
  void* et1_hi;                  // bf16 hi / lo planes of et1, or NULL
  void* et1_lo;
  float* c_part;                 // [B][chunks][2 D + 4]: sum e h for the two mention vertices, 4 edge sums
  float* s2_part;                // [B][chunks][2 D]: layer-2 mention aggregates
  int B, N, D4, R4, Km, chunks, ldfu, ldhm, dynamic;
  int act_v, act_e;              // drin_activation of vertices / edges, resolved (gelu / sigmoid by default)
  float mask[4];
  float cos_eps, miei_eps, clip, ln_eps;
};

// k_cached_pairs clamps a candidate row outside the tables (memory safety) but has no register left to REPORT it (256 VGPRs:
// the compare-and-report cost a spill): when the caller gave drin_batch.index_status, this pass over the M indices does -
// 8 bytes per pair against the 16-23 KB of cache row the pair's scoring reads (a 4 096 x 1 001 chunk: ~10 us of 35 ms).
__global__ void __launch_bounds__(256) k_check_entity_index(const int64_t* __restrict__ index, int64_t M, int64_t num_entities,
                                                            int32_t* __restrict__ status) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < M; p += (int64_t)gridDim.x * 256) {
    const int64_t e = index[p];
    if (e < 0 || e >= num_entities) report_bad_index(status, p, e);
  }
}

// grid (chunks, B), 256 threads; wave w takes candidates c0 + w, c0 + w + 4, ... of its chunk
#ifndef DRIN_CACHED_PAIRS_WG_PER_CU
#define DRIN_CACHED_PAIRS_WG_PER_CU 2
#endif
// EXACT: D = 256 DV and R = 256 RV exactly (768 / 2048) - the column guards of the row helpers fold away and the
// loop body becomes straight-line code
// the edge-phase operands of one cache row: 17 KB of the 23.5 KB fp32 row; in the mixed format 10.2 KB - c^ and the raw
// fp16 units of fv_t | fv_i and o^ (widened where they are used: half the registers while the row is in flight) + the scales
template <int DV, int RV, bool MIXED>
struct CachedPairRow;
template <int DV, int RV>
struct CachedPairRow<DV, RV, false> {
  Row<DV> chat, fvt, fvi;
  Row<RV> ohat;
  float sg, mtei, miet;
};
template <int DV, int RV>
struct CachedPairRow<DV, RV, true> {
  Row<DV> chat;
  u32x4_t fv[DV];       // per float4 column: fv_t (two dwords), fv_i (two dwords)
  u32x4_t oh[RV / 2];   // o^ in the PAIR layout: eight consecutive columns per lane and load
  float sg, s_fvt, s_fvi, s_o, mtei, miet;
  // the vertex-phase operands travel a whole step ahead too: the fp16 units leave the registers for it (256 VGPRs, no scratch;
  // same box, alternating: k_cached_pairs 15.0-15.9 -> 14.9 ms and steadier - profiles/r4_mixed_cache_ab.txt 4).  The fp32
  // format's rows request them at the top of their own step: its 255 registers leave no room.
  Row<DV> ht, hi;
};

// MIXED: rows in the DRIN_CACHE_MIXED_F16 format (RV even)
template <int DV, int RV, bool EXACT, bool GENERIC_ACT = false, bool MIXED = false>
__global__ void __launch_bounds__(256, DRIN_CACHED_PAIRS_WG_PER_CU) k_cached_pairs(const CachedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D4 = EXACT ? DV * 64 : a.D4, R4 = EXACT ? RV * 64 : a.R4, D = D4 * 4, R = R4 * 4;
  float* l_s = lds;               // [D]   span mean / max(|.|, eps)
  float* l_mu = l_s + D;          // [R]   sum_i ms_i mobj_i / max(|mobj_i|, eps)
  float* l_fu = l_mu + R;         // [2 D] W_u(mt0), W_u(mi0)
  float* l_const = l_fu + 2 * D;  // [6 D] hm_t, hm_i, c_t, c_i, gamma, beta
  float* l_small = l_const + 6 * D; // [Km] ms_i / |mobj_i| (Km <= 8), then [4][4] per-wave edge sums
  float* l_acc = l_small + 32;    // [4 waves][3 D] A_t, A_i, S2: the cross-candidate sums live in LDS, not in VGPRs -
                                  // the register file is what bounds how much of the NEXT row can be in flight
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = blockIdx.y;
  const int64_t M = (int64_t)a.B * a.N;
  const int per = (a.N + a.chunks - 1) / a.chunks;
  const int n_begin = blockIdx.x * per, n_end = min(a.N, n_begin + per);
  const bool dyn = a.dynamic != 0;

  // Candidate rows of this wave, 4 at a time (candidates base + wave + 4 k of a 16-candidate group): requested as
  // vector loads well before they are needed, then made wave-uniform - as scalars the row base lives in SGPRs and
  // every load of the row shares it.
  auto request_indices = [&](int64_t (&v)[4], int base) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = base + wave + 4 * k;
      v[k] = n < n_end ? a.entity_index[b * a.N + n] : 0;
    }
  };
  auto uniform_indices = [&](int64_t (&dst)[4], const int64_t (&v)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t e = v[k];
      e = e < 0 ? 0 : (e >= a.num_entities ? a.num_entities - 1 : e);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)e), hi = __builtin_amdgcn_readfirstlane((uint32_t)(e >> 32));
      dst[k] = (int64_t)(((uint64_t)hi << 32) | lo);
    }
  };
  int64_t ent[4], ent_v[4];
  request_indices(ent_v, n_begin);  // lands under the prologue

  // ---- per-workgroup prologue ---------------------------------------------------------------------------
  {
    const float* src[6] = {a.hm + b * a.ldhm, a.hm + ((int64_t)a.B + b) * a.ldhm, a.c_t, a.c_i, a.gamma, a.beta};
#pragma unroll
    for (int v = 0; v < 6; ++v)
      for (int i = threadIdx.x; i < D4; i += 256) st4(l_const + v * D + i * 4, ld4(src[v] + i * 4));
    if (dyn)
      for (int i = threadIdx.x; i < 2 * D4; i += 256) {
        const int which = i / D4, c4 = i - which * D4;
        st4(l_fu + i * 4, ld4(a.fu + ((int64_t)which * a.B + b) * a.ldfu + c4 * 4));
      }
  }
  typedef CachedPairRow<DV, RV, MIXED> PairRow;
  auto fetch = [&](PairRow& r, int64_t e, int64_t p) {
    const float* row = a.cache + e * a.ldc;
    if constexpr (MIXED) {
      r.chat = load_row_stream<DV>(row + 2 * D, lane, D4);
      const char* op = reinterpret_cast<const char*>(row + 4 * D);
#pragma unroll
      for (int q = 0; q < RV / 2; ++q) {
        const int c8 = lane + 64 * q;
        if (2 * c8 < R4) r.oh[q] = ld16_stream(op + (size_t)c8 * 16);
        else r.oh[q] = u32x4_t{0u, 0u, 0u, 0u};
      }
      const float* tail = row + 4 * D + R / 2;
      r.sg = tail[0];
      r.s_fvt = tail[1];
      r.s_fvi = tail[2];
      r.s_o = tail[3];
      const char* fp = reinterpret_cast<const char*>(row + 3 * D);
#pragma unroll
      for (int j = 0; j < DV; ++j) {
        const int c4 = lane + 64 * j;
        if (dyn && c4 < D4) r.fv[j] = ld16_stream(fp + (size_t)c4 * 16);
        else r.fv[j] = u32x4_t{0u, 0u, 0u, 0u};
      }
      r.ht = load_row_stream<DV>(row, lane, D4);
      r.hi = load_row_stream<DV>(row + D, lane, D4);
    } else {
      r.chat = load_row_stream<DV>(row + 4 * D, lane, D4);
      r.ohat = load_row_stream<RV>(row + 5 * D, lane, R4);
      r.sg = row[5 * D + R];
      if (dyn) {
        r.fvt = load_row_stream<DV>(row + 2 * D, lane, D4);
        r.fvi = load_row_stream<DV>(row + 3 * D, lane, D4);
      }
    }
    r.mtei = a.mtei[p];
    r.miet = a.miet[p];
  };
  PairRow ra, rb;  // two named buffers (an indexed array of them ends up in scratch memory)
  uniform_indices(ent, ent_v);
  if (n_begin + wave < n_end) fetch(ra, ent[0], b * a.N + n_begin + wave);  // first row: in flight under the rest of the prologue

  for (int i = wave; i < a.Km; i += 4) {  // model.py:88 re-normalises the same mention rows for every pair
    const Row<RV> m = load_row<RV>(a.mobj + (b * a.Km + i) * R, lane, R4);
    const float nrm = fmaxf(sqrtf(wave_sum(dot_rows<RV>(m, m))), a.cos_eps);
    if (lane == 0) l_small[i] = a.mscore[b * a.Km + i] / nrm;
  }
  if (wave == 0) {
    Row<DV> s = load_row<DV>(a.span_mean + b * D, lane, D4);
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(dot_rows<DV>(s, s))), a.cos_eps);
#pragma unroll
    for (int j = 0; j < DV; ++j) s.v[j] = s.v[j] * inv;
    store_row<DV>(l_s, s, lane, D4);
  }
  __syncthreads();
  float sum_ms = 0.f;
  for (int i = 0; i < a.Km; ++i) sum_ms += a.mscore[b * a.Km + i];
  for (int c4 = threadIdx.x; c4 < R4; c4 += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < a.Km; ++i) acc = fma4(l_small[i], ld4(a.mobj + (b * a.Km + i) * R + c4 * 4), acc);
    st4(l_mu + c4 * 4, acc);
  }
  __syncthreads();
  const float *l_hm_t = l_const, *l_hm_i = l_const + D, *l_ct = l_const + 2 * D, *l_ci = l_const + 3 * D;
  const float *l_gamma = l_const + 4 * D, *l_beta = l_const + 5 * D;

  // A_t / A_i: layer-1 aggregates of the mention text / image vertex (already through W_h1);
  // S2 = sum_n (e1_tt et' + e1_ti ei'): the two layer-2 aggregates only ever appear added (model.py:143)
  float* acc = l_acc + wave * 3 * D;
#pragma unroll
  for (int j = 0; j < 3 * DV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < 3 * D4) st4(acc + c4 * 4, make_float4(0.f, 0.f, 0.f, 0.f));
  }
  auto accumulate2 = [&](float* dst, float w1, const Row<DV>& x1, float w2, const Row<DV>& x2) {
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < D4) st4(dst + c4 * 4, fma4(w1, x1.v[j], fma4(w2, x2.v[j], ld4(dst + c4 * 4))));
    }
  };
  float sg_tt = 0.f, sg_ti = 0.f, sg_it = 0.f, sg_ii = 0.f;
  const float inv_d = 1.0f / (float)D;

  // A chunk is one group of at most 16 candidates, 4 per wave.  Per pair the dependent chain is index -> row ->
  // arithmetic; left alone it costs three serial HBM round trips per pair.  So: candidate indices are fetched
  // before the prologue, all loads of a row are issued together, and the edge-phase operands of the NEXT row are
  // requested as soon as the (short) edge phase of the current one has freed their registers - they land
  // under the long LayerNorm / GELU phase.
  auto step = [&](const PairRow& cur, PairRow& nxt, const int n, const int64_t e_cur, const int64_t e_next) {
    if (n >= n_end) return;
    const int64_t p = b * a.N + n;
    const PairRow& r = cur;
    __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting later pairs' loads into this one (spills)
    // the vertex-phase operands of THIS pair: requested now, needed after the edge phase
    Row<DV> ht, hi;
    if constexpr (MIXED) {   // (arrived with the edge-phase operands, one step ahead)
      ht = r.ht;
      hi = r.hi;
    } else {
      ht = load_row_stream<DV>(a.cache + e_cur * a.ldc, lane, D4);
      hi = load_row_stream<DV>(a.cache + e_cur * a.ldc + D, lane, D4);
    }
    // ---- static edges (model.py:71-92, 201-204) ------------------------------------------------------------
    const float e_tt = wave_sum(dot_row_lds<DV>(r.chat, l_s, lane, D4)) * a.mask[0];
    const float e_ti = (r.mtei / a.clip) * a.mask[1];
    const float e_it = (r.miet / a.clip) * a.mask[2];
    float e_ii;
    if constexpr (MIXED) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < RV / 2; ++q) {
        const int c8 = lane + 64 * q;
        if (2 * c8 < R4) {
          s += dot4(f16x4_to_float4(r.oh[q][0], r.oh[q][1]), ld4(l_mu + c8 * 8));
          s += dot4(f16x4_to_float4(r.oh[q][2], r.oh[q][3]), ld4(l_mu + c8 * 8 + 4));
        }
      }
      e_ii = wave_sum(s) * r.s_o / (sum_ms * r.sg + a.miei_eps) * a.mask[3];
    } else {
      e_ii = wave_sum(dot_row_lds<RV>(r.ohat, l_mu, lane, R4)) / (sum_ms * r.sg + a.miei_eps) * a.mask[3];
    }
    // ---- layer-2 edges (model.py:148-153; static: pass-through, model.py:136) ------------------------------
    float n_tt = e_tt, n_ti = e_ti, n_it = e_it, n_ii = e_ii;
    if (dyn) {
      if constexpr (MIXED) {
        float d_tt = 0.f, d_ti = 0.f, d_it = 0.f, d_ii = 0.f;
#pragma unroll
        for (int j = 0; j < DV; ++j) {
          const int c4 = lane + 64 * j;
          if (c4 < D4) {
            const float4 t = f16x4_to_float4(r.fv[j][0], r.fv[j][1]), i = f16x4_to_float4(r.fv[j][2], r.fv[j][3]);
            const float4 u0 = ld4(l_fu + c4 * 4), u1 = ld4(l_fu + D + c4 * 4);
            d_tt += dot4(t, u0);
            d_ti += dot4(i, u0);
            d_it += dot4(t, u1);
            d_ii += dot4(i, u1);
          }
        }
        n_tt = edge_act_apply(a.act_e, wave_sum(d_tt) * r.s_fvt * inv_d + e_tt);
        n_ti = edge_act_apply(a.act_e, wave_sum(d_ti) * r.s_fvi * inv_d + e_ti);
        n_it = edge_act_apply(a.act_e, wave_sum(d_it) * r.s_fvt * inv_d + e_it);
        n_ii = edge_act_apply(a.act_e, wave_sum(d_ii) * r.s_fvi * inv_d + e_ii);
      } else {
        n_tt = edge_act_apply(a.act_e, wave_sum(dot_row_lds<DV>(r.fvt, l_fu, lane, D4)) * inv_d + e_tt);
        n_ti = edge_act_apply(a.act_e, wave_sum(dot_row_lds<DV>(r.fvi, l_fu, lane, D4)) * inv_d + e_ti);
        n_it = edge_act_apply(a.act_e, wave_sum(dot_row_lds<DV>(r.fvt, l_fu + D, lane, D4)) * inv_d + e_it);
        n_ii = edge_act_apply(a.act_e, wave_sum(dot_row_lds<DV>(r.fvi, l_fu + D, lane, D4)) * inv_d + e_ii);
      }
    }
    n_tt *= a.mask[0];
    n_ti *= a.mask[1];
    n_it *= a.mask[2];
    n_ii *= a.mask[3];
    __builtin_amdgcn_sched_barrier(0);
    if (n + 4 < n_end) fetch(nxt, e_next, p + 4);
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) {
      a.e1m[p] = n_tt;
      a.e1m[M + p] = n_ti;
      a.e1m[2 * M + p] = n_it;
      a.e1m[3 * M + p] = n_ii;
    }
    // ---- layer-1 mention aggregates, already through W_h1 (model.py:143-144) -------------------------------
    accumulate2(acc, e_tt, ht, e_ti, hi);
    accumulate2(acc + D, e_it, ht, e_ii, hi);
    sg_tt += e_tt;
    sg_ti += e_ti;
    sg_it += e_it;
    sg_ii += e_ii;
    // ---- layer-1 entity vertices (model.py:128,146) and the layer-2 mention aggregates ----------------------
    const Row<DV> et1 = ln_gelu_row_lds<DV, GENERIC_ACT>(combine_rows_lds<DV>(ht, e_tt, l_hm_t, e_it, l_hm_i, l_ct, lane, D4),
                                                         l_gamma, l_beta, lane, D4, a.ln_eps, a.act_v);
    if (a.et1) store_row<DV>(a.et1 + p * D, et1, lane, D4);
    if (a.et1_hi) store_row_planes<DV>(a.et1_hi, a.et1_lo, p * D, et1, lane, D4);
    const Row<DV> ei1 = ln_gelu_row_lds<DV, GENERIC_ACT>(combine_rows_lds<DV>(hi, e_ti, l_hm_t, e_ii, l_hm_i, l_ci, lane, D4),
                                                         l_gamma, l_beta, lane, D4, a.ln_eps, a.act_v);
    accumulate2(acc + 2 * D, n_tt, et1, n_ti, ei1);
  };
  // a chunk is a whole number of groups of 16 candidates (the last one ragged); the indices of the next group are requested
  // at the top of the current one and its first row during the current one's last step
  // (EXACT widths only: the guarded instantiations have no registers left for the look-ahead and keep ONE group per workgroup)
  for (int g = n_begin; g < n_end; g += 16) {
    const bool more = EXACT && g + 16 < n_end;
    int64_t nxt[4] = {0, 0, 0, 0}, nxt_v[4];
    if (more) request_indices(nxt_v, g + 16);
    const int n0 = g + wave;
    step(ra, rb, n0, ent[0], ent[1]);
    step(rb, ra, n0 + 4, ent[1], ent[2]);
    step(ra, rb, n0 + 8, ent[2], ent[3]);
    if (more) uniform_indices(nxt, nxt_v);
    step(rb, ra, n0 + 12, ent[3], nxt[0]);
    if (!EXACT) break;
#pragma unroll
    for (int k = 0; k < 4; ++k) ent[k] = nxt[k];
  }

  // ---- fixed-order cross-wave reduction, one partial per (mention, chunk) --------------------------------
  if (lane == 0) {
    float* s4 = l_small + 8 + 4 * wave;
    s4[0] = sg_tt;
    s4[1] = sg_ti;
    s4[2] = sg_it;
    s4[3] = sg_ii;
  }
  __syncthreads();
  float* out1 = a.c_part + (b * a.chunks + blockIdx.x) * (int64_t)(2 * D + 4);
  float* out2 = a.s2_part + (b * a.chunks + blockIdx.x) * (int64_t)(2 * D);
  auto waves_sum = [&](int i) {  // float4 column i of the [3 D] accumulators, waves in order
    return ((ld4(l_acc + i * 4) + ld4(l_acc + 3 * D + i * 4)) + ld4(l_acc + 6 * D + i * 4)) + ld4(l_acc + 9 * D + i * 4);
  };
  for (int i = threadIdx.x; i < 2 * D4; i += 256) {
    st4(out1 + i * 4, waves_sum(i));
    // k_mention_input2 adds the two halves of its partial: the merged sum goes in the first, zero in the second
    st4(out2 + i * 4, i < D4 ? waves_sum(2 * D4 + i) : make_float4(0.f, 0.f, 0.f, 0.f));
  }
  if (threadIdx.x < 4) {
    const float* s4 = l_small + 8 + threadIdx.x;
    out1[2 * D + threadIdx.x] = ((s4[0] + s4[4]) + s4[8]) + s4[12];
  }
}

// Pre-LayerNorm layer-1 mention vertices from the chunk partials (model.py:143-144 + :128), W_h1 already applied:
//   out[0][b] = (A_t + sg_tt wb_t + sg_ti wb_i) / N + hm_t[b] + b_h1,  wb = W_h1 b_e = cb - b_h1
//   out[1][b] = (A_i + sg_it wb_t + sg_ii wb_i) / N + hm_i[b] + b_h1
__global__ void __launch_bounds__(256) k_mention_layer1_cached(const float* __restrict__ part, const float* __restrict__ hm,
                                                               int ldhm, const float* __restrict__ cb_t,
                                                               const float* __restrict__ cb_i, const float* __restrict__ b_h,
                                                               float* __restrict__ out, int B, int D, int chunks, float inv_n) {
  const int64_t b = blockIdx.y;
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const int width = 2 * D + 4;
  float at = 0.f, ai = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const float* p = part + (b * chunks + c) * (int64_t)width;
    at += p[d];
    ai += p[D + d];
    s0 += p[2 * D];
    s1 += p[2 * D + 1];
    s2 += p[2 * D + 2];
    s3 += p[2 * D + 3];
  }
  const float bh = b_h[d], wt = cb_t[d] - bh, wi = cb_i[d] - bh;
  out[b * D + d] = (at + s0 * wt + s1 * wi) * inv_n + hm[b * ldhm + d] + bh;
  out[((int64_t)B + b) * D + d] = (ai + s2 * wt + s3 * wi) * inv_n + hm[((int64_t)B + b) * ldhm + d] + bh;
}

// Same box, config 5 (1001 candidates), ms per call at 16 / 64 / 128 candidates per workgroup: 64 mentions 0.72 / 0.70 / 0.72,
// 256 mentions 2.44 / 2.29 / 2.38, 1 024 mentions 9.7 / 9.45 / 9.38, 4 096 mentions 38.0 / 35.3 / 35.1 (k_cached_pairs 21.6 -> 19.5 ms,
// the row kernels behind it 4.16 -> 3.35): profiles/r3_cached_chunk_ab.txt.  (Round 2's "64 candidates: slower" was a version that
// lost the index / row look-ahead at the group boundary.)
static int cached_chunk_default(const drin_config& c) { return c.batch >= 2048 ? 128 : 64; }

// Candidates per workgroup of the cached path's kernels.  16 unless the exact-width kernel runs (see k_cached_pairs) on a
// call large enough that the mentions alone fill the chip; DRIN_CACHED_CHUNK = candidates (multiple of 16) for probes.
static int cached_chunk_candidates(const drin_config& c) {
  const bool exact = c.embed_dim == 768 && c.image_dim == 2048 && vertex_act(&c) == DRIN_ACT_GELU;
  if (!exact) return 16;
  static const char* cc_env = getenv("DRIN_CACHED_CHUNK");
  const int per = cc_env ? atoi(cc_env) : 0;
  if (per >= 16) return per - per % 16;
  return cached_chunk_default(c);
}

struct CachedLayout {  // workspace of drin_forward_cached, offsets in floats
  size_t span_mean, mimg, vm0, hmfu, e1m, c_part, s2_part, vm1, hm2, agg2, mt2, et1, p_et1, h2, splitk, splitk_floats, total;
  int chunks;
  void build(const drin_config& c) {
    const size_t B = c.batch, N = c.num_candidates, D = c.embed_dim, R = c.image_dim, M = B * N;
    // k_cached_pairs walks its chunk in groups of 16 candidates (4 per wave); only the exact-width instantiation (D = 768,
    // R = 2048, default activation) has the registers to look ahead across groups, every other one gets ONE group per
    // workgroup.  cached_chunk_candidates() is the one place that knows.
    chunks = (int)((N + cached_chunk_candidates(c) - 1) / cached_chunk_candidates(c));
    size_t off = 0;
    auto take = [&off](size_t n) {
      const size_t o = off;
      off += (n + 63) & ~(size_t)63;
      return o;
    };
    const bool planes = c.precision == DRIN_PREC_BF16X3 || c.precision == DRIN_PREC_BF16X3_ALL;
    span_mean = take(B * D);
    mimg = take(B * R);
    vm0 = take(2 * B * D);
    hmfu = take(2 * B * 2 * D);
    e1m = take(4 * M);
    c_part = take(B * chunks * (2 * D + 4));
    s2_part = take(B * chunks * 2 * D);
    vm1 = take(2 * B * D);
    hm2 = take(2 * B * D);
    agg2 = take(B * D);
    mt2 = take(B * D);
    et1 = take(planes ? 0 : M * D);
    p_et1 = take(planes ? M * D : 0);  // hi plane (M*D bf16) then lo plane
    h2 = take(M * D);
    splitk_floats = 2 * B <= 512 ? 8 * 2 * B * 2 * D : 0;  // split-K partials of the mention-sized fp32 products
    splitk = take(splitk_floats);
    total = off;
  }
};

int cached_chunks_per_mention(const drin_config& c) {
  CachedLayout L;
  L.build(c);
  return L.chunks;
}

static int cache_supported(const drin_config* c) {
  DRIN_TRY(fused_supported(c));
  if (c->num_entities <= 0) {
    set_error("entity cache: cfg.num_entities = %d (the entity_* tensors must be tables)", c->num_entities);
    return DRIN_E_SHAPE;
  }
  if (c->mention_objects > 8) {
    set_error("entity cache: at most 8 mention objects (got %d)", c->mention_objects);
    return DRIN_E_UNSUPPORTED;
  }
  if (c->feature_dtype != DRIN_FEAT_F32 || c->precision == DRIN_PREC_BF16X3_IF16) {
    set_error("entity cache: bf16 feature storage / DRIN_PREC_BF16X3_IF16 belong to drin_forward_prepared only");
    return DRIN_E_UNSUPPORTED;
  }
  if (cache_mixed(*c) && (c->embed_dim % 8 || c->image_dim % 8)) {
    set_error("entity cache: DRIN_CACHE_MIXED_F16 needs embed_dim %% 8 == 0 and image_dim %% 8 == 0 (got %d, %d)", c->embed_dim,
              c->image_dim);
    return DRIN_E_SHAPE;
  }
  return DRIN_OK;
}

template <int DV, int RV, bool EXACT, bool GENERIC_ACT = false, bool MIXED = false>
static int launch_cached_pairs_t(const CachedArgs& a, hipStream_t st) {
  const size_t D = (size_t)a.D4 * 4, R = (size_t)a.R4 * 4;
  const size_t lds = sizeof(float) * (9 * D + R + 32 + 12 * D);
  auto kern = k_cached_pairs<DV, RV, EXACT, GENERIC_ACT, MIXED>;
  static DynLdsOptIn opt_in;  // one per template instantiation
  if (lds > 48 * 1024)
    DRIN_TRY(ensure_dynamic_lds(opt_in, reinterpret_cast<const void*>(kern), (int)lds, "hipFuncSetAttribute(cached_pairs)"));
  KernelTimer timer(DRIN_KC_STREAM, st);
  hipLaunchKernelGGL(kern, dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), lds, st, a);
  DRIN_CHECK_LAUNCH("k_cached_pairs");
  return DRIN_OK;
}

}  // namespace drin

using namespace drin;

extern "C" {

DRIN_API size_t drin_entity_cache_bytes(const drin_config* cfg) {
  if (validate_config(cfg) != DRIN_OK || cache_supported(cfg) != DRIN_OK) return 0;
  return (size_t)cfg->num_entities * cache_row_floats(*cfg) * sizeof(float);
}

DRIN_API size_t drin_entity_cache_build_workspace_bytes(const drin_config* cfg) {
  if (validate_config(cfg) != DRIN_OK || cache_supported(cfg) != DRIN_OK) return 0;
  const int64_t slab = cfg->num_entities < kCacheSlab ? cfg->num_entities : kCacheSlab;
  // pooled text of a slab; mixed format: + the two table-sized products that are stored as fp16 ([slab][2 D] fp32)
  return (size_t)slab * cfg->embed_dim * sizeof(float) * (cache_mixed(*cfg) ? 3 : 1);
}

DRIN_API size_t drin_cached_workspace_bytes(const drin_config* cfg) {
  if (validate_config(cfg) != DRIN_OK || cache_supported(cfg) != DRIN_OK) return 0;
  CachedLayout L;
  L.build(*cfg);
  return L.total * sizeof(float);
}

DRIN_API int drin_build_entity_cache(const drin_config* cfg, const drin_batch* tables, const drin_params* params,
                                     const void* prepared, void* cache, size_t cache_bytes, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  DRIN_BIND_DEVICE(stream, cache, "drin_build_entity_cache");
  RoctxRange range("drin_build_entity_cache");
  DRIN_TRY(validate_config(cfg));
  DRIN_TRY(cache_supported(cfg));
  if (!tables || !params || !prepared || !cache || !workspace) {
    set_error("drin_build_entity_cache: NULL argument");
    return DRIN_E_NULL;
  }
  if (!tables->entity_text || !tables->entity_image || !tables->entity_object || !tables->entity_object_score ||
      (cfg->entity_tokens > 0 && !tables->entity_text_mask)) {
    set_error("drin_build_entity_cache: NULL entity table");
    return DRIN_E_NULL;
  }
  if (cache_bytes < drin_entity_cache_bytes(cfg) || workspace_bytes < drin_entity_cache_build_workspace_bytes(cfg) ||
      !aligned16(cache) || !aligned16(workspace)) {
    set_error("drin_build_entity_cache: cache %zu / workspace %zu bytes (need %zu / %zu, 16-byte aligned)", cache_bytes,
              workspace_bytes, drin_entity_cache_bytes(cfg), drin_entity_cache_build_workspace_bytes(cfg));
    return DRIN_E_WORKSPACE;
  }
  Prepared P;
  P.build(*cfg);
  hipStream_t st = (hipStream_t)stream;
  const float* pb = (const float*)prepared;
  const int D = cfg->embed_dim, R = cfg->image_dim, T = cfg->entity_tokens;
  const int64_t E = cfg->num_entities;
  const int64_t ldc = (int64_t)cache_row_floats(*cfg);
  const bool dyn = cfg->dynamic_edges != 0;
  const int prec = cfg->precision;
  float* xt = (float*)workspace;
  const bool mixed = cache_mixed(*cfg);
  const int64_t slab_rows = E < kCacheSlab ? E : kCacheSlab;
  float* tmp = xt + slab_rows * (int64_t)D;   // mixed format: [slab][2 D] = fv_t | fv_i
  for (int64_t e0 = 0; e0 < E; e0 += kCacheSlab) {
    const int64_t rows = E - e0 < kCacheSlab ? E - e0 : kCacheSlab;
    float* crow = (float*)cache + e0 * ldc;
    CacheBuildArgs a;
    memset(&a, 0, sizeof(a));
    a.entity_text = tables->entity_text + e0 * (int64_t)(T > 0 ? T : 1) * D;
    a.entity_mask = T > 0 ? tables->entity_text_mask + e0 * T : nullptr;
    a.entity_object = tables->entity_object + e0 * (int64_t)cfg->entity_objects * R;
    a.entity_object_score = tables->entity_object_score + e0 * cfg->entity_objects;
    a.xt = xt;
    a.cache = crow;
    a.ldc = ldc;
    a.rows = rows;
    a.D4 = D / 4;
    a.R4 = R / 4;
    a.T = T;
    a.Ke = cfg->entity_objects;
    a.cos_eps = cfg->cosine_eps;
    a.mixed = mixed ? 1 : 0;
    {
      KernelTimer timer(DRIN_KC_STREAM, st);
      const dim3 grid((unsigned)cdiv(rows, 4)), block(256);
      const bool tiny = a.D4 <= 64 && a.R4 <= 64;
      if (tiny && T > 0)
        hipLaunchKernelGGL((k_entity_cache_rows<1, 1, true>), grid, block, 0, st, a);
      else if (tiny)
        hipLaunchKernelGGL((k_entity_cache_rows<1, 1, false>), grid, block, 0, st, a);
      else if (T > 0)
        hipLaunchKernelGGL((k_entity_cache_rows<3, 8, true>), grid, block, 0, st, a);
      else
        hipLaunchKernelGGL((k_entity_cache_rows<3, 8, false>), grid, block, 0, st, a);
      DRIN_CHECK_LAUNCH("k_entity_cache_rows");
    }
    const float* ximg = tables->entity_image + e0 * (int64_t)R;
    // h_t = x_t C_t^T, h_i = x_i C_i^T
    // (mixed format: h_t, h_i straight into the row, the two fp16 fields through the fp32 scratch and k_cache_pack_mixed)
    float* const y_hi = crow + D;
    float* const y_fvt = mixed ? tmp : crow + 2 * D;
    float* const y_fvi = mixed ? tmp + D : crow + 3 * D;
    const int64_t ldy = mixed ? 2 * (int64_t)D : ldc;
    DRIN_TRY(launch_gemm_nt(xt, D, pb + P.c_txt, D, nullptr, crow, ldc, rows, D, D, false, prec, st));
    DRIN_TRY(launch_gemm_nt(ximg, R, pb + P.c_img, R, nullptr, y_hi, ldc, rows, D, R, false, prec, st));
    if (dyn) {  // fv = x (W_v1 W_e)^T + (W_v1 b_e + b_v1); etmp keeps W_v1 [W_et | W_ei] un-transposed, row stride D + R
      DRIN_TRY(launch_gemm_nt(xt, D, pb + P.etmp, D + R, pb + P.k_t, y_fvt, ldy, rows, D, D, false, prec, st));
      DRIN_TRY(launch_gemm_nt(ximg, R, pb + P.etmp + D, D + R, pb + P.k_i, y_fvi, ldy, rows, D, R, false, prec, st));
    }
    if (mixed) {   // (static edges: zero units, unit scales)
      KernelTimer timer(DRIN_KC_STREAM, st);
      const dim3 grid((unsigned)cdiv(rows, 4)), block(256);
      if (D / 4 <= 64)
        hipLaunchKernelGGL((k_cache_pack_mixed<1>), grid, block, 0, st, tmp, crow, ldc, rows, D / 4, R / 4, dyn ? 1 : 0);
      else
        hipLaunchKernelGGL((k_cache_pack_mixed<3>), grid, block, 0, st, tmp, crow, ldc, rows, D / 4, R / 4, dyn ? 1 : 0);
      DRIN_CHECK_LAUNCH("k_cache_pack_mixed");
    }
  }
  return DRIN_OK;
}

DRIN_API int drin_forward_cached(const drin_config* cfg, const drin_batch* b, const drin_params* params, const void* prepared,
                                 const void* cache, void* workspace, size_t workspace_bytes, float* scores, void* stream) {
  DRIN_BIND_DEVICE(stream, workspace, "drin_forward_cached");
  RoctxRange range("drin_forward_cached");
  DRIN_TRY(validate_config(cfg));
  DRIN_TRY(cache_supported(cfg));
  if (!b || !params || !prepared || !cache || !workspace || !scores) {
    set_error("drin_forward_cached: NULL argument");
    return DRIN_E_NULL;
  }
  if (!b->entity_index || !b->mention_text || !b->mention_start || !b->mention_end || !b->mention_image ||
      !b->mention_object || !b->mention_object_score || !b->miet_similarity || !b->mtei_similarity) {
    set_error("drin_forward_cached: NULL mention tensor / entity_index / similarity");
    return DRIN_E_NULL;
  }
  CachedLayout L;
  L.build(*cfg);
  if (workspace_bytes < L.total * sizeof(float) || !aligned16(workspace) || !aligned16(cache)) {
    set_error("drin_forward_cached: workspace has %zu bytes (needs %zu) or a buffer is not 16-byte aligned",
              workspace_bytes, L.total * sizeof(float));
    return DRIN_E_WORKSPACE;
  }
  Prepared P;
  P.build(*cfg);
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const float* pb = (const float*)prepared;
  const int B = cfg->batch, N = cfg->num_candidates, D = cfg->embed_dim, R = cfg->image_dim;
  const int64_t M = (int64_t)B * N;
  if (B == 0) return DRIN_OK;
  if (B > 65535) {
    set_error("drin_forward_cached: batch %d exceeds the grid limit; split the batch", B);
    return DRIN_E_SHAPE;
  }
  const int prec = cfg->precision;
  const bool planes = (prec == DRIN_PREC_BF16X3 || prec == DRIN_PREC_BF16X3_ALL) && (D % 32 == 0);
  const drin_layer_params& L1 = params->layer[0];
  const drin_layer_params& L2 = params->layer[1];
  __bf16* e1_hi = reinterpret_cast<__bf16*>(ws + L.p_et1);
  const size_t MD = (size_t)M * D;
  if (!planes && (prec == DRIN_PREC_BF16X3 || prec == DRIN_PREC_BF16X3_ALL)) {
    set_error("drin_forward_cached: split-bf16 precision needs embed_dim %% 32 == 0 (got %d)", D);
    return DRIN_E_UNSUPPORTED;
  }

  float* const sk = L.splitk_floats ? ws + L.splitk : nullptr;
  const size_t skf = L.splitk_floats;
  // (1) mention-side pooling and vertex-encoder Linears, [hm | fu] = [mt0; mi0] [W_h1; W_u1]^T + [0; b_u1]
  DRIN_TRY(launch_span_mean(b->mention_text, b->mention_start, b->mention_end, ws + L.span_mean, B, cfg->mention_tokens, D, st));
  DRIN_TRY(launch_axis_mean(b->mention_image, ws + L.mimg, B, cfg->image_regions, R, st));
  float* vm0 = ws + L.vm0;
  DRIN_TRY(launch_gemm_nt_pair({ws + L.span_mean, params->w_mention_text, params->b_mention_text, vm0, D, D, D, B, D, D, nullptr},
                               {ws + L.mimg, params->w_mention_image, params->b_mention_image, vm0 + (size_t)B * D, R, R, D, B, D, R, nullptr},
                               prec, st, sk, skf));
  float* hmfu = ws + L.hmfu;
  DRIN_TRY(launch_gemm_nt(vm0, D, pb + P.wcat1, D, pb + P.bcat1, hmfu, 2 * D, 2 * (int64_t)B, 2 * D, D, false, prec, st, sk, skf));

  // candidate rows outside the tables: clamped by k_cached_pairs, reported here (drin_batch.index_status)
  if (b->index_status) {
    const int64_t pairs = (int64_t)B * cfg->num_candidates;
    KernelTimer timer(DRIN_KC_EDGE, st);
    hipLaunchKernelGGL(k_check_entity_index, dim3((unsigned)std::min<int64_t>(cdiv(pairs, 256), 2048)), dim3(256), 0, st, b->entity_index, pairs,
                       (int64_t)cfg->num_entities, b->index_status);
    DRIN_CHECK_LAUNCH("k_check_entity_index");
  }
  // (2) one gathered pass over the cache rows: edges, layer-1 entity vertices, all cross-candidate sums
  CachedArgs a;
  memset(&a, 0, sizeof(a));
  a.cache = (const float*)cache;
  a.ldc = (int64_t)cache_row_floats(*cfg);
  a.num_entities = cfg->num_entities;
  a.entity_index = b->entity_index;
  a.miet = b->miet_similarity;
  a.mtei = b->mtei_similarity;
  a.span_mean = ws + L.span_mean;
  a.mobj = b->mention_object;
  a.mscore = b->mention_object_score;
  a.fu = hmfu + D;
  a.ldfu = 2 * D;
  a.hm = hmfu;
  a.ldhm = 2 * D;
  a.c_t = pb + P.cb_t;
  a.c_i = pb + P.cb_i;
  a.gamma = L1.ln_weight;
  a.beta = L1.ln_bias;
  a.e1m = ws + L.e1m;
  a.et1 = planes ? nullptr : ws + L.et1;
  a.et1_hi = planes ? e1_hi : nullptr;
  a.et1_lo = planes ? e1_hi + MD : nullptr;
  a.c_part = ws + L.c_part;
  a.s2_part = ws + L.s2_part;
  a.B = B;
  a.N = N;
  a.D4 = D / 4;
  a.R4 = R / 4;
  a.Km = cfg->mention_objects;
  a.chunks = L.chunks;
  a.dynamic = cfg->dynamic_edges != 0;
  for (int k = 0; k < 4; ++k) a.mask[k] = cfg->edge_enabled[k];
  a.cos_eps = cfg->cosine_eps;
  a.miei_eps = cfg->miei_eps;
  a.clip = cfg->clip_scale;
  a.ln_eps = cfg->layer_norm_eps;
  a.act_v = vertex_act(cfg);
  a.act_e = edge_act(cfg);
  if (cache_mixed(*cfg)) {   // DRIN_CACHE_MIXED_F16 rows (the o^ row is read in the PAIR layout: an even number of slots)
    const bool gen = a.act_v != DRIN_ACT_GELU, tiny = a.D4 <= 64 && a.R4 <= 128;
    if (gen && tiny)
      DRIN_TRY((launch_cached_pairs_t<1, 2, false, true, true>(a, st)));
    else if (gen)
      DRIN_TRY((launch_cached_pairs_t<3, 8, false, true, true>(a, st)));
    else if (tiny)
      DRIN_TRY((launch_cached_pairs_t<1, 2, false, false, true>(a, st)));
    else if (a.D4 == 192 && a.R4 == 512)
      DRIN_TRY((launch_cached_pairs_t<3, 8, true, false, true>(a, st)));
    else
      DRIN_TRY((launch_cached_pairs_t<3, 8, false, false, true>(a, st)));
  } else if (a.act_v != DRIN_ACT_GELU && a.D4 <= 64 && a.R4 <= 64)   // non-default vertex activation: the generic-width instantiations
    DRIN_TRY((launch_cached_pairs_t<1, 1, false, true>(a, st)));
  else if (a.act_v != DRIN_ACT_GELU)
    DRIN_TRY((launch_cached_pairs_t<3, 8, false, true>(a, st)));
  else if (a.D4 <= 64 && a.R4 <= 64)
    DRIN_TRY((launch_cached_pairs_t<1, 1, false>(a, st)));
  else if (a.D4 == 192 && a.R4 == 512)
    DRIN_TRY((launch_cached_pairs_t<3, 8, true>(a, st)));
  else
    DRIN_TRY((launch_cached_pairs_t<3, 8, false>(a, st)));

  // (3) layer-1 mention vertices, then what layer 2 needs from them
  float* vm1 = ws + L.vm1;
  {
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_mention_layer1_cached, dim3((unsigned)cdiv(D, 256), (unsigned)B), dim3(256), 0, st, ws + L.c_part,
                       hmfu, 2 * D, pb + P.cb_t, pb + P.cb_i, L1.b_h, vm1, B, D, L.chunks, 1.0f / (float)N);
    DRIN_CHECK_LAUNCH("k_mention_layer1_cached");
  }
  DRIN_TRY(launch_layernorm_gelu(vm1, L1.ln_weight, L1.ln_bias, vm1, nullptr, nullptr, 2 * (int64_t)B, D, cfg->layer_norm_eps, st,
                                 vertex_act(cfg)));
  DRIN_TRY(launch_gemm_nt(vm1, D, L2.w_h, D, nullptr, ws + L.hm2, D, 2 * (int64_t)B, D, D, false, prec, st, sk, skf));
  // (4) layer-2 mention-text vertex
  DRIN_TRY(launch_mention_input2(ws + L.s2_part, vm1, ws + L.agg2, B, D, N, L.chunks, st));
  DRIN_TRY(launch_gemm_nt(ws + L.agg2, D, L2.w_h, D, L2.b_h, ws + L.mt2, D, B, D, D, false, prec, st, sk, skf));
  DRIN_TRY(launch_layernorm_gelu(ws + L.mt2, L2.ln_weight, L2.ln_bias, ws + L.mt2, nullptr, nullptr, B, D, cfg->layer_norm_eps, st,
                                 vertex_act(cfg)));
  // (5) layer-2 entity-text contraction, vertex and score
  float* h2 = ws + L.h2;
  if (planes) {
    const __bf16* w2 = reinterpret_cast<const __bf16*>(pb + P.p_wh2);
    DRIN_TRY(launch_gemm_x3_planes(e1_hi, e1_hi + MD, D, w2, w2 + (size_t)D * D, D, nullptr, h2, D, M, D, D, st));
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
  fa.chunks = L.chunks;
  {
    static const char* fc_env = getenv("DRIN_CACHED_FINAL_CHUNK");   // probe: k_pair_final alone (it reads no partial sums)
    const int per = fc_env ? atoi(fc_env) : 0;
    if (per >= 16) fa.chunks = (int)cdiv(N, per);
  }
  fa.ln_eps = cfg->layer_norm_eps;
  fa.act_v = vertex_act(cfg);
  fa.cos_eps = cfg->cosine_eps;
  return launch_pair_final(fa, st);
}

}  // extern "C"
