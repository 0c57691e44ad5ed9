// One feature row spread over a wave: lane l holds float4 columns l, l + 64, ... (V of them).  Shared by the
// row kernels of the folded inference pipelines (fused_kernels.hip, entity_cache.hip).
#pragma once
#include "device_utils.h"

namespace drin {

template <int V>
struct Row {
  float4 v[V];
};

// ---- packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 per lane and issue slot) --------------------------------
// The row kernels are VALU-ISSUE bound (round 6 SQ counters, profiles/r6_stream_sq.json: k_pair_final keeps the vector pipe busy
// 100 % of its cycles, k_pair_layer1 78 %; ~33 vector instructions per element, none packed).  LayerNorm + GELU + the combine and the
// dots on PAIRS of columns: every fma / mul / add serves two elements; the two transcendentals (v_rcp_f32, v_exp_f32) and |x| stay
// per element.  Same operations on every element as the scalar form (the sums of a row are associated differently: a few ulps on
// mean / variance).  -DDRIN_PK_ROWS=0 builds the scalar form (A/B: tools/variant_ab.sh).
#ifndef DRIN_PK_ROWS
#define DRIN_PK_ROWS 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk(float a, float b) { return f32x2{a, b}; }
__device__ __forceinline__ f32x2 splat2(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 lo2(const float4& v) { return f32x2{v.x, v.y}; }
__device__ __forceinline__ f32x2 hi2(const float4& v) { return f32x2{v.z, v.w}; }
__device__ __forceinline__ float4 join2(f32x2 a, f32x2 b) { return make_float4(a.x, a.y, b.x, b.y); }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// gelu_fast on two values: erf by Abramowitz-Stegun 7.1.26 as there; 0.5 x (1 + sign(x) erf|z|) = 0.5 x + 0.5 |x| erf|z| saves the two
// sign transfers
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 ax = pk(fabsf(x.x), fabsf(x.y));
  const f32x2 z = ax * 0.70710678118654752440f;
  const f32x2 den = pk_fma(z, splat2(0.3275911f), splat2(1.0f));
  const f32x2 t = pk(__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y));
  f32x2 p = pk_fma(t, splat2(1.061405429f), splat2(-1.453152027f));
  p = pk_fma(t, p, splat2(1.421413741f));
  p = pk_fma(t, p, splat2(-0.284496736f));
  p = pk_fma(t, p, splat2(0.254829592f));
  p = t * p;
  const f32x2 arg = (z * z) * (-1.44269504088896340736f);                      // exp(-z^2) = 2^(-z^2 log2 e)
  const f32x2 ex = pk(__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y));
  const f32x2 erf_abs = pk_fma(-p, ex, splat2(1.0f));                          // erf(|x| / sqrt 2)
  return pk_fma(ax * 0.5f, erf_abs, x * 0.5f);
}

// Which float4 column of the row register slot jj of lane `lane` holds.  Default: lane, lane + 64, ... (one 16-byte fp32 load
// per slot).  PAIR (rows stored as bf16): slots 2 q and 2 q + 1 are the two halves of EIGHT consecutive columns
// 8 (lane + 64 q) .. + 7 - one 16-byte load of bf16 per pair of slots instead of two 8-byte ones (8-byte accesses move at
// 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md).  Every helper that meets memory takes the same flag; the arithmetic
// on whole rows (dot_rows, axpy_row) does not care.
template <bool PAIR>
__device__ __forceinline__ int row_col4(int lane, int jj) {
  return PAIR ? 2 * (lane + 64 * (jj >> 1)) + (jj & 1) : lane + 64 * jj;
}

template <int V, typename T = float>
__device__ __forceinline__ Row<V> load_row(const T* __restrict__ p, int lane, int n4) {
  Row<V> r;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c4 = lane + 64 * j;
    r.v[j] = c4 < n4 ? ld4(p + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  return r;
}
// the same with the streaming (non-temporal) cache policy: rows that are read exactly once
template <int V, typename T = float>
__device__ __forceinline__ Row<V> load_row_stream(const T* __restrict__ p, int lane, int n4) {
  Row<V> r;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c4 = lane + 64 * j;
    r.v[j] = c4 < n4 ? ld4_stream(p + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  return r;
}
// a bf16 row in the PAIR layout: V / 2 loads of 16 bytes per lane (n4 even)
template <int V>
__device__ __forceinline__ Row<V> load_row_stream_pair(const __bf16* __restrict__ p, int lane, int n4) {
  static_assert(V % 2 == 0, "the PAIR layout takes two register slots per load");
  Row<V> r;
#pragma unroll
  for (int q = 0; q < V / 2; ++q) {
    const int c8 = lane + 64 * q;
    if (2 * c8 < n4) {
      const u32x4_t v = ld16_stream(reinterpret_cast<const char*>(p) + (size_t)c8 * 16);
      r.v[2 * q] = make_float4(__builtin_bit_cast(float, v[0] << 16), __builtin_bit_cast(float, v[0] & 0xffff0000u),
                               __builtin_bit_cast(float, v[1] << 16), __builtin_bit_cast(float, v[1] & 0xffff0000u));
      r.v[2 * q + 1] = make_float4(__builtin_bit_cast(float, v[2] << 16), __builtin_bit_cast(float, v[2] & 0xffff0000u),
                                   __builtin_bit_cast(float, v[3] << 16), __builtin_bit_cast(float, v[3] & 0xffff0000u));
    } else {
      r.v[2 * q] = r.v[2 * q + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  return r;
}
template <int V>
__device__ __forceinline__ void store_row(float* __restrict__ p, const Row<V>& r, int lane, int n4) {
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < n4) st4(p + c4 * 4, r.v[j]);
  }
}
// v = hi + lo in bf16 (operand planes of the split-bf16 GEMM, gemm_x3_planes.hip)
template <int V, bool PAIR = false>
__device__ __forceinline__ void store_row_planes(void* __restrict__ hi, void* __restrict__ lo, int64_t row_off,
                                                 const Row<V>& r, int lane, int n4) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  __bf16* ph = reinterpret_cast<__bf16*>(hi) + row_off;
  __bf16* pl = reinterpret_cast<__bf16*>(lo) + row_off;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c4 = row_col4<PAIR>(lane, j);
    if (c4 < n4) {
      const float4 v = r.v[j];
      bf16x4 h, l;
      split_bf16x4(v, h, l);
      // streaming stores: the planes are consumed by a later GEMM launch, after > 1 GB of other traffic
      // (measured on one box: k_entity_stream 8.06 -> 7.94 ms; on the GEMM epilogues the same policy costs 2 %)
      __builtin_nontemporal_store(h, reinterpret_cast<bf16x4*>(ph + c4 * 4));
      if (lo != nullptr) __builtin_nontemporal_store(l, reinterpret_cast<bf16x4*>(pl + c4 * 4));  // NULL: exact in bf16
    }
  }
}
// row / scale as ONE fp16 plane (the A operand of the single-plane fp16 contraction, gemm_x3_planes.hip), exact widths only
// (no column guards), 16-byte streaming stores.  PAIR: a lane's slot pair already holds eight consecutive columns.  Otherwise a
// lane holds four columns per slot: the lanes of a pair (l, l ^ 1) swap one packed slot each, so that the even lane stores the
// eight columns 4 l .. 4 l + 7 of slot 2 q and the odd lane those of slot 2 q + 1 (8-byte stores move at 0.54-0.70 of the
// 16-byte rate, MI355X_MICROARCH.md).  `inv` = 1 / scale, a power of two: the division is exact.
template <int V, bool PAIR = false>
__device__ __forceinline__ void store_row_f16_scaled(void* __restrict__ out, int64_t row_off, const Row<V>& r, float inv, int lane) {
  static_assert(V % 2 == 0, "slot pairs");
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  _Float16* base = reinterpret_cast<_Float16*>(out) + row_off;
  auto pack = [inv](float4 v, uint32_t& a, uint32_t& b) {
    const f32x2 lo = {v.x * inv, v.y * inv}, hi = {v.z * inv, v.w * inv};
    a = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, f16x2));
    b = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, f16x2));
  };
#pragma unroll
  for (int q = 0; q < V / 2; ++q) {
    uint32_t a0, a1, b0, b1;
    pack(r.v[2 * q], a0, a1);
    pack(r.v[2 * q + 1], b0, b1);
    u32x4_t o;
    int col;
    if (PAIR) {
      o = u32x4_t{a0, a1, b0, b1};
      col = 8 * (lane + 64 * q);
    } else {
      const bool odd = lane & 1;
      const uint32_t s0 = odd ? a0 : b0, s1 = odd ? a1 : b1;        // the slot the partner stores
      const uint32_t r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
      o = odd ? u32x4_t{r0, r1, b0, b1} : u32x4_t{a0, a1, r0, r1};
      col = odd ? 4 * (lane - 1 + 64 * (2 * q + 1)) : 4 * (lane + 64 * (2 * q));
    }
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t*>(base + col));   // (write-back stores measured the same: round 5)
  }
}
template <int V>
__device__ __forceinline__ float dot_rows(const Row<V>& a, const Row<V>& b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) s += dot4(a.v[j], b.v[j]);
  return s;
}
template <int V>
__device__ __forceinline__ void axpy_row(Row<V>& acc, float w, const Row<V>& x) {
#pragma unroll
  for (int j = 0; j < V; ++j) acc.v[j] = fma4(w, x.v[j], acc.v[j]);
}
// the same two on the packed pipe (row kernels of the folded paths; a dot's terms are summed in another association)
template <int V>
__device__ __forceinline__ float dot_rows_pk(const Row<V>& a, const Row<V>& b) {
#if DRIN_PK_ROWS
  f32x2 s2 = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < V; ++j) s2 = pk_fma(hi2(a.v[j]), hi2(b.v[j]), pk_fma(lo2(a.v[j]), lo2(b.v[j]), s2));
  return s2.x + s2.y;
#else
  return dot_rows<V>(a, b);
#endif
}
template <int V>
__device__ __forceinline__ void axpy_row_pk(Row<V>& acc, float w, const Row<V>& x) {
#if DRIN_PK_ROWS
  const f32x2 w2 = {w, w};
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const f32x2 l = pk_fma(w2, f32x2{x.v[j].x, x.v[j].y}, f32x2{acc.v[j].x, acc.v[j].y});
    const f32x2 h = pk_fma(w2, f32x2{x.v[j].z, x.v[j].w}, f32x2{acc.v[j].z, acc.v[j].w});
    acc.v[j] = make_float4(l.x, l.y, h.x, h.y);
  }
#else
  axpy_row<V>(acc, w, x);
#endif
}
template <int V>
__device__ __forceinline__ Row<V> zero_row() {
  Row<V> r;
#pragma unroll
  for (int j = 0; j < V; ++j) r.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  return r;
}
// dot with a vector that lives in LDS (same lane -> column map)
template <int V, bool PAIR = false>
__device__ __forceinline__ float dot_row_lds(const Row<V>& a, const float* lds, int lane, int n4) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c4 = row_col4<PAIR>(lane, j);
    if (c4 < n4) s += dot4(a.v[j], ld4(lds + c4 * 4));
  }
  return s;
}

// LayerNorm + GELU of one row held in registers (model.py:128), gamma / beta in LDS (per-workgroup constants).
// GENERIC_ACT: the vertex activation by id (`act`, drin_activation resolved) instead of the default's branch-free gelu -
// a separate instantiation, so the default's code is what it was.
template <int DV, bool GENERIC_ACT = false>
__device__ __forceinline__ Row<DV> ln_gelu_row_lds(const Row<DV>& h, const float* gamma, const float* beta, int lane,
                                                   int D4, float eps, int act = DRIN_ACT_GELU) {
  const float inv_d = 1.0f / (float)(D4 * 4);
#if DRIN_PK_ROWS
  if constexpr (!GENERIC_ACT) {
    f32x2 s2 = splat2(0.f);
#pragma unroll
    for (int j = 0; j < DV; ++j) s2 = (s2 + lo2(h.v[j])) + hi2(h.v[j]);   // (columns past D4 are zero in every caller's row)
    const float mu_ = wave_sum(s2.x + s2.y) * inv_d;
    const f32x2 mu2 = splat2(mu_);
    f32x2 q2 = splat2(0.f);
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < D4) {
        const f32x2 dl = lo2(h.v[j]) - mu2, dh = hi2(h.v[j]) - mu2;
        q2 = pk_fma(dl, dl, q2);
        q2 = pk_fma(dh, dh, q2);
      }
    }
    const f32x2 rstd2 = splat2(1.0f / sqrtf(wave_sum(q2.x + q2.y) * inv_d + eps));
    Row<DV> y;
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < D4) {
        const float4 g = ld4(gamma + c4 * 4), bt = ld4(beta + c4 * 4);
        const f32x2 yl = gelu_fast2(pk_fma((lo2(h.v[j]) - mu2) * rstd2, lo2(g), lo2(bt)));
        const f32x2 yh = gelu_fast2(pk_fma((hi2(h.v[j]) - mu2) * rstd2, hi2(g), hi2(bt)));
        y.v[j] = join2(yl, yh);
      } else {
        y.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return y;
  }
#endif
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < DV; ++j) s += (h.v[j].x + h.v[j].y) + (h.v[j].z + h.v[j].w);
  // (Measured in round 4 and dropped - profiles/r4_ln_stats_ab.txt: both moments in ONE round of wave reductions,
  //  var = E[x^2] - mu^2, so that the mean -> centred squares chain is one stage shorter.  Same box, alternating: the row
  //  kernels of the headline 1.12 -> 1.24 ms, of config 5 3.35 -> 3.48 ms - slower, although it is fewer instructions; the
  //  dependent reductions are not what these kernels wait for.)
  const float mu = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < DV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < D4) {
      const float dx = h.v[j].x - mu, dy = h.v[j].y - mu, dz = h.v[j].z - mu, dw = h.v[j].w - mu;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
  Row<DV> y;
#pragma unroll
  for (int j = 0; j < DV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < D4) {
      const float4 g = ld4(gamma + c4 * 4), bt = ld4(beta + c4 * 4);
      y.v[j].x = GENERIC_ACT ? act_apply(act, (h.v[j].x - mu) * rstd * g.x + bt.x) : gelu_fast((h.v[j].x - mu) * rstd * g.x + bt.x);
      y.v[j].y = GENERIC_ACT ? act_apply(act, (h.v[j].y - mu) * rstd * g.y + bt.y) : gelu_fast((h.v[j].y - mu) * rstd * g.y + bt.y);
      y.v[j].z = GENERIC_ACT ? act_apply(act, (h.v[j].z - mu) * rstd * g.z + bt.z) : gelu_fast((h.v[j].z - mu) * rstd * g.z + bt.z);
      y.v[j].w = GENERIC_ACT ? act_apply(act, (h.v[j].w - mu) * rstd * g.w + bt.w) : gelu_fast((h.v[j].w - mu) * rstd * g.w + bt.w);
    } else {
      y.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  return y;
}

// out = base + w1 u1 + w2 u2 + c with u1, u2, c in LDS
template <int DV>
__device__ __forceinline__ Row<DV> combine_rows_lds(const Row<DV>& base, float w1, const float* u1, float w2,
                                                    const float* u2, const float* c, int lane, int D4) {
  Row<DV> r;
#pragma unroll
  for (int j = 0; j < DV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < D4) {
#if DRIN_PK_ROWS
      const float4 a1 = ld4(u1 + c4 * 4), a2 = ld4(u2 + c4 * 4), cc = ld4(c + c4 * 4);
      const f32x2 w1_ = splat2(w1), w2_ = splat2(w2);
      r.v[j] = join2(pk_fma(w1_, lo2(a1), pk_fma(w2_, lo2(a2), lo2(base.v[j]) + lo2(cc))),
                     pk_fma(w1_, hi2(a1), pk_fma(w2_, hi2(a2), hi2(base.v[j]) + hi2(cc))));
#else
      r.v[j] = fma4(w1, ld4(u1 + c4 * 4), fma4(w2, ld4(u2 + c4 * 4), base.v[j] + ld4(c + c4 * 4)));
#endif
    } else
      r.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  return r;
}

}  // namespace drin
