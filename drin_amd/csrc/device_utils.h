// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace drin {

constexpr int kWave = 64;

// An entity_index outside [0, E-1] is clamped (memory safety) AND reported: the first reporter - one compare-and-swap on the
// error path only - leaves {1, pair, value low, value high} in the caller's int32[4] (drin_batch.index_status; sticky).
__device__ __forceinline__ void report_bad_index(int32_t* status, int64_t pair, int64_t value) {
  if (status == nullptr) return;
  if (atomicCAS(reinterpret_cast<int*>(status), 0, 1) == 0) {
    status[1] = (int32_t)pair;
    status[2] = (int32_t)(uint32_t)(uint64_t)value;
    status[3] = (int32_t)(uint32_t)((uint64_t)value >> 32);
  }
}

// Sum over the 64 lanes, result in every lane.  Four DPP adds (quad swaps, half-row and row mirrors: plain
// VALU, a few cycles each) leave every lane of a 16-lane row with the row sum; the four row sums are
// then read as scalars.  The usual __shfl_xor butterfly compiles to six dependent ds_bpermute_b32
// round trips through the LDS crossbar (~100+ cycles each), which is what the per-pair row kernels -
// ten or more reductions per pair - were actually waiting on.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v = dpp_add<0x141>(v);  // row_half_mirror: lane i <-> 7 - i within 8
  v = dpp_add<0x140>(v);  // row_mirror: lane i <-> 15 - i within 16
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

// max over the 64 lanes of non-negative values, the same way (the DPP fill value 0 is neutral for them)
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)));
}
__device__ __forceinline__ float wave_max(float v) {
  v = dpp_max<0xB1>(v);
  v = dpp_max<0x4E>(v);
  v = dpp_max<0x141>(v);
  v = dpp_max<0x140>(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// the smallest power of two >= m for a positive finite m (1 for m == 0, a NaN or an infinity: such a row is not scaled -
// its NaNs / infinities propagate through the product as they are)
__device__ __forceinline__ float pow2_at_least(float m) {
  if (!(m > 0.f) || !(m < __builtin_inff())) return 1.0f;
  const uint32_t bits = __builtin_bit_cast(uint32_t, m);
  const uint32_t up = (bits & 0x007fffffu) ? ((bits & 0x7f800000u) + 0x00800000u) : (bits & 0x7f800000u);
  // (a denormal m rounds up to the smallest normal; the largest binade rounds up to 2^127 at most - finite)
  const float p = __builtin_bit_cast(float, up ? (up >= 0x7f800000u ? 0x7f000000u : up) : 0x00800000u);
  return p;
}

// the scale of one fp16 field of a DRIN_CACHE_MIXED_F16 row: |x| / scale <= 1, and the scale's reciprocal is a normal
// number too (2^-126 .. 2^126), so that dividing and multiplying by it are exact in any denormal mode
__device__ __forceinline__ float cache_field_scale(float max_abs) { return fminf(pow2_at_least(max_abs), 0x1p126f); }

// v = hi + lo with hi = bf16(v), lo = bf16(v - hi), both rounded to nearest-even (the operand planes of the split-bf16
// products).  Written on PAIRS: one packed conversion gives both hi's, two bit operations widen them again, a packed
// subtract and one packed conversion give both lo's - five vector instructions per pair.  Element by element the same
// arithmetic compiled to eight per pair for half of the pairs (each hi converted once alone and once packed); same bits.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  const f32x2_t w = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - w, bf16x2_t));
}
__device__ __forceinline__ void split_bf16x4(float4 v, bf16x4_t& hi, bf16x4_t& lo) {
  uint2 h, l;
  split_bf16_pair(v.x, v.y, h.x, l.x);
  split_bf16_pair(v.z, v.w, h.y, l.y);
  hi = __builtin_bit_cast(bf16x4_t, h);
  lo = __builtin_bit_cast(bf16x4_t, l);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// once-read streaming loads (entity rows of k_entity_stream): non-temporal cache policy
#ifdef DRIN_NO_NT_LOADS
__device__ __forceinline__ float4 ld4_stream(const float* p) { return ld4(p); }
#else
__device__ __forceinline__ float4 ld4_stream(const float* p) {
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const f4_t v = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
#endif
// features stored as bf16 (drin_config.feature_dtype): four values in 8 bytes, widened exactly
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ float4 ld4_stream(const __bf16* p) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#ifdef DRIN_NO_NT_LOADS
  const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
#else
  const bf16x4_t v = __builtin_nontemporal_load(reinterpret_cast<const bf16x4_t*>(p));
#endif
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// sixteen raw bytes (eight bf16) per lane with the streaming policy: the flat walk of bf16 token rows (fused_kernels.hip)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t ld16_stream(const char* p) {
#ifdef DRIN_NO_NT_LOADS
  return *reinterpret_cast<const u32x4_t*>(p);
#else
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
#endif
}
// eight raw bytes (four fp16 of a DRIN_CACHE_MIXED_F16 row) per lane, same policy
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2_t ld8_stream(const char* p) {
#ifdef DRIN_NO_NT_LOADS
  return *reinterpret_cast<const u32x2_t*>(p);
#else
  return __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
#endif
}
// four fp16 values in two dwords <-> float4 (widening is exact; narrowing rounds to nearest even)
__device__ __forceinline__ float4 f16x4_to_float4(uint32_t a, uint32_t b) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t x = __builtin_bit_cast(h2_t, a), y = __builtin_bit_cast(h2_t, b);
  return make_float4((float)x[0], (float)x[1], (float)y[0], (float)y[1]);
}
__device__ __forceinline__ u32x2_t float4_to_f16x4(float4 v) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t x = {(_Float16)v.x, (_Float16)v.y}, y = {(_Float16)v.z, (_Float16)v.w};
  u32x2_t r;
  r[0] = __builtin_bit_cast(uint32_t, x);
  r[1] = __builtin_bit_cast(uint32_t, y);
  return r;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float4 operator+(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 operator*(float4 a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// exact-erf GELU (torch.nn.functional.gelu default, args.py:35)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// Branch-free GELU for the fused inference row kernels: erf by Abramowitz-Stegun 7.1.26
// (|error| <= 6e-7 in fp32 arithmetic, i.e. <= 5e-7 on gelu - the same order as fp32 rounding of the
// reference's own erf) with the hardware exp2 / rcp; ~14 VALU instructions against ~60 with branches
// for the library erff.  The row kernels are VALU-bound on this function.
// (v_rcp_f32 itself, 1 ulp: `__frcp_rn` expands to the correctly rounded division sequence - ten more instructions per
//  element, a fifth of k_pair_final - for an error forty times below the polynomial's own: row kernels 1.63 -> 1.41 ms per
//  413 696 pairs.  Measured on top of it and dropped: the same arithmetic two columns at a time on the packed fp32 pipe
//  (v_pk_fma_f32; 381 -> 244 VALU instructions per row of k_pair_final): no further change - no longer issue-bound.)
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly =
      t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = 1.0f - poly * __expf(-z * z);  // erf(|x| / sqrt 2)
  return 0.5f * x * (1.0f + copysignf(e, x));
}
// d/dx gelu_erf = Phi(x) + x phi(x).  Branch-free like gelu_fast: the Abramowitz-Stegun erf and the density share one
// hardware exponential exp(-x^2 / 2); |error| <= 4e-7 against the library erff / expf form (~100 VALU instructions
// with branches; the LayerNorm backward kernel was half VALU time on it).
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly =
      t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float g = __expf(-0.5f * x * x);  // = exp(-z^2)
  const float erf_abs = 1.0f - poly * g;
  const float cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  return fmaf(x * 0.39894228040143267794f, g, cdf);
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// gcn_vertex_activation / gcn_edge_activation by id (drin_activation, already resolved: never DRIN_ACT_DEFAULT).
// The id is uniform over a launch, so the switch is a scalar branch.
__device__ __forceinline__ float act_apply(int id, float x) {
  switch (id) {
    case 2: return sigmoidf(x);
    case 3: return x < 0.f ? 0.f : x;            // F.relu; a NaN stays a NaN
    case 4: return tanhf(x);
    case 5: return x / (1.0f + expf(-x));        // F.silu = x sigmoid(x)
    default: return gelu_erf(x);
  }
}
// the edge activation on the folded inference kernels: the default stays the plain sigmoid expression it was
__device__ __forceinline__ float edge_act_apply(int id, float x) { return id == 2 ? sigmoidf(x) : act_apply(id, x); }
// derivative at the pre-activation z (LayerNorm backward recomputes z)
__device__ __forceinline__ float act_grad(int id, float z) {
  switch (id) {
    case 2: { const float s = sigmoidf(z); return s * (1.0f - s); }
    case 3: return z > 0.f ? 1.0f : 0.f;
    case 4: { const float t = tanhf(z); return 1.0f - t * t; }
    case 5: { const float s = sigmoidf(z); return s * fmaf(z, 1.0f - s, 1.0f); }
    default: return gelu_erf_grad(z);
  }
}
// derivative from the OUTPUT y = act(z) (the edge update keeps e', not its pre-activation): sigmoid, tanh, relu only
__device__ __forceinline__ float act_grad_from_output(int id, float y) {
  switch (id) {
    case 3: return y > 0.f ? 1.0f : 0.f;
    case 4: return 1.0f - y * y;
    default: return y * (1.0f - y);
  }
}

// The edge update's backward: `act` carries kActFromPre when the buffer it reads holds the PRE-activation z (kept by the
// forward for gelu / silu edges, model.py:118 takes any F.* name), else the buffer is the stored edge e' = act(z).
constexpr int kActFromPre = 0x100;
__device__ __forceinline__ float edge_act_grad(int act, float v) {
  return (act & kActFromPre) ? act_grad(act & 0xff, v) : act_grad_from_output(act, v);
}

// nn.CosineSimilarity as torch>=2 evaluates it: each norm clamped at eps separately.
__device__ __forceinline__ float cosine_from_sums(float xy, float xx, float yy, float eps) {
  return xy / (fmaxf(sqrtf(xx), eps) * fmaxf(sqrtf(yy), eps));
}

// One wave per (mention, candidate) pair, four pairs per 256-thread block.  Launched as grid (ceil(N / 4), B) - see
// pair_grid() - the pair comes from the block indices; the flat grid (ceil(pairs / 4), 1) of B == 1 or B > 65 535 pays a
// 64-bit division per wave: ~100 VALU instructions, more than the arithmetic of a 3 KB row.
__device__ __forceinline__ bool wave_pair(int64_t pairs, int N, int64_t& p, int64_t& b) {
  const int wave = threadIdx.x >> 6;
  if (gridDim.y > 1) {
    const int n = (int)blockIdx.x * 4 + wave;
    b = blockIdx.y;
    p = b * N + n;
    return n < N;
  }
  p = (int64_t)blockIdx.x * 4 + wave;
  b = p / N;
  return p < pairs;
}

}  // namespace drin
