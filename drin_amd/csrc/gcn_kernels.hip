// Message-passing kernels of GCNLayer.forward (drin/model.py:121-153) around the W_h / W_u / W_v
// contractions: neighbour aggregation, LayerNorm + GELU, dynamic edge update.  All HBM/L2-bound
// row kernels: 16-byte lane accesses, wave64 shuffle reductions, no LDS.
#include "device_utils.h"
#include "internal.h"

namespace drin {

// ------------------------------------------------------------------------------------------------
// mention <- entity (model.py:143-144) summed over the two neighbour types, plus the self term (:128):
//   out[b, :] = mean_n(e1[b,n] v1[b,n,:]) + mean_n(e2[b,n] v2[b,n,:]) + u[b, :]
// grid (ceil(D4/64), B): one wave per 64 float4 columns of a mention, walking the N candidates.
__global__ void __launch_bounds__(64) k_mention_aggregate(const float* __restrict__ e1, const float* __restrict__ v1,
                                                          const float* __restrict__ e2, const float* __restrict__ v2,
                                                          const float* __restrict__ u, float* __restrict__ out, int N,
                                                          int D4) {
  const int c4 = blockIdx.x * 64 + threadIdx.x;
  if (c4 >= D4) return;
  const int64_t b = blockIdx.y;
  const float* p1 = v1 + (b * N) * (int64_t)D4 * 4 + (int64_t)c4 * 4;
  const float* p2 = v2 + (b * N) * (int64_t)D4 * 4 + (int64_t)c4 * 4;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  int n = 0;
  for (; n + 4 <= N; n += 4) {
    float4 a[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = ld4(p1 + (int64_t)(n + j) * D4 * 4);
      c[j] = ld4(p2 + (int64_t)(n + j) * D4 * 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1 = fma4(e1[b * N + n + j], a[j], s1);
      s2 = fma4(e2[b * N + n + j], c[j], s2);
    }
  }
  for (; n < N; ++n) {
    s1 = fma4(e1[b * N + n], ld4(p1 + (int64_t)n * D4 * 4), s1);
    s2 = fma4(e2[b * N + n], ld4(p2 + (int64_t)n * D4 * 4), s2);
  }
  const float cnt = (float)N;
  const float4 uu = ld4(u + b * (int64_t)D4 * 4 + (int64_t)c4 * 4);
  float4 r;
  r.x = (s1.x / cnt + s2.x / cnt) + uu.x;
  r.y = (s1.y / cnt + s2.y / cnt) + uu.y;
  r.z = (s1.z / cnt + s2.z / cnt) + uu.z;
  r.w = (s1.w / cnt + s2.w / cnt) + uu.w;
  st4(out + b * (int64_t)D4 * 4 + (int64_t)c4 * 4, r);
}

int launch_mention_aggregate(const float* e1, const float* v1, const float* e2, const float* v2, const float* u,
                             float* out, int B, int N, int D, hipStream_t st) {
  if (B <= 0) return DRIN_OK;
  if (D % 4) {
    set_error("mention_aggregate: D=%d must be a multiple of 4", D);
    return DRIN_E_SHAPE;
  }
  for (int b0 = 0; b0 < B; b0 += 65535) {
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    const int64_t po = (int64_t)b0 * N, vo = po * D;
    dim3 grid((unsigned)cdiv(D / 4, 64), (unsigned)nb);
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_mention_aggregate, grid, dim3(64), 0, st, e1 + po, v1 + vo, e2 + po, v2 + vo,
                       u + (int64_t)b0 * D, out + (int64_t)b0 * D, N, D / 4);
    DRIN_CHECK_LAUNCH("k_mention_aggregate");
  }
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// entity <- mention (model.py:146) summed over the two neighbour types, plus the self term (:128):
//   out[b, n, :] = e1[b,n] m1[b, :] + e2[b,n] m2[b, :] + v[b, n, :]
__global__ void __launch_bounds__(256) k_entity_aggregate(const float* __restrict__ e1, const float* __restrict__ m1,
                                                          const float* __restrict__ e2, const float* __restrict__ m2,
                                                          const float* __restrict__ v, float* __restrict__ out,
                                                          int64_t total4, int N, int D4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int64_t p = i / D4;
  const int c4 = (int)(i - p * D4);
  const int64_t b = p / N;
  const float a1 = e1[p], a2 = e2[p];
  const float4 x1 = ld4(m1 + b * (int64_t)D4 * 4 + c4 * 4), x2 = ld4(m2 + b * (int64_t)D4 * 4 + c4 * 4);
  const float4 vv = ld4(v + i * 4);
  float4 r;
  r.x = (a1 * x1.x + a2 * x2.x) + vv.x;
  r.y = (a1 * x1.y + a2 * x2.y) + vv.y;
  r.z = (a1 * x1.z + a2 * x2.z) + vv.z;
  r.w = (a1 * x1.w + a2 * x2.w) + vv.w;
  st4(out + i * 4, r);
}

int launch_entity_aggregate(const float* e1, const float* m1, const float* e2, const float* m2, const float* v,
                            float* out, int B, int N, int D, hipStream_t st) {
  const int64_t total4 = (int64_t)B * N * (D / 4);
  if (total4 <= 0) return DRIN_OK;
  if (D % 4) {
    set_error("entity_aggregate: D=%d must be a multiple of 4", D);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_entity_aggregate, dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, st, e1, m1, e2, m2, v, out,
                     total4, N, D / 4);
  DRIN_CHECK_LAUNCH("k_entity_aggregate");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// y = gelu(layer_norm(h)) (model.py:128; nn.LayerNorm eps 1e-5, biased variance, affine; exact-erf gelu).
// One wave per row, the row held in registers (up to 4 float4 per lane: D <= 1024); y may alias h.
constexpr int LN_MAXV = 4;

__global__ void __launch_bounds__(256) k_layernorm_gelu(const float* h, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* y,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                        int64_t rows, int D4, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* hr = h + row * (int64_t)D4 * 4;
  float4 x[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c4 = lane + 64 * j;
    x[j] = c4 < D4 ? ld4(hr + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
  }
  const float inv_d = 1.0f / (float)(D4 * 4);
  const float mu = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < D4) {
      const float dx = x[j].x - mu, dy = x[j].y - mu, dz = x[j].z - mu, dw = x[j].w - mu;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float var = wave_sum(q) * inv_d;
  const float rstd = 1.0f / sqrtf(var + eps);
  float* yr = y + row * (int64_t)D4 * 4;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < D4) {
      const float4 g = ld4(gamma + c4 * 4), bt = ld4(beta + c4 * 4);
      float4 r;
      r.x = gelu_erf((x[j].x - mu) * rstd * g.x + bt.x);
      r.y = gelu_erf((x[j].y - mu) * rstd * g.y + bt.y);
      r.z = gelu_erf((x[j].z - mu) * rstd * g.z + bt.z);
      r.w = gelu_erf((x[j].w - mu) * rstd * g.w + bt.w);
      st4(yr + c4 * 4, r);
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mu;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

int launch_layernorm_gelu(const float* h, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          int64_t rows, int D, float eps, hipStream_t st) {
  if (rows <= 0) return DRIN_OK;
  if (D % 4 || D > 256 * LN_MAXV) {
    set_error("layernorm_gelu: D=%d must be a multiple of 4 and <= %d", D, 256 * LN_MAXV);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_layernorm_gelu, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, h, gamma, beta, y, mean, rstd,
                     rows, D / 4, eps);
  DRIN_CHECK_LAUNCH("k_layernorm_gelu");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// dynamic edge (model.py:148-153 with w_m = Identity :112, sigmoid :133):
//   out[b, n] = sigmoid(mean_d(fu[b, :] * fv[b, n, :]) + e[b, n])       one wave per pair
__global__ void __launch_bounds__(256) k_edge_update(const float* __restrict__ fu, const float* __restrict__ fv,
                                                     const float* __restrict__ e, float* __restrict__ out,
                                                     int64_t pairs, int N, int D4) {
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= pairs) return;
  const int lane = threadIdx.x & 63;
  const float* ur = fu + (p / N) * (int64_t)D4 * 4;
  const float* vr = fv + p * (int64_t)D4 * 4;
  float s = 0.f;
  for (int c4 = lane; c4 < D4; c4 += 64) s += dot4(ld4(ur + c4 * 4), ld4(vr + c4 * 4));
  s = wave_sum(s);
  if (lane == 0) out[p] = sigmoidf(s / (float)(D4 * 4) + e[p]);
}

int launch_edge_update(const float* fu, const float* fv, const float* e, float* out, int B, int N, int D,
                       hipStream_t st) {
  const int64_t pairs = (int64_t)B * N;
  if (pairs <= 0) return DRIN_OK;
  if (D % 4) {
    set_error("edge_update: D=%d must be a multiple of 4", D);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_edge_update, dim3((unsigned)cdiv(pairs, 4)), dim3(256), 0, st, fu, fv, e, out, pairs, N, D / 4);
  DRIN_CHECK_LAUNCH("k_edge_update");
  return DRIN_OK;
}

}  // namespace drin
