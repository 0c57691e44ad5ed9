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
// Both aggregations of a scalar-edge layer in ONE pass over the entity vertices (model.py:143-146 + the self terms of
// :128): every et / ei element is read once and feeds the mention sums and its own entity update.
//   agg_mt[b] = mean_n(e_tt et) + mean_n(e_ti ei) + mt[b]      agg_et[b,n] = e_tt mt[b] + e_it mi[b] + et[b,n]
//   agg_mi[b] = mean_n(e_it et) + mean_n(e_ii ei) + mi[b]      agg_ei[b,n] = e_ti mt[b] + e_ii mi[b] + ei[b,n]
// (the image rows only when LIVE_IMAGE: the last layer's image vertices never reach the score).  Same grid, same
// per-column summation order as k_mention_aggregate / k_entity_aggregate: bit-identical to the four separate launches,
// at half their HBM traffic.
template <bool LIVE_IMAGE>
__global__ void __launch_bounds__(64) k_layer_aggregate(const float* __restrict__ e, int64_t ES,
                                                        const float* __restrict__ vm, const float* __restrict__ ve,
                                                        float* __restrict__ agg_m, float* __restrict__ agg_e, int B,
                                                        int N, int D4) {
  const int c4 = blockIdx.x * 64 + threadIdx.x;
  if (c4 >= D4) return;
  const int64_t b = blockIdx.y;
  const int64_t D = (int64_t)D4 * 4, MD = (int64_t)B * N * D, BD = (int64_t)B * D;
  const int64_t row0 = b * N * D + (int64_t)c4 * 4;
  const float* pt = ve + row0;
  const float* pi = ve + MD + row0;
  float* ot = agg_e + row0;
  float* oi = agg_e + MD + row0;
  const float* e_tt = e + b * N;
  const float* e_ti = e_tt + ES;
  const float* e_it = e_tt + 2 * ES;
  const float* e_ii = e_tt + 3 * ES;
  const float4 mt = ld4(vm + b * D + (int64_t)c4 * 4), mi = ld4(vm + BD + b * D + (int64_t)c4 * 4);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s_tt = zero, s_ti = zero, s_it = zero, s_ii = zero;
  auto entity_out = [&](float a1, float a2, const float4& v) {
    float4 r;
    r.x = (a1 * mt.x + a2 * mi.x) + v.x;
    r.y = (a1 * mt.y + a2 * mi.y) + v.y;
    r.z = (a1 * mt.z + a2 * mi.z) + v.z;
    r.w = (a1 * mt.w + a2 * mi.w) + v.w;
    return r;
  };
  int n = 0;
  for (; n + 4 <= N; n += 4) {
    float4 a[4], c[4];
    float w_tt[4], w_ti[4], w_it[4], w_ii[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = ld4(pt + (int64_t)(n + j) * D);
      c[j] = ld4(pi + (int64_t)(n + j) * D);
      w_tt[j] = e_tt[n + j];
      w_ti[j] = e_ti[n + j];
      w_it[j] = e_it[n + j];
      w_ii[j] = e_ii[n + j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s_tt = fma4(w_tt[j], a[j], s_tt);
      s_ti = fma4(w_ti[j], c[j], s_ti);
      st4(ot + (int64_t)(n + j) * D, entity_out(w_tt[j], w_it[j], a[j]));
      if (LIVE_IMAGE) {
        s_it = fma4(w_it[j], a[j], s_it);
        s_ii = fma4(w_ii[j], c[j], s_ii);
        st4(oi + (int64_t)(n + j) * D, entity_out(w_ti[j], w_ii[j], c[j]));
      }
    }
  }
  for (; n < N; ++n) {
    const float4 a = ld4(pt + (int64_t)n * D), c = ld4(pi + (int64_t)n * D);
    s_tt = fma4(e_tt[n], a, s_tt);
    s_ti = fma4(e_ti[n], c, s_ti);
    st4(ot + (int64_t)n * D, entity_out(e_tt[n], e_it[n], a));
    if (LIVE_IMAGE) {
      s_it = fma4(e_it[n], a, s_it);
      s_ii = fma4(e_ii[n], c, s_ii);
      st4(oi + (int64_t)n * D, entity_out(e_ti[n], e_ii[n], c));
    }
  }
  const float cnt = (float)N;
  auto mention_out = [&](const float4& s1, const float4& s2, const float4& u) {
    float4 r;
    r.x = (s1.x / cnt + s2.x / cnt) + u.x;
    r.y = (s1.y / cnt + s2.y / cnt) + u.y;
    r.z = (s1.z / cnt + s2.z / cnt) + u.z;
    r.w = (s1.w / cnt + s2.w / cnt) + u.w;
    return r;
  };
  st4(agg_m + b * D + (int64_t)c4 * 4, mention_out(s_tt, s_ti, mt));
  if (LIVE_IMAGE) st4(agg_m + BD + b * D + (int64_t)c4 * 4, mention_out(s_it, s_ii, mi));
}

int launch_layer_aggregate(const float* e, int64_t edge_stride, const float* vm, const float* ve, float* agg_m,
                           float* agg_e, int B, int N, int D, bool live_image, hipStream_t st) {
  if (B <= 0 || N <= 0) return DRIN_OK;
  if (D % 4) {
    set_error("layer_aggregate: D=%d must be a multiple of 4", D);
    return DRIN_E_SHAPE;
  }
  if (B > 65535) {
    set_error("layer_aggregate: batch %d exceeds the grid limit of 65535 mentions per launch", B);
    return DRIN_E_SHAPE;
  }
  dim3 grid((unsigned)cdiv(D / 4, 64), (unsigned)B);
  KernelTimer timer(DRIN_KC_GCN, st);
  if (live_image)
    hipLaunchKernelGGL(k_layer_aggregate<true>, grid, dim3(64), 0, st, e, edge_stride, vm, ve, agg_m, agg_e, B, N, D / 4);
  else
    hipLaunchKernelGGL(k_layer_aggregate<false>, grid, dim3(64), 0, st, e, edge_stride, vm, ve, agg_m, agg_e, B, N, D / 4);
  DRIN_CHECK_LAUNCH("k_layer_aggregate");
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

// (a second row segment h2 / y2 / ... of rows2 rows may follow the first: the mention and the entity vertices of a layer
//  share the LayerNorm, model.py:128 - one launch for both)
__global__ void __launch_bounds__(256) k_layernorm_gelu(const float* h, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* y,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                        int64_t rows, int D4, float eps, const float* h2, float* y2,
                                                        float* __restrict__ mean2, float* __restrict__ rstd2, int64_t rows2,
                                                        int act) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows + rows2) return;
  if (row >= rows) {
    row -= rows;
    h = h2, y = y2, mean_out = mean2, rstd_out = rstd2;
  }
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
      r.x = act_apply(act, (x[j].x - mu) * rstd * g.x + bt.x);
      r.y = act_apply(act, (x[j].y - mu) * rstd * g.y + bt.y);
      r.z = act_apply(act, (x[j].z - mu) * rstd * g.z + bt.z);
      r.w = act_apply(act, (x[j].w - mu) * rstd * g.w + bt.w);
      st4(yr + c4 * 4, r);
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mu;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

int launch_layernorm_gelu(const float* h, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          int64_t rows, int D, float eps, hipStream_t st, int act) {
  if (rows <= 0) return DRIN_OK;
  if (D % 4 || D > 256 * LN_MAXV) {
    set_error("layernorm_gelu: D=%d must be a multiple of 4 and <= %d", D, 256 * LN_MAXV);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_layernorm_gelu, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, h, gamma, beta, y, mean, rstd,
                     rows, D / 4, eps, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (int64_t)0, act);
  DRIN_CHECK_LAUNCH("k_layernorm_gelu");
  return DRIN_OK;
}

int launch_layernorm_gelu2(const float* h, float* y, float* mean, float* rstd, int64_t rows, const float* h2, float* y2,
                           float* mean2, float* rstd2, int64_t rows2, const float* gamma, const float* beta, int D, float eps,
                           hipStream_t st, int act) {
  if (rows + rows2 <= 0) return DRIN_OK;
  if (D % 4 || D > 256 * LN_MAXV) {
    set_error("layernorm_gelu: D=%d must be a multiple of 4 and <= %d", D, 256 * LN_MAXV);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_layernorm_gelu, dim3((unsigned)cdiv(rows + rows2, 4)), dim3(256), 0, st, h, gamma, beta, y, mean, rstd,
                     rows, D / 4, eps, h2, y2, mean2, rstd2, rows2, act);
  DRIN_CHECK_LAUNCH("k_layernorm_gelu");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// dynamic edge (model.py:148-153 with w_m = Identity :112, sigmoid :133):
//   out[b, n] = sigmoid(mean_d(fu[b, :] * fv[b, n, :]) + e[b, n])       one wave per pair,
// all four edge types of a layer in one pass: fu = [2][B][D] (W_u of mt, mi), fv = [2][M][D] (W_v of et, ei),
// e / out = [4][M] in the order tt, ti, it, ii.  Each fv row is read once instead of twice; the per-lane summation
// order is fixed, so the results do not depend on the launch geometry.
__global__ void __launch_bounds__(256) k_edge_update4(const float* __restrict__ fu, const float* __restrict__ fv,
                                                      const float* __restrict__ e, float* __restrict__ out,
                                                      int64_t pairs, int B, int N, int D4, int act, float* __restrict__ z_out) {
  int64_t p, b;
  if (!wave_pair(pairs, N, p, b)) return;
  const int lane = threadIdx.x & 63;
  const int64_t D = (int64_t)D4 * 4;
  const float* ut = fu + b * D;
  const float* ui = ut + (int64_t)B * D;
  const float* vt = fv + p * D;
  const float* vi = vt + pairs * D;
  float s_tt = 0.f, s_ti = 0.f, s_it = 0.f, s_ii = 0.f;
  for (int c4 = lane; c4 < D4; c4 += 64) {
    const float4 a = ld4(ut + c4 * 4), b = ld4(ui + c4 * 4), x = ld4(vt + c4 * 4), y = ld4(vi + c4 * 4);
    s_tt += dot4(a, x);
    s_ti += dot4(a, y);
    s_it += dot4(b, x);
    s_ii += dot4(b, y);
  }
  s_tt = wave_sum(s_tt);
  s_ti = wave_sum(s_ti);
  s_it = wave_sum(s_it);
  s_ii = wave_sum(s_ii);
  if (lane == 0) {
    const float d = (float)(D4 * 4);
    const float z0 = s_tt / d + e[p], z1 = s_ti / d + e[pairs + p], z2 = s_it / d + e[2 * pairs + p], z3 = s_ii / d + e[3 * pairs + p];
    out[p] = act_apply(act, z0);
    out[pairs + p] = act_apply(act, z1);
    out[2 * pairs + p] = act_apply(act, z2);
    out[3 * pairs + p] = act_apply(act, z3);
    if (z_out != nullptr) {  // an activation whose derivative cannot be taken from its output (gelu, silu): backward reads z
      z_out[p] = z0, z_out[pairs + p] = z1, z_out[2 * pairs + p] = z2, z_out[3 * pairs + p] = z3;
    }
  }
}

int launch_edge_update4(const float* fu, const float* fv, const float* e, float* out, int B, int N, int D,
                        hipStream_t st, int act, float* z_out) {
  const int64_t pairs = (int64_t)B * N;
  if (pairs <= 0) return DRIN_OK;
  if (D % 4) {
    set_error("edge_update: D=%d must be a multiple of 4", D);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_edge_update4, pair_grid(B, N), dim3(256), 0, st, fu, fv, e, out, pairs, B, N, D / 4, act, z_out);
  DRIN_CHECK_LAUNCH("k_edge_update4");
  return DRIN_OK;
}

}  // namespace drin
