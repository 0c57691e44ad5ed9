// gcn_edge_feature == "vector" (drin/model.py:112-116,140-141,151-152,202): every edge is a D-vector per
// (mention, candidate) pair instead of a scalar.  The ablation runs on the layer-by-layer path; these are its
// element-wise / reduction kernels (forward and backward).  All are HBM/L2-bound row kernels on float4 columns;
// the W_m / W_u / W_v contractions use the common GEMM kernels.
#include "device_utils.h"
#include "internal.h"

namespace drin {

// out[k][p][:] = es[k][p]      (model.py:202: e.unsqueeze(-1).expand(-1, -1, D))
__global__ void __launch_bounds__(256) k_expand_edges(const float* __restrict__ es, float* __restrict__ out,
                                                      int64_t total4, int D4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const float v = es[i / D4];
  st4(out + i * 4, make_float4(v, v, v, v));
}

int launch_expand_edges(const float* es, float* out, int64_t pairs4, int D, hipStream_t st) {
  const int64_t total4 = pairs4 * (D / 4);
  if (total4 <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_EDGE, st);
  hipLaunchKernelGGL(k_expand_edges, dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, st, es, out, total4, D / 4);
  DRIN_CHECK_LAUNCH("k_expand_edges");
  return DRIN_OK;
}

// out[b, :] = scale (sum_n w1[p, :] v1[p, :] + sum_n w2[p, :] v2[p, :]) + u[b, :]        (v2 / u optional)
// forward: mention <- entity (model.py:143-144, scale = 1/N); backward: its transpose (scale = 1).
__global__ void __launch_bounds__(64) k_mention_reduce_vec(const float* __restrict__ w1, const float* __restrict__ v1,
                                                           const float* __restrict__ w2, const float* __restrict__ v2,
                                                           const float* __restrict__ u, float* __restrict__ out, int N,
                                                           int D4, float scale, int mean_style) {
  const int c4 = blockIdx.x * 64 + threadIdx.x;
  if (c4 >= D4) return;
  const int64_t b = blockIdx.y;
  const int64_t off = (b * N) * (int64_t)D4 * 4 + (int64_t)c4 * 4;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int n = 0; n < N; ++n) {
    const int64_t o = off + (int64_t)n * D4 * 4;
    const float4 a = ld4(w1 + o), x = ld4(v1 + o);
    s1 = make_float4(fmaf(a.x, x.x, s1.x), fmaf(a.y, x.y, s1.y), fmaf(a.z, x.z, s1.z), fmaf(a.w, x.w, s1.w));
    if (v2 != nullptr) {
      const float4 c = ld4(w2 + o), y = ld4(v2 + o);
      s2 = make_float4(fmaf(c.x, y.x, s2.x), fmaf(c.y, y.y, s2.y), fmaf(c.z, y.z, s2.z), fmaf(c.w, y.w, s2.w));
    }
  }
  float4 r;
  if (mean_style) {  // (s1 / N + s2 / N) + u, the reference's order
    const float cnt = (float)N;
    r = make_float4(s1.x / cnt + s2.x / cnt, s1.y / cnt + s2.y / cnt, s1.z / cnt + s2.z / cnt, s1.w / cnt + s2.w / cnt);
  } else {
    r = (s1 + s2) * scale;
  }
  if (u != nullptr) r = r + ld4(u + b * (int64_t)D4 * 4 + (int64_t)c4 * 4);
  st4(out + b * (int64_t)D4 * 4 + (int64_t)c4 * 4, r);
}

int launch_mention_reduce_vec(const float* w1, const float* v1, const float* w2, const float* v2, const float* u,
                              float* out, int B, int N, int D, float scale, bool mean_style, hipStream_t st) {
  if (B <= 0) return DRIN_OK;
  for (int b0 = 0; b0 < B; b0 += 65535) {
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    const int64_t vo = (int64_t)b0 * N * D;
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_mention_reduce_vec, dim3((unsigned)cdiv(D / 4, 64), (unsigned)nb), dim3(64), 0, st, w1 + vo,
                       v1 + vo, w2 ? w2 + vo : nullptr, v2 ? v2 + vo : nullptr, u ? u + (int64_t)b0 * D : nullptr,
                       out + (int64_t)b0 * D, N, D / 4, scale, mean_style ? 1 : 0);
    DRIN_CHECK_LAUNCH("k_mention_reduce_vec");
  }
  return DRIN_OK;
}

// out[p, :] = e1[p, :] m1[b, :] + e2[p, :] m2[b, :] + v[p, :]          (model.py:146 + :128 self term)
__global__ void __launch_bounds__(256) k_entity_aggregate_vec(const float* __restrict__ e1, const float* __restrict__ m1,
                                                              const float* __restrict__ e2, const float* __restrict__ m2,
                                                              const float* __restrict__ v, float* __restrict__ out,
                                                              int64_t total4, int N, int D4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int64_t p = i / D4;
  const int c4 = (int)(i - p * D4);
  const int64_t b = p / N;
  const float4 a1 = ld4(e1 + i * 4), a2 = ld4(e2 + i * 4), vv = ld4(v + i * 4);
  const float4 x1 = ld4(m1 + b * (int64_t)D4 * 4 + c4 * 4), x2 = ld4(m2 + b * (int64_t)D4 * 4 + c4 * 4);
  float4 r;
  r.x = (a1.x * x1.x + a2.x * x2.x) + vv.x;
  r.y = (a1.y * x1.y + a2.y * x2.y) + vv.y;
  r.z = (a1.z * x1.z + a2.z * x2.z) + vv.z;
  r.w = (a1.w * x1.w + a2.w * x2.w) + vv.w;
  st4(out + i * 4, r);
}

int launch_entity_aggregate_vec(const float* e1, const float* m1, const float* e2, const float* m2, const float* v,
                                float* out, int B, int N, int D, hipStream_t st) {
  const int64_t total4 = (int64_t)B * N * (D / 4);
  if (total4 <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_entity_aggregate_vec, dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, st, e1, m1, e2, m2, v, out,
                     total4, N, D / 4);
  DRIN_CHECK_LAUNCH("k_entity_aggregate_vec");
  return DRIN_OK;
}

// pre[k][p][:] = cat(fu[u(k)][b][: D/2], fv[v(k)][p][: D/2]) + e[k][p][:]      (model.py:150-152, :133 input)
// edge k = (u, v) with u = k >> 1 (mention text / image), v = k & 1 (entity text / image)  (model.py:107)
__global__ void __launch_bounds__(256) k_edge_pre_vec(const float* __restrict__ fu, const float* __restrict__ fv,
                                                      const float* __restrict__ e, float* __restrict__ pre, int64_t M,
                                                      int B, int N, int D4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over [4][M][D4]
  const int64_t total4 = 4 * M * D4;
  if (i >= total4) return;
  const int c4 = (int)(i % D4);
  const int64_t kp = i / D4;
  const int k = (int)(kp / M);
  const int64_t p = kp - (int64_t)k * M;
  const int64_t b = p / N;
  const int H4 = D4 / 2;
  float4 c;
  if (c4 < H4)
    c = ld4(fu + ((int64_t)(k >> 1) * B + b) * H4 * 4 + c4 * 4);
  else
    c = ld4(fv + ((int64_t)(k & 1) * M + p) * H4 * 4 + (c4 - H4) * 4);
  st4(pre + i * 4, c + ld4(e + i * 4));
}

int launch_edge_pre_vec(const float* fu, const float* fv, const float* e, float* pre, int B, int N, int D,
                        hipStream_t st) {
  const int64_t M = (int64_t)B * N, total4 = 4 * M * (D / 4);
  if (total4 <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_edge_pre_vec, dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, st, fu, fv, e, pre, M, B, N, D / 4);
  DRIN_CHECK_LAUNCH("k_edge_pre_vec");
  return DRIN_OK;
}

// x = sigmoid(x) in place
__global__ void __launch_bounds__(256) k_sigmoid_inplace(float* __restrict__ x, int64_t n4, int act, float* __restrict__ z_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = ld4(x + i * 4);
  if (z_out != nullptr) st4(z_out + i * 4, v);   // the pre-activation, for activations without a derivative-from-output
  st4(x + i * 4, make_float4(act_apply(act, v.x), act_apply(act, v.y), act_apply(act, v.z), act_apply(act, v.w)));
}

int launch_sigmoid_inplace(float* x, int64_t n, hipStream_t st, int act, float* z_out) {
  if (n <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_sigmoid_inplace, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, st, x, n / 4, act, z_out);
  DRIN_CHECK_LAUNCH("k_sigmoid_inplace");
  return DRIN_OK;
}

// Backward of k_edge_pre_vec w.r.t. fu / fv (the +e branch is a plain pass-through handled by the caller):
//   dfv[v][p][j] = sum_{k: k & 1 == v} dpre[k][p][D/2 + j]
//   dfu[u][b][j] = sum_n sum_{k: k >> 1 == u} dpre[k][p][j]
__global__ void __launch_bounds__(256) k_edge_pre_vec_bwd_fv(const float* __restrict__ dpre, float* __restrict__ dfv,
                                                             int64_t M, int D4) {
  const int H4 = D4 / 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over [2][M][H4]
  if (i >= 2 * M * H4) return;
  const int c4 = (int)(i % H4);
  const int64_t vp = i / H4;
  const int v = (int)(vp / M);
  const int64_t p = vp - (int64_t)v * M;
  const float4 a = ld4(dpre + (((int64_t)v * M + p) * D4 + H4 + c4) * 4);        // k = v      (u = 0)
  const float4 c = ld4(dpre + (((int64_t)(v + 2) * M + p) * D4 + H4 + c4) * 4);  // k = v + 2  (u = 1)
  st4(dfv + i * 4, a + c);
}

__global__ void __launch_bounds__(64) k_edge_pre_vec_bwd_fu(const float* __restrict__ dpre, float* __restrict__ dfu,
                                                            int64_t M, int B, int N, int D4) {
  const int H4 = D4 / 2;
  const int c4 = blockIdx.x * 64 + threadIdx.x;
  if (c4 >= H4) return;
  const int64_t b = blockIdx.y;
  const int u = blockIdx.z;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = 0; n < N; ++n) {
    const int64_t p = b * N + n;
    s = s + ld4(dpre + (((int64_t)(2 * u) * M + p) * D4 + c4) * 4) + ld4(dpre + (((int64_t)(2 * u + 1) * M + p) * D4 + c4) * 4);
  }
  st4(dfu + (((int64_t)u * B + b) * H4 + c4) * 4, s);
}

int launch_edge_pre_vec_bwd(const float* dpre, float* dfu, float* dfv, int B, int N, int D, hipStream_t st) {
  const int64_t M = (int64_t)B * N;
  if (M <= 0) return DRIN_OK;
  if (B > 65535) {
    set_error("edge_pre_vec_bwd: batch %d exceeds the grid limit", B);
    return DRIN_E_SHAPE;
  }
  const int D4 = D / 4;
  {
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_edge_pre_vec_bwd_fv, dim3((unsigned)cdiv(2 * M * (D4 / 2), 256)), dim3(256), 0, st, dpre, dfv, M,
                       D4);
    DRIN_CHECK_LAUNCH("k_edge_pre_vec_bwd_fv");
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_edge_pre_vec_bwd_fu, dim3((unsigned)cdiv(D4 / 2, 64), (unsigned)B, 2), dim3(64), 0, st, dpre, dfu,
                     M, B, N, D4);
  DRIN_CHECK_LAUNCH("k_edge_pre_vec_bwd_fu");
  return DRIN_OK;
}

// Entity side of the aggregation backward with vector edges (transposes of model.py:143-146), element-wise:
//   d_et = dA_et + (e_tt dA_mt[b] + e_it dA_mi[b]) / N          d_ei = dA_ei + (e_ti dA_mt[b] + e_ii dA_mi[b]) / N
//   de_tt = dA_mt[b] et / N + dA_et mt[b]    de_ti = dA_mt[b] ei / N + dA_ei mt[b]
//   de_it = dA_mi[b] et / N + dA_et mi[b]    de_ii = dA_mi[b] ei / N + dA_ei mi[b]
// de_k additionally receives de_extra[k] and is multiplied by the edge switch m_k.  dA_mi / dA_ei may be NULL.
__global__ void __launch_bounds__(256)
    k_entity_side_bwd_vec(const float* __restrict__ dA_mt, const float* __restrict__ dA_mi,
                          const float* __restrict__ dA_et, const float* __restrict__ dA_ei, const float* __restrict__ mt,
                          const float* __restrict__ mi, const float* __restrict__ et, const float* __restrict__ ei,
                          const float* __restrict__ e, const float* __restrict__ de_extra, float* __restrict__ d_et,
                          float* __restrict__ d_ei, float* __restrict__ de, int64_t M, int N, int D4, float m0, float m1,
                          float m2, float m3) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over [M][D4]
  if (i >= M * D4) return;
  const int64_t p = i / D4;
  const int c4 = (int)(i - p * D4);
  const int64_t b = p / N, mo = (b * D4 + c4) * 4, eo = i * 4, ES = M * (int64_t)D4 * 4;
  const float inv_n = 1.0f / (float)N;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 gmt = ld4(dA_mt + mo), gmi = dA_mi ? ld4(dA_mi + mo) : z;
  const float4 get = ld4(dA_et + eo), gei = dA_ei ? ld4(dA_ei + eo) : z;
  const float4 vmt = ld4(mt + mo), vmi = ld4(mi + mo), vet = ld4(et + eo), vei = ld4(ei + eo);
  const float4 e_tt = ld4(e + eo), e_ti = ld4(e + ES + eo), e_it = ld4(e + 2 * ES + eo), e_ii = ld4(e + 3 * ES + eo);
  auto mul = [](float4 a, float4 c) { return make_float4(a.x * c.x, a.y * c.y, a.z * c.z, a.w * c.w); };
  st4(d_et + eo, get + (mul(e_tt, gmt) + mul(e_it, gmi)) * inv_n);
  st4(d_ei + eo, gei + (mul(e_ti, gmt) + mul(e_ii, gmi)) * inv_n);
  const float4 x0 = de_extra ? ld4(de_extra + eo) : z, x1 = de_extra ? ld4(de_extra + ES + eo) : z;
  const float4 x2 = de_extra ? ld4(de_extra + 2 * ES + eo) : z, x3 = de_extra ? ld4(de_extra + 3 * ES + eo) : z;
  st4(de + eo, (mul(gmt, vet) * inv_n + mul(get, vmt) + x0) * m0);
  st4(de + ES + eo, (mul(gmt, vei) * inv_n + mul(gei, vmt) + x1) * m1);
  st4(de + 2 * ES + eo, (mul(gmi, vet) * inv_n + mul(get, vmi) + x2) * m2);
  st4(de + 3 * ES + eo, (mul(gmi, vei) * inv_n + mul(gei, vmi) + x3) * m3);
}

int launch_entity_side_bwd_vec(const float* dA_mt, const float* dA_mi, const float* dA_et, const float* dA_ei,
                               const float* mt, const float* mi, const float* et, const float* ei, const float* e,
                               const float* de_extra, float* d_et, float* d_ei, float* de, int B, int N, int D,
                               const float* mask, hipStream_t st) {
  const int64_t M = (int64_t)B * N;
  if (M <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_entity_side_bwd_vec, dim3((unsigned)cdiv(M * (D / 4), 256)), dim3(256), 0, st, dA_mt, dA_mi, dA_et,
                     dA_ei, mt, mi, et, ei, e, de_extra, d_et, d_ei, de, M, N, D / 4, mask[0], mask[1], mask[2], mask[3]);
  DRIN_CHECK_LAUNCH("k_entity_side_bwd_vec");
  return DRIN_OK;
}

}  // namespace drin
