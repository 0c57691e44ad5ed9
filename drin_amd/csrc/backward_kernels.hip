// Backward of the row kernels of the path (loss.backward() through drin/model.py:121-153,207-209).
// Same style as the forward: one wave per row / pair, 16-byte lane accesses, shuffle reductions;
// column sums that feed bias / LayerNorm gradients are accumulated per lane across a grid-stride
// row loop, combined across the block's waves in LDS, stored as one partial row per block and added
// to the caller's gradient buffer in block order by a second launch (no atomics: reproducible bits).
#include "device_utils.h"
#include "internal.h"

namespace drin {

constexpr int MAXV = 4;  // float4 columns per lane: D <= 1024
constexpr int kLnBwdMaxBlocks = 1024;

// ------------------------------------------------------------------------------------------------
// score = cos(x[b], y[p]) (model.py:207-209), torch>=2 form: xn = x / max(|x|, eps), yn likewise.
//   dy[p]   = g (xn - [|y|>eps] c yn) / ny
//   coef[p] = g / ny,  gc[p] = g c,  xnorm[p] = |x[b]|          (consumed by the mention-side kernel)
__global__ void __launch_bounds__(256) k_cosine_bwd_entity(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ g, float* __restrict__ dy,
                                                           float* __restrict__ coef, float* __restrict__ gc,
                                                           float* __restrict__ xnorm, int64_t pairs, int N, int D4,
                                                           float eps) {
  int64_t p, b;
  if (!wave_pair(pairs, N, p, b)) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + b * (int64_t)D4 * 4;
  const float* yr = y + p * (int64_t)D4 * 4;
  float4 xv[MAXV], yv[MAXV];
  float xy = 0.f, xx = 0.f, yy = 0.f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < D4) {
      xv[j] = ld4(xr + c4 * 4);
      yv[j] = ld4(yr + c4 * 4);
    } else {
      xv[j] = yv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    xy += dot4(xv[j], yv[j]);
    xx += dot4(xv[j], xv[j]);
    yy += dot4(yv[j], yv[j]);
  }
  xy = wave_sum(xy);
  xx = wave_sum(xx);
  yy = wave_sum(yy);
  const float nxr = sqrtf(xx), nyr = sqrtf(yy);
  const float nx = fmaxf(nxr, eps), ny = fmaxf(nyr, eps);
  const float c = xy / (nx * ny);
  const float gg = g[p];
  const float a = gg / (nx * ny);                         // multiplies x
  const float bcoef = nyr > eps ? gg * c / (ny * ny) : 0.f;  // multiplies y
  float* dr = dy + p * (int64_t)D4 * 4;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < D4) {
      float4 r;
      r.x = a * xv[j].x - bcoef * yv[j].x;
      r.y = a * xv[j].y - bcoef * yv[j].y;
      r.z = a * xv[j].z - bcoef * yv[j].z;
      r.w = a * xv[j].w - bcoef * yv[j].w;
      st4(dr + c4 * 4, r);
    }
  }
  if (lane == 0) {
    coef[p] = gg / ny;
    gc[p] = gg * c;
    xnorm[p] = nxr;
  }
}

//   dx[b] = (sum_n coef[p] y[p] - [|x|>eps] xn[b] sum_n gc[p]) / nx
// One workgroup per (mention, 64 float4 columns); its four waves split the candidate loop (n = w, w + 4, ...: a single wave
// walking 101 dependent 3 KB rows was 26 us at 64 mentions) and are combined through LDS in a fixed order.
__global__ void __launch_bounds__(256) k_cosine_bwd_mention(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ coef,
                                                            const float* __restrict__ gc,
                                                            const float* __restrict__ xnorm, float* __restrict__ dx,
                                                            int N, int D4, float eps) {
  __shared__ float4 part[4][64];
  __shared__ float part_g[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c4 = blockIdx.x * 64 + lane;
  const int64_t b = blockIdx.y;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float sg = 0.f;
  if (c4 < D4) {
    const float* yp = y + (b * N) * (int64_t)D4 * 4 + (int64_t)c4 * 4;
    for (int n = wave; n < N; n += 4) {
      s = fma4(coef[b * N + n], ld4(yp + (int64_t)n * D4 * 4), s);
      sg += gc[b * N + n];
    }
  }
  part[wave][lane] = s;
  part_g[wave][lane] = sg;
  __syncthreads();
  if (wave != 0 || c4 >= D4) return;
  s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  sg = (part_g[0][lane] + part_g[1][lane]) + (part_g[2][lane] + part_g[3][lane]);
  const float nxr = xnorm[b * N];
  const float nx = fmaxf(nxr, eps);
  const float4 xv = ld4(x + b * (int64_t)D4 * 4 + (int64_t)c4 * 4);
  const float k = nxr > eps ? sg / (nx * nx) : 0.f;
  float4 r;
  r.x = s.x / nx - k * xv.x;
  r.y = s.y / nx - k * xv.y;
  r.z = s.z / nx - k * xv.z;
  r.w = s.w / nx - k * xv.w;
  st4(dx + b * (int64_t)D4 * 4 + (int64_t)c4 * 4, r);
}

int launch_cosine_bwd(const float* x, const float* y, const float* g, float* dx, float* dy, float* scratch3,
                      int B, int N, int D, float eps, hipStream_t st) {
  const int64_t pairs = (int64_t)B * N;
  if (pairs <= 0) return DRIN_OK;
  float* coef = scratch3;
  float* gcv = scratch3 + pairs;
  float* xn = scratch3 + 2 * pairs;
  {
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_cosine_bwd_entity, pair_grid(B, N), dim3(256), 0, st, x, y, g, dy, coef, gcv,
                       xn, pairs, N, D / 4, eps);
    DRIN_CHECK_LAUNCH("k_cosine_bwd_entity");
  }
  for (int b0 = 0; b0 < B; b0 += 65535) {
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    const int64_t po = (int64_t)b0 * N;
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_cosine_bwd_mention, dim3((unsigned)cdiv(D / 4, 64), (unsigned)nb), dim3(256), 0, st,
                       x + (int64_t)b0 * D, y + po * D, coef + po, gcv + po, xn + po, dx + (int64_t)b0 * D, N, D / 4,
                       eps);
    DRIN_CHECK_LAUNCH("k_cosine_bwd_mention");
  }
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of y = gelu(LN(h)) (model.py:128), in place: on entry `g` holds dL/dy, on exit dL/dh.
//   z = xhat gamma + beta, dz = g gelu'(z), dxhat = dz gamma,
//   dh = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))
// and the three column sums  dgamma += sum_rows dz xhat,  dbeta += sum_rows dz,  dbias += sum_rows dh
// (dbias is the W_h bias gradient: h = a W_h^T + b_h).
// V = float4 columns per lane (3 for D = 768).  The next row's h / g are requested before the current row's
// erf / exp chain starts: the kernel is half memory (3 rows of traffic per row), half VALU, and four resident waves
// per SIMD did not overlap the two on their own (2 x 51 712 rows: 257 us without the prefetch).
// one row segment of the LayerNorm backward: pre-LayerNorm values, saved statistics, gradient (in / out)
struct LnBwdSeg {
  const float* h;
  const float* mean;
  const float* rstd;
  float* g;
  int64_t rows;
};

template <int V>
__global__ void __launch_bounds__(256) k_layernorm_gelu_bwd(const LnBwdSeg sa, const LnBwdSeg sb,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ partial, int D4, int act) {
  // rows of segment a, then rows of segment b (the mention vertices and the entity vertices of a layer share W_h and the
  // LayerNorm, model.py:128: one launch and one set of column sums for both)
  const int64_t rows = sa.rows + sb.rows;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 gm[V], bt[V];
  float4 a_dg[V], a_db[V], a_dh[V];
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c4 = lane + 64 * j;
    gm[j] = c4 < D4 ? ld4(gamma + c4 * 4) : zero;
    bt[j] = c4 < D4 ? ld4(beta + c4 * 4) : zero;
    a_dg[j] = a_db[j] = a_dh[j] = zero;
  }
  const float inv_d = 1.0f / (float)(D4 * 4);
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  float4 hv[V], gv[V];
  float mu = 0.f, rs = 0.f;
  auto fetch = [&](int64_t r, float4* ph, float4* pg, float& m, float& s) {
    const bool first = r < sa.rows;
    const int64_t lr = first ? r : r - sa.rows;
    const float* hr = (first ? sa.h : sb.h) + lr * (int64_t)D4 * 4;
    const float* gr = (first ? sa.g : sb.g) + lr * (int64_t)D4 * 4;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int c4 = lane + 64 * j;
      ph[j] = c4 < D4 ? ld4(hr + c4 * 4) : zero;
      pg[j] = c4 < D4 ? ld4(gr + c4 * 4) : zero;
    }
    m = (first ? sa.mean : sb.mean)[lr];
    s = (first ? sa.rstd : sb.rstd)[lr];
  };
  if (row < rows) fetch(row, hv, gv, mu, rs);
  for (; row < rows; row += stride) {
    float4 nh[V], ng[V];
    float nmu = 0.f, nrs = 0.f;
    if (row + stride < rows) fetch(row + stride, nh, ng, nmu, nrs);
    float* gr = (row < sa.rows ? sa.g + row * (int64_t)D4 * 4 : sb.g + (row - sa.rows) * (int64_t)D4 * 4);
    float4 xh[V], dxh[V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < D4) {
        xh[j] = make_float4((hv[j].x - mu) * rs, (hv[j].y - mu) * rs, (hv[j].z - mu) * rs, (hv[j].w - mu) * rs);
        float4 dz;
        dz.x = gv[j].x * act_grad(act, xh[j].x * gm[j].x + bt[j].x);
        dz.y = gv[j].y * act_grad(act, xh[j].y * gm[j].y + bt[j].y);
        dz.z = gv[j].z * act_grad(act, xh[j].z * gm[j].z + bt[j].z);
        dz.w = gv[j].w * act_grad(act, xh[j].w * gm[j].w + bt[j].w);
        a_dg[j] = make_float4(fmaf(dz.x, xh[j].x, a_dg[j].x), fmaf(dz.y, xh[j].y, a_dg[j].y),
                              fmaf(dz.z, xh[j].z, a_dg[j].z), fmaf(dz.w, xh[j].w, a_dg[j].w));
        a_db[j] = a_db[j] + dz;
        dxh[j] = make_float4(dz.x * gm[j].x, dz.y * gm[j].y, dz.z * gm[j].z, dz.w * gm[j].w);
        s1 += (dxh[j].x + dxh[j].y) + (dxh[j].z + dxh[j].w);
        s2 += dot4(dxh[j], xh[j]);
      } else {
        xh[j] = dxh[j] = zero;
      }
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < D4) {
        float4 dh;
        dh.x = rs * (dxh[j].x - s1 - xh[j].x * s2);
        dh.y = rs * (dxh[j].y - s1 - xh[j].y * s2);
        dh.z = rs * (dxh[j].z - s1 - xh[j].z * s2);
        dh.w = rs * (dxh[j].w - s1 - xh[j].w * s2);
        a_dh[j] = a_dh[j] + dh;
        st4(gr + c4 * 4, dh);
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      hv[j] = nh[j];
      gv[j] = ng[j];
    }
    mu = nmu;
    rs = nrs;
  }
  // combine the four waves' column sums through LDS (one quantity at a time keeps LDS at 16 KiB), then ONE row of
  // partial sums per block: partial[block][q][D].  A second kernel adds the blocks in order - no atomics: a thousand
  // blocks adding to the same 2 304 addresses serialised in L2 (106 us per call at 12 928 rows), and the sums were
  // run-order dependent.
  __shared__ float4 comb[4][V][64];
  const int D = D4 * 4;
  for (int q = 0; q < 3; ++q) {
    float* dst = partial + ((int64_t)blockIdx.x * 3 + q) * D;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < V; ++j) comb[wave][j][lane] = q == 0 ? a_dg[j] : (q == 1 ? a_db[j] : a_dh[j]);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < D4) st4(dst + c4 * 4, (comb[0][j][lane] + comb[1][j][lane]) + (comb[2][j][lane] + comb[3][j][lane]));
      }
    }
  }
}

// Sum of per-block partial rows, in two levels so that every load is independent of the others:
//   level 1 (out_partial != NULL): out_partial[z][q][c] = sum of the 64 rows z*64 .. of partial[.][q][c]
//   level 2 (out_partial == NULL): dst_q[c] += sum of all `blocks` (<= 64) rows      (q: dgamma, dbeta, dbias)
// Block = 4 waves x 64 columns; wave w takes rows w, w + 4, ... of its slice (at most 16, all in flight at once);
// the waves are combined through LDS in order.
// (q_major: level-1 rows laid out [q][z][D] instead of [z][q][D] - one contiguous [z][D] slice array per quantity, the form
//  the ordered slice sum at the end of the backward pass takes: level 2 then rides in that launch, internal.h SliceSum)
__global__ void __launch_bounds__(256) k_sum_partials(const float* __restrict__ partial, int blocks, int D,
                                                      float* __restrict__ out_partial, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, float* __restrict__ dbias, int q_major) {
  __shared__ float comb[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + lane, q = blockIdx.y;
  const int b0 = blockIdx.z * 64, b1 = min(blocks, b0 + 64);
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int b = b0 + wave + 4 * i;
    v[i] = (b < b1 && c < D) ? partial[((int64_t)b * 3 + q) * D + c] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  comb[wave][lane] = s;
  __syncthreads();
  if (wave != 0 || c >= D) return;
  s = (comb[0][lane] + comb[1][lane]) + (comb[2][lane] + comb[3][lane]);
  if (out_partial != nullptr) {
    out_partial[(q_major ? (int64_t)q * gridDim.z + blockIdx.z : (int64_t)blockIdx.z * 3 + q) * D + c] = s;
  } else {
    float* dst = q == 0 ? dgamma : (q == 1 ? dbeta : dbias);
    if (dst != nullptr) dst[c] += s;
  }
}

// (h2 .. rows2: an optional second row segment sharing gamma / beta and the column sums.  Folding the two k_sum_partials
//  launches into this kernel with arrival tickets - last block of a group adds the group, last group adds the groups - was
//  measured and dropped: every block's device-scope release is an L2 write-back on this 8-XCD part; the B = 64 training step
//  went from 1.61 to 1.81 ms with it, same box)
int launch_layernorm_gelu_bwd2(const float* h, const float* mean, const float* rstd, float* g, int64_t rows, const float* h2,
                               const float* mean2, const float* rstd2, float* g2, int64_t rows2, const float* gamma,
                               const float* beta, float* dgamma, float* dbeta, float* dbias, float* partial, int D,
                               hipStream_t st, int act, SliceSum* defer, float* level1_own) {
  const int64_t total = rows + (h2 != nullptr ? rows2 : 0);
  if (total <= 0) return DRIN_OK;
  if (D % 4 || D > 256 * MAXV) {
    set_error("layernorm_gelu_bwd: D=%d must be a multiple of 4 and <= %d", D, 256 * MAXV);
    return DRIN_E_SHAPE;
  }
  // 161 VGPRs at V = 3: three workgroups per CU are resident, so 768 (of the 1024 the partial buffer holds) run as one round
  const int64_t cap = 768;
  const int64_t blocks = cdiv(total, 4) < cap ? cdiv(total, 4) : cap;
  KernelTimer timer(DRIN_KC_GCN, st);
  const int v = (int)cdiv(D / 4, 64);
  auto kern = v <= 1 ? k_layernorm_gelu_bwd<1> : v == 2 ? k_layernorm_gelu_bwd<2> : v == 3 ? k_layernorm_gelu_bwd<3>
                                                                                         : k_layernorm_gelu_bwd<4>;
  const LnBwdSeg sa{h, mean, rstd, g, rows};
  const LnBwdSeg sb{h2, mean2, rstd2, g2, h2 != nullptr ? rows2 : 0};
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, sa, sb, gamma, beta, partial, D / 4, act);
  DRIN_CHECK_LAUNCH("k_layernorm_gelu_bwd");
  // partial: [kLnBwdMaxBlocks][3][D] block rows, then [kLnBwdMaxBlocks / 64][3][D] for the first reduction level
  const int z = (int)cdiv(blocks, 64);
  // defer: the second level - up to 16 rows per quantity - is handed to the caller's ordered slice sum (one launch at the end
  // of the backward pass for every split reduction) instead of being a launch of its own; the level-1 rows then live in
  // `level1_own`, which nothing overwrites before that launch (one area per layer)
  const bool deferred = defer != nullptr && level1_own != nullptr && defer->n + 3 <= SliceSum::MAX_DST - 24 /* the pass's other split reductions: three collections of at most eight */ &&
                        defer->n_seg + 3 <= SliceSum::MAX_SEG && (D % 4) == 0 && aligned16(level1_own) &&
                        (dgamma == nullptr || aligned16(dgamma)) && (dbeta == nullptr || aligned16(dbeta)) &&
                        (dbias == nullptr || aligned16(dbias));
  float* level1 = deferred ? level1_own : partial + (int64_t)kLnBwdMaxBlocks * 3 * D;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)cdiv(D, 64), 3, (unsigned)z), dim3(256), 0, st, partial, (int)blocks, D,
                     level1, (float*)nullptr, (float*)nullptr, (float*)nullptr, deferred ? 1 : 0);
  DRIN_CHECK_LAUNCH("k_sum_partials");
  if (deferred) {
    float* dst[3] = {dgamma, dbeta, dbias};
    for (int q = 0; q < 3; ++q) DRIN_TRY(defer->add(dst[q], D, 1, D, level1 + (int64_t)q * z * D, z));   // no-op for a NULL destination
    return DRIN_OK;
  }
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)cdiv(D, 64), 3, 1), dim3(256), 0, st, level1, z, D, (float*)nullptr, dgamma,
                     dbeta, dbias, 0);
  DRIN_CHECK_LAUNCH("k_sum_partials");
  return DRIN_OK;
}

int launch_layernorm_gelu_bwd(const float* h, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, float* g, float* dgamma, float* dbeta, float* dbias, float* partial,
                              int64_t rows, int D, hipStream_t st, int act) {
  return launch_layernorm_gelu_bwd2(h, mean, rstd, g, rows, nullptr, nullptr, nullptr, nullptr, 0, gamma, beta, dgamma, dbeta,
                                    dbias, partial, D, st, act, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// out[c] += sum_rows x[row, c]      (bias gradients of W_u / W_v / the vertex encoders)
// Workgroup (x, y) sums the rows y, y + gridDim.y, ... of 256 columns (four waves interleaved, combined through LDS in a
// fixed order) and stores the sums as row y of `partial` ([gridDim.y][C]); a SliceSum launch adds the rows to out in
// order.  partial == NULL (one workgroup per 256 columns, gridDim.y == 1): added to out in place.  No atomics.
__device__ __forceinline__ void colsum_block(const float* __restrict__ x, float* __restrict__ out, float* __restrict__ partial,
                                             int64_t rows, int C4, int by, float4 (*comb)[64]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c4 = blockIdx.x * 64 + lane;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < C4) {
    const int64_t step = (int64_t)by * 4;
    const float* col = x + (int64_t)c4 * 4;
    int64_t row = (int64_t)blockIdx.y * 4 + wave;
    for (; row + 3 * step < rows; row += 4 * step) {  // four independent loads in flight per wave
      const float4 v0 = ld4(col + row * (int64_t)C4 * 4), v1 = ld4(col + (row + step) * (int64_t)C4 * 4);
      const float4 v2 = ld4(col + (row + 2 * step) * (int64_t)C4 * 4), v3 = ld4(col + (row + 3 * step) * (int64_t)C4 * 4);
      acc = acc + ((v0 + v1) + (v2 + v3));
    }
    for (; row < rows; row += step) acc = acc + ld4(col + row * (int64_t)C4 * 4);
  }
  comb[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c4 < C4) {
    const float4 t = (comb[0][lane] + comb[1][lane]) + (comb[2][lane] + comb[3][lane]);
    if (partial != nullptr) {
      st4(partial + ((int64_t)blockIdx.y * C4 + c4) * 4, t);
    } else {
      st4(out + c4 * 4, ld4(out + c4 * 4) + t);
    }
  }
}

__global__ void __launch_bounds__(256) k_colsum(const float* __restrict__ x, float* __restrict__ out, float* __restrict__ partial,
                                                int64_t rows, int C4) {
  __shared__ float4 comb[4][64];
  colsum_block(x, out, partial, rows, C4, (int)gridDim.y, comb);
}

// up to 8 column sums in one launch (blockIdx.z = which): the bias gradients of one backward pass
struct ColsumArgs {
  const float* x[8];
  float* partial[8];
  int64_t rows[8];
  int c4[8], by[8];
};
__global__ void __launch_bounds__(256) k_colsum_batch(const ColsumArgs b) {
  __shared__ float4 comb[4][64];
  const int seg = blockIdx.z;
  if ((int)blockIdx.y >= b.by[seg] || (int)blockIdx.x * 64 >= b.c4[seg]) return;   // uniform per block
  colsum_block(b.x[seg], nullptr, b.partial[seg], b.rows[seg], b.c4[seg], b.by[seg], comb);
}

static int colsum_slices(int64_t nrows) {
  const int64_t want = cdiv(nrows, 4 * 16);
  return (int)(want < 1 ? 1 : (want > kColsumMaxSlices ? kColsumMaxSlices : want));
}

int ColsumBatch::add(const float* src, float* dst, int64_t nrows, int C) {
  if (nrows <= 0 || !dst) return DRIN_OK;
  if (C % 4 || n >= 8) {
    set_error("colsum: C=%d must be a multiple of 4 (and at most 8 sums per batch)", C);
    return DRIN_E_SHAPE;
  }
  x[n] = src, out[n] = dst, rows[n] = nrows, c4[n] = C / 4, by[n] = colsum_slices(nrows);
  ++n;
  return DRIN_OK;
}

int launch_colsum_batch(const ColsumBatch& b, hipStream_t st, float* scratch, size_t scratch_floats, SliceSum* defer) {
  if (b.n == 0) return DRIN_OK;
  SliceSum local;
  SliceSum& sums = defer != nullptr ? *defer : local;
  ColsumArgs a;
  int gx = 1, gy = 1;
  size_t used = 0;
  for (int i = 0; i < b.n; ++i) {
    gx = gx > (int)cdiv(b.c4[i], 64) ? gx : (int)cdiv(b.c4[i], 64);
    gy = gy > b.by[i] ? gy : b.by[i];
    const size_t need = (size_t)b.by[i] * b.c4[i] * 4;
    if (scratch == nullptr || !aligned16(scratch) || used + need > scratch_floats) {
      set_error("colsum batch: %zu floats of slice scratch, %zu needed", scratch_floats, used + need);
      return DRIN_E_WORKSPACE;
    }
    a.x[i] = b.x[i], a.partial[i] = scratch + used, a.rows[i] = b.rows[i], a.c4[i] = b.c4[i], a.by[i] = b.by[i];
    DRIN_TRY(sums.add(b.out[i], b.c4[i] * 4, 1, b.c4[i] * 4, scratch + used, b.by[i]));
    used += need;
  }
  {
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_colsum_batch, dim3((unsigned)gx, (unsigned)gy, (unsigned)b.n), dim3(256), 0, st, a);
    DRIN_CHECK_LAUNCH("k_colsum_batch");
  }
  return defer != nullptr ? DRIN_OK : launch_slice_sum(local, st);
}

int launch_colsum(const float* x, float* out, int64_t rows, int C, hipStream_t st, float* scratch, size_t scratch_floats) {
  if (rows <= 0 || !out) return DRIN_OK;
  if (C % 4) {
    set_error("colsum: C=%d must be a multiple of 4", C);
    return DRIN_E_SHAPE;
  }
  if (!aligned16(out) || !aligned16(x)) {
    set_error("colsum: operands must be 16-byte aligned");
    return DRIN_E_ALIGN;
  }
  int by = colsum_slices(rows);
  const bool sliced = by > 1 && scratch != nullptr && aligned16(scratch) && (size_t)by * C <= scratch_floats;
  if (!sliced) by = 1;
  {
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)cdiv(C / 4, 64), (unsigned)by), dim3(256), 0, st, x, out,
                       sliced ? scratch : (float*)nullptr, rows, C / 4);
    DRIN_CHECK_LAUNCH("k_colsum");
  }
  if (!sliced) return DRIN_OK;
  SliceSum sums;
  DRIN_TRY(sums.add(out, C, 1, C, scratch, by));
  return launch_slice_sum(sums, st);
}

// ------------------------------------------------------------------------------------------------
// dpre[k][p] = g[k][p] * e'[k][p] * (1 - e'[k][p])     (sigmoid of model.py:133)
__global__ void __launch_bounds__(256) k_sigmoid_bwd(const float* __restrict__ g, const float* __restrict__ e_new,
                                                     float* __restrict__ dpre, int64_t n, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    dpre[i] = g[i] * edge_act_grad(act, e_new[i]);
  }
}

int launch_sigmoid_bwd(const float* g, const float* e_new, float* dpre, int64_t n, hipStream_t st, int act) {
  if (n <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_sigmoid_bwd, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, g, e_new, dpre, n, act);
  DRIN_CHECK_LAUNCH("k_sigmoid_bwd");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// out[b, :] = scale * (sum_n w1[b,n] v1[b,n,:] + sum_n w2[b,n] v2[b,n,:]) + u[b, :]
// (v2 / u may be NULL).  The transposes of the entity<-mention and edge-update products.
// Block = 4 waves x 64 float4 columns: wave w sums candidates w, w + 4, ... (four loads in flight), the waves are
// combined through LDS in order - a mention-sized batch (B = 64) is only 3 B blocks, so the candidate loop must not
// be one serial chain per thread.
__global__ void __launch_bounds__(256) k_mention_reduce(const float* __restrict__ w1, const float* __restrict__ v1,
                                                        const float* __restrict__ w2, const float* __restrict__ v2,
                                                        const float* __restrict__ u, float* __restrict__ out, int N,
                                                        int D4, float scale) {
  __shared__ float4 comb[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c4 = blockIdx.x * 64 + lane;
  const int64_t b = blockIdx.y;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < D4) {
    const int64_t off = (b * N) * (int64_t)D4 * 4 + (int64_t)c4 * 4, row = (int64_t)D4 * 4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    int n = wave;
    for (; n + 4 < N; n += 8) {  // two candidates of this wave per trip
      const float4 a0 = ld4(v1 + off + n * row), a1 = ld4(v1 + off + (n + 4) * row);
      s = fma4(w1[b * N + n], a0, s);
      t = fma4(w1[b * N + n + 4], a1, t);
      if (v2 != nullptr) {
        const float4 c0 = ld4(v2 + off + n * row), c1 = ld4(v2 + off + (n + 4) * row);
        s = fma4(w2[b * N + n], c0, s);
        t = fma4(w2[b * N + n + 4], c1, t);
      }
    }
    for (; n < N; n += 4) {
      s = fma4(w1[b * N + n], ld4(v1 + off + n * row), s);
      if (v2 != nullptr) s = fma4(w2[b * N + n], ld4(v2 + off + n * row), s);
    }
    s = s + t;
  }
  comb[wave][lane] = s;
  __syncthreads();
  if (wave != 0 || c4 >= D4) return;
  s = ((comb[0][lane] + comb[1][lane]) + (comb[2][lane] + comb[3][lane])) * scale;
  if (u != nullptr) s = s + ld4(u + b * (int64_t)D4 * 4 + (int64_t)c4 * 4);
  st4(out + b * (int64_t)D4 * 4 + (int64_t)c4 * 4, s);
}

int launch_mention_reduce(const float* w1, const float* v1, const float* w2, const float* v2, const float* u,
                          float* out, int B, int N, int D, float scale, hipStream_t st) {
  if (B <= 0) return DRIN_OK;
  for (int b0 = 0; b0 < B; b0 += 65535) {
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    const int64_t po = (int64_t)b0 * N, vo = po * D;
    KernelTimer timer(DRIN_KC_GCN, st);
    hipLaunchKernelGGL(k_mention_reduce, dim3((unsigned)cdiv(D / 4, 64), (unsigned)nb), dim3(256), 0, st, w1 + po,
                       v1 + vo, w2 ? w2 + po : nullptr, v2 ? v2 + vo : nullptr, u ? u + (int64_t)b0 * D : nullptr,
                       out + (int64_t)b0 * D, N, D / 4, scale);
    DRIN_CHECK_LAUNCH("k_mention_reduce");
  }
  return DRIN_OK;
}

// Two reductions over the same rows in one pass (the text and the image mention of a layer):
//   outA[b] = scale (sum_n w[0][p] v1[p] + sum_n w[1][p] v2[p]) + uA[b]
//   outB[b] = scale (sum_n w[2][p] v1[p] + sum_n w[3][p] v2[p]) + uB[b]         w = [4][pairs]; v2 / uA / uB may be NULL
// Candidate split and summation order are those of k_mention_reduce: bit-identical to two launches of it, v1 / v2 read once.
struct MentionReduceJob {
  const float *w, *v1, *v2, *uA, *uB;
  float *outA, *outB;
  float scale;
};
__device__ __forceinline__ void mention_reduce2_body(const float* __restrict__ w, int64_t pairs,
                                                     const float* __restrict__ v1, const float* __restrict__ v2,
                                                     const float* __restrict__ uA, const float* __restrict__ uB,
                                                     float* __restrict__ outA, float* __restrict__ outB, int N,
                                                     int D4, float scale) {
  __shared__ float4 comb[2][4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c4 = blockIdx.x * 64 + lane;
  const int64_t b = blockIdx.y;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sA = zero, sB = zero;
  if (c4 < D4) {
    const int64_t off = (b * N) * (int64_t)D4 * 4 + (int64_t)c4 * 4, row = (int64_t)D4 * 4;
    const float* w0 = w + b * N;
    const float* w1 = w0 + pairs;
    const float* w2 = w0 + 2 * pairs;
    const float* w3 = w0 + 3 * pairs;
    float4 tA = zero, tB = zero;
    int n = wave;
    for (; n + 4 < N; n += 8) {  // two candidates of this wave per trip
      const float4 a0 = ld4(v1 + off + n * row), a1 = ld4(v1 + off + (n + 4) * row);
      float4 c0 = zero, c1 = zero;
      if (v2 != nullptr) {
        c0 = ld4(v2 + off + n * row);
        c1 = ld4(v2 + off + (n + 4) * row);
      }
      sA = fma4(w0[n], a0, sA);
      tA = fma4(w0[n + 4], a1, tA);
      sB = fma4(w2[n], a0, sB);
      tB = fma4(w2[n + 4], a1, tB);
      if (v2 != nullptr) {
        sA = fma4(w1[n], c0, sA);
        tA = fma4(w1[n + 4], c1, tA);
        sB = fma4(w3[n], c0, sB);
        tB = fma4(w3[n + 4], c1, tB);
      }
    }
    for (; n < N; n += 4) {
      const float4 a0 = ld4(v1 + off + n * row);
      sA = fma4(w0[n], a0, sA);
      sB = fma4(w2[n], a0, sB);
      if (v2 != nullptr) {
        const float4 c0 = ld4(v2 + off + n * row);
        sA = fma4(w1[n], c0, sA);
        sB = fma4(w3[n], c0, sB);
      }
    }
    sA = sA + tA;
    sB = sB + tB;
  }
  comb[0][wave][lane] = sA;
  comb[1][wave][lane] = sB;
  __syncthreads();
  if (wave != 0 || c4 >= D4) return;
  sA = ((comb[0][0][lane] + comb[0][1][lane]) + (comb[0][2][lane] + comb[0][3][lane])) * scale;
  sB = ((comb[1][0][lane] + comb[1][1][lane]) + (comb[1][2][lane] + comb[1][3][lane])) * scale;
  if (uA != nullptr) sA = sA + ld4(uA + b * (int64_t)D4 * 4 + (int64_t)c4 * 4);
  if (uB != nullptr) sB = sB + ld4(uB + b * (int64_t)D4 * 4 + (int64_t)c4 * 4);
  st4(outA + b * (int64_t)D4 * 4 + (int64_t)c4 * 4, sA);
  st4(outB + b * (int64_t)D4 * 4 + (int64_t)c4 * 4, sB);
}
__global__ void __launch_bounds__(256) k_mention_reduce2(const float* __restrict__ w, int64_t pairs,
                                                         const float* __restrict__ v1, const float* __restrict__ v2,
                                                         const float* __restrict__ uA, const float* __restrict__ uB,
                                                         float* __restrict__ outA, float* __restrict__ outB, int N,
                                                         int D4, float scale) {
  mention_reduce2_body(w, pairs, v1, v2, uA, uB, outA, outB, N, D4, scale);
}
// two such reductions that do not depend on each other (the edge update's dfu and the aggregation's mention side of one
// layer's backward) in ONE launch: blockIdx.z picks the job; the arithmetic - hence every bit - is the single kernel's
__global__ void __launch_bounds__(256) k_mention_reduce2_pair(const MentionReduceJob a, const MentionReduceJob b, int64_t pairs,
                                                              int N, int D4) {
  const MentionReduceJob& j = blockIdx.z == 0 ? a : b;
  mention_reduce2_body(j.w, pairs, j.v1, j.v2, j.uA, j.uB, j.outA, j.outB, N, D4, j.scale);
}

int launch_mention_reduce2_pair(const float* wa, const float* va1, const float* va2, const float* uaA, const float* uaB, float* outaA,
                                float* outaB, float scale_a, const float* wb, const float* vb1, const float* vb2, const float* ubA,
                                const float* ubB, float* outbA, float* outbB, float scale_b, int B, int N, int D, hipStream_t st) {
  if (B <= 0 || N <= 0) return DRIN_OK;
  if (B > 65535 || (D % 4)) {
    set_error("mention_reduce2_pair: B=%d D=%d outside the grid / 16-byte contract", B, D);
    return DRIN_E_SHAPE;
  }
  const MentionReduceJob ja{wa, va1, va2, uaA, uaB, outaA, outaB, scale_a}, jb{wb, vb1, vb2, ubA, ubB, outbA, outbB, scale_b};
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_mention_reduce2_pair, dim3((unsigned)cdiv(D / 4, 64), (unsigned)B, 2), dim3(256), 0, st, ja, jb,
                     (int64_t)B * N, N, D / 4);
  DRIN_CHECK_LAUNCH("k_mention_reduce2_pair");
  return DRIN_OK;
}

int launch_mention_reduce2(const float* w, const float* v1, const float* v2, const float* uA, const float* uB,
                           float* outA, float* outB, int B, int N, int D, float scale, hipStream_t st) {
  if (B <= 0) return DRIN_OK;
  if (B > 65535) {
    set_error("mention_reduce2: batch %d exceeds the grid limit of 65535 mentions per launch", B);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_mention_reduce2, dim3((unsigned)cdiv(D / 4, 64), (unsigned)B), dim3(256), 0, st, w,
                     (int64_t)B * N, v1, v2, uA, uB, outA, outB, N, D / 4, scale);
  DRIN_CHECK_LAUNCH("k_mention_reduce2");
  return DRIN_OK;
}

// out[p, :] = scale * (w1[p] m1[b, :] + w2[p] m2[b, :])        (m2 may be NULL); one wave per pair
__global__ void __launch_bounds__(256) k_entity_combine(const float* __restrict__ w1, const float* __restrict__ m1,
                                                        const float* __restrict__ w2, const float* __restrict__ m2,
                                                        float* __restrict__ out, int64_t pairs, int N, int D4,
                                                        float scale) {
  int64_t p, b;
  if (!wave_pair(pairs, N, p, b)) return;
  const int lane = threadIdx.x & 63;
  const float a1 = w1[p], a2 = m2 != nullptr ? w2[p] : 0.f;
  const float* r1 = m1 + b * (int64_t)D4 * 4;
  const float* r2 = m2 != nullptr ? m2 + b * (int64_t)D4 * 4 : nullptr;
  float* o = out + p * (int64_t)D4 * 4;
  for (int c4 = lane; c4 < D4; c4 += 64) {
    float4 s = ld4(r1 + c4 * 4) * a1;
    if (r2 != nullptr) s = fma4(a2, ld4(r2 + c4 * 4), s);
    st4(o + c4 * 4, s * scale);
  }
}

// The scalar-edge update's backward in one pass over the pairs (model.py:148-153,133), one wave per pair:
//   dpre_k[p] = g_k[p] act'(e'_k[p])      [sigmoid: e' (1 - e')]       k = tt, ti, it, ii      (written: the mention side reads it)
//   dfv_t[p]  = (dpre_tt fu_t[b] + dpre_it fu_i[b]) / D            dfv_i[p] = (dpre_ti fu_t[b] + dpre_ii fu_i[b]) / D
// g, e_new, dpre = [4][pairs]; fu = [2][B][D]; dfv = [2][pairs][D].
__global__ void __launch_bounds__(256) k_edge_update_bwd(const float* __restrict__ g, const float* __restrict__ e_new,
                                                         const float* __restrict__ fu, float* __restrict__ dpre,
                                                         float* __restrict__ dfv, int64_t pairs, int B, int N, int D4,
                                                         float scale, int act) {
  int64_t p, b;
  if (!wave_pair(pairs, N, p, b)) return;
  const int lane = threadIdx.x & 63;
  float d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float e = e_new[k * pairs + p];
    d[k] = g[k * pairs + p] * edge_act_grad(act, e);
  }
  if (lane < 4) dpre[(int64_t)lane * pairs + p] = lane == 0 ? d[0] : lane == 1 ? d[1] : lane == 2 ? d[2] : d[3];
  const float* ft = fu + b * (int64_t)D4 * 4;
  const float* fi = fu + ((int64_t)B + b) * (int64_t)D4 * 4;
  float* ot = dfv + p * (int64_t)D4 * 4;
  float* oi = dfv + (pairs + p) * (int64_t)D4 * 4;
  for (int c4 = lane; c4 < D4; c4 += 64) {
    const float4 t = ld4(ft + c4 * 4), i = ld4(fi + c4 * 4);
    // (same association as k_entity_combine: (w1 m1) then fma(w2, m2, .), scaled last)
    st4(ot + c4 * 4, fma4(d[2], i, t * d[0]) * scale);
    st4(oi + c4 * 4, fma4(d[3], i, t * d[1]) * scale);
  }
}

int launch_edge_update_bwd(const float* g, const float* e_new, const float* fu, float* dpre, float* dfv, int B, int N, int D,
                           float scale, hipStream_t st, int act) {
  const int64_t pairs = (int64_t)B * N;
  if (pairs <= 0 || D <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_edge_update_bwd, pair_grid(B, N), dim3(256), 0, st, g, e_new, fu, dpre, dfv, pairs, B, N, D / 4, scale, act);
  DRIN_CHECK_LAUNCH("k_edge_update_bwd");
  return DRIN_OK;
}

int launch_entity_combine(const float* w1, const float* m1, const float* w2, const float* m2, float* out, int B, int N,
                          int D, float scale, hipStream_t st) {
  const int64_t pairs = (int64_t)B * N;
  if (pairs <= 0 || D <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_entity_combine, pair_grid(B, N), dim3(256), 0, st, w1, m1, w2, m2, out, pairs, N, D / 4, scale);
  DRIN_CHECK_LAUNCH("k_entity_combine");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// Entity side of the aggregation backward (transposes of model.py:143-146), one wave per pair.
// With A_mt/A_mi/A_et/A_ei the W_h inputs of the four vertex types and e = (tt, ti, it, ii):
//   d_et[p] = dA_et[p] + (e_tt dA_mt[b] + e_it dA_mi[b]) / N
//   d_ei[p] = dA_ei[p] + (e_ti dA_mt[b] + e_ii dA_mi[b]) / N
//   de_tt = dA_mt.et / N + dA_et.mt     de_ti = dA_mt.ei / N + dA_ei.mt
//   de_it = dA_mi.et / N + dA_et.mi     de_ii = dA_mi.ei / N + dA_ei.mi
// de_k additionally receives `de_extra[k]` (the edge-update / static pass-through gradient) and is
// multiplied by the edge switch m_k (model.py:122).  dA_mi / dA_ei are NULL in the last layer, whose
// image vertices are dead.
__global__ void __launch_bounds__(256)
    k_entity_side_bwd(const float* __restrict__ dA_mt, const float* __restrict__ dA_mi, const float* __restrict__ dA_et,
                      const float* __restrict__ dA_ei, const float* __restrict__ mt, const float* __restrict__ mi,
                      const float* __restrict__ et, const float* __restrict__ ei, const float* __restrict__ e,
                      const float* __restrict__ de_extra, float* __restrict__ d_et, float* __restrict__ d_ei,
                      float* __restrict__ de, int64_t pairs, int N, int D4, float m0, float m1, float m2, float m3,
                      bool accumulate) {
  int64_t p, b;
  if (!wave_pair(pairs, N, p, b)) return;
  const int lane = threadIdx.x & 63;
  const float inv_n = 1.0f / (float)N;
  const float e_tt = e[p], e_ti = e[pairs + p], e_it = e[2 * pairs + p], e_ii = e[3 * pairs + p];
  float s_tt = 0.f, s_ti = 0.f, s_it = 0.f, s_ii = 0.f;
  const int64_t mo = b * (int64_t)D4 * 4, po = p * (int64_t)D4 * 4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c4 = lane; c4 < D4; c4 += 64) {
    const int o = c4 * 4;
    const float4 gmt = ld4(dA_mt + mo + o);
    const float4 gmi = dA_mi ? ld4(dA_mi + mo + o) : zero;
    const float4 get = ld4(dA_et + po + o);
    const float4 gei = dA_ei ? ld4(dA_ei + po + o) : zero;
    const float4 vmt = ld4(mt + mo + o), vmi = ld4(mi + mo + o);
    const float4 vet = ld4(et + po + o), vei = ld4(ei + po + o);
    s_tt += dot4(gmt, vet) * inv_n + dot4(get, vmt);
    s_ti += dot4(gmt, vei) * inv_n + dot4(gei, vmt);
    s_it += dot4(gmi, vet) * inv_n + dot4(get, vmi);
    s_ii += dot4(gmi, vei) * inv_n + dot4(gei, vmi);
    float4 o_et = fma4(e_tt * inv_n, gmt, fma4(e_it * inv_n, gmi, get));
    float4 o_ei = fma4(e_ti * inv_n, gmt, fma4(e_ii * inv_n, gmi, gei));
    if (accumulate) {  // the edge-update term dfv W_v is already there (adding it here is ~4x cheaper than a
      o_et = o_et + ld4(d_et + po + o);  // read-modify-write GEMM epilogue: 681 -> 424 us at 2 x 51 712 rows)
      o_ei = o_ei + ld4(d_ei + po + o);
    }
    st4(d_et + po + o, o_et);
    st4(d_ei + po + o, o_ei);
  }
  s_tt = wave_sum(s_tt);
  s_ti = wave_sum(s_ti);
  s_it = wave_sum(s_it);
  s_ii = wave_sum(s_ii);
  if (lane == 0) {
    const float x0 = de_extra ? de_extra[p] : 0.f, x1 = de_extra ? de_extra[pairs + p] : 0.f;
    const float x2 = de_extra ? de_extra[2 * pairs + p] : 0.f, x3 = de_extra ? de_extra[3 * pairs + p] : 0.f;
    de[p] = (s_tt + x0) * m0;
    de[pairs + p] = (s_ti + x1) * m1;
    de[2 * pairs + p] = (s_it + x2) * m2;
    de[3 * pairs + p] = (s_ii + x3) * m3;
  }
}

int launch_entity_side_bwd(const float* dA_mt, const float* dA_mi, const float* dA_et, const float* dA_ei,
                           const float* mt, const float* mi, const float* et, const float* ei, const float* e,
                           const float* de_extra, float* d_et, float* d_ei, float* de, int B, int N, int D,
                           const float* mask, bool accumulate, hipStream_t st) {
  const int64_t pairs = (int64_t)B * N;
  if (pairs <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_entity_side_bwd, pair_grid(B, N), dim3(256), 0, st, dA_mt, dA_mi, dA_et, dA_ei,
                     mt, mi, et, ei, e, de_extra, d_et, d_ei, de, pairs, N, D / 4, mask[0], mask[1], mask[2], mask[3], accumulate);
  DRIN_CHECK_LAUNCH("k_entity_side_bwd");
  return DRIN_OK;
}

}  // namespace drin
