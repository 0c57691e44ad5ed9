// HBM-bound, parameter-free kernels of the path: input pooling (VertexEncoder, drin/model.py:41-45,
// baselines/ghmfc.py:54-60,245-249) and the static edges (EdgeEncoder, drin/model.py:60-94).
// All are pure streaming: 16-byte loads per lane, one pass over the bytes they need, wave64 shuffle
// reductions, no LDS.
#include "device_utils.h"
#include "internal.h"

namespace drin {

// ------------------------------------------------------------------------------------------------
// out[g, c] = mean_s in[g, s, c]      (region mean over P, "inner" means of model.py:43-44,78-83)
// grid: (ceil(cols/4 / 256), groups); each thread owns one float4 column and walks the inner axis.
template <typename T>
__global__ void __launch_bounds__(256) k_axis_mean(const T* __restrict__ in, float* __restrict__ out, int inner,
                                                   int cols4) {
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (c4 >= cols4) return;
  const int64_t g = blockIdx.y;
  const T* p = in + (g * inner) * (int64_t)cols4 * 4 + (int64_t)c4 * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 7 <= inner; s += 7) {  // 7 independent 16-B loads in flight (P = 49 = 7 * 7)
    float4 v[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = ld4_stream(p + (int64_t)(s + j) * cols4 * 4);  // read once
#pragma unroll
    for (int j = 0; j < 7; ++j) acc = acc + v[j];
  }
  for (; s < inner; ++s) acc = acc + ld4_stream(p + (int64_t)s * cols4 * 4);
  const float cnt = (float)inner;  // sum / count like torch.mean (a true division, not a reciprocal multiply)
  st4(out + g * (int64_t)cols4 * 4 + (int64_t)c4 * 4, make_float4(acc.x / cnt, acc.y / cnt, acc.z / cnt, acc.w / cnt));
}

template <typename T>
static int launch_axis_mean_t(const T* in, float* out, int64_t groups, int inner, int cols, hipStream_t st) {
  if (groups <= 0) return DRIN_OK;
  if (cols % 4 != 0 || inner <= 0) {
    set_error("axis_mean: cols=%d must be a multiple of 4 and inner=%d positive", cols, inner);
    return DRIN_E_SHAPE;
  }
  if (groups > 65535) {  // grid.y limit: fold in chunks
    for (int64_t g0 = 0; g0 < groups; g0 += 65535) {
      const int64_t n = groups - g0 < 65535 ? groups - g0 : 65535;
      DRIN_TRY(launch_axis_mean_t<T>(in + g0 * inner * cols, out + g0 * cols, n, inner, cols, st));
    }
    return DRIN_OK;
  }
  const int cols4 = cols / 4;
  dim3 grid((unsigned)cdiv(cols4, 256), (unsigned)groups);
  KernelTimer timer(DRIN_KC_POOL, st);
  hipLaunchKernelGGL(k_axis_mean<T>, grid, dim3(256), 0, st, in, out, inner, cols4);
  DRIN_CHECK_LAUNCH("k_axis_mean");
  return DRIN_OK;
}

int launch_axis_mean(const float* in, float* out, int64_t groups, int inner, int cols, hipStream_t st) {
  return launch_axis_mean_t<float>(in, out, groups, inner, cols, st);
}
int launch_axis_mean_bf16(const void* in, float* out, int64_t groups, int inner, int cols, hipStream_t st) {
  return launch_axis_mean_t<__bf16>(static_cast<const __bf16*>(in), out, groups, inner, cols, st);
}

// ------------------------------------------------------------------------------------------------
// Avg.avg (baselines/ghmfc.py:54-60).  Python slice semantics: the range is clipped to [0, L]; an
// empty range gives 0/0 = NaN like torch.mean of an empty slice.  grid: (ceil(D/4/256), B).
template <typename T>
__global__ void __launch_bounds__(256) k_span_mean(const T* __restrict__ seq, const int64_t* __restrict__ start,
                                                   const int64_t* __restrict__ end, float* __restrict__ out, int L,
                                                   int D4) {
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (c4 >= D4) return;
  const int b = blockIdx.y;
  int64_t s = start[b], e = end[b];
  if (s < 0) s = s + L < 0 ? 0 : s + L;  // python negative index
  if (e < 0) e = e + L < 0 ? 0 : e + L;
  if (e > L) e = L;
  if (s > L) s = L;
  const T* p = seq + ((int64_t)b * L) * D4 * 4 + (int64_t)c4 * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t t = s; t < e; ++t) acc = acc + ld4(p + t * D4 * 4);
  const float cnt = e > s ? (float)(e - s) : 0.0f;  // empty span: 0 / 0 = NaN, as the reference
  st4(out + (int64_t)b * D4 * 4 + (int64_t)c4 * 4, make_float4(acc.x / cnt, acc.y / cnt, acc.z / cnt, acc.w / cnt));
}

template <typename T>
static int launch_span_mean_t(const T* seq, const int64_t* start, const int64_t* end, float* out, int B, int L, int D,
                              hipStream_t st) {
  if (B <= 0) return DRIN_OK;
  if (D % 4 != 0) {
    set_error("span_mean: D=%d must be a multiple of 4", D);
    return DRIN_E_SHAPE;
  }
  for (int b0 = 0; b0 < B; b0 += 65535) {
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    dim3 grid((unsigned)cdiv(D / 4, 256), (unsigned)nb);
    KernelTimer timer(DRIN_KC_POOL, st);
    hipLaunchKernelGGL(k_span_mean<T>, grid, dim3(256), 0, st, seq + (int64_t)b0 * L * D, start + b0, end + b0,
                       out + (int64_t)b0 * D, L, D / 4);
    DRIN_CHECK_LAUNCH("k_span_mean");
  }
  return DRIN_OK;
}

int launch_span_mean(const float* seq, const int64_t* start, const int64_t* end, float* out, int B, int L, int D,
                     hipStream_t st) {
  return launch_span_mean_t<float>(seq, start, end, out, B, L, D, st);
}
int launch_span_mean_bf16(const void* seq, const int64_t* start, const int64_t* end, float* out, int B, int L, int D,
                          hipStream_t st) {
  return launch_span_mean_t<__bf16>(static_cast<const __bf16*>(seq), start, end, out, B, L, D, st);
}

// ------------------------------------------------------------------------------------------------
// WikiMEL entity pooling (baselines/ghmfc.py:245-249): x[p] = mean(feat[p, 1:ntok-1]), ntok = sum(mask[p]).
// Only the rows inside the slice are read (the compulsory bytes).  One block per pair;
// each thread owns float4 columns and keeps 8 token rows in flight.
// T_ = float or __bf16 (features stored as bf16: values widen exactly, the sums are the fp32 sums of the widened rows).
template <typename T_>
__global__ void __launch_bounds__(256) k_entity_token_mean(const T_* __restrict__ feat,
                                                           const int64_t* __restrict__ mask, float* __restrict__ out,
                                                           int T, int D4) {
  const int64_t p = blockIdx.x;
  // every wave recomputes ntok (T <= 512 int64 values, L2/L1 resident) - cheaper than a barrier
  int cnt = 0;
  for (int t = threadIdx.x & 63; t < T; t += 64) cnt += (int)mask[p * T + t];
  cnt = (int)wave_sum((float)cnt);  // T <= 2^24: exact in fp32
  int stop = cnt - 1;
  if (stop < 0) stop += T;  // python negative stop index (ntok == 0 -> 1:-1)
  if (stop < 0) stop = 0;
  if (stop > T) stop = T;
  const int n = stop - 1;  // rows 1 .. stop-1
  const T_* base = feat + p * (int64_t)T * D4 * 4;
  for (int c4 = threadIdx.x; c4 < D4; c4 += blockDim.x) {
    const T_* col = base + (int64_t)c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int t = 1;
    for (; t + 8 <= stop; t += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ld4_stream(col + (int64_t)(t + j) * D4 * 4);  // read once
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = acc + v[j];
    }
    for (; t < stop; ++t) acc = acc + ld4_stream(col + (int64_t)t * D4 * 4);
    const float den = n > 0 ? (float)n : 0.0f;  // empty slice: 0 / 0 = NaN, as the reference
    st4(out + p * (int64_t)D4 * 4 + (int64_t)c4 * 4, make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den));
  }
}

template <typename T_>
static int launch_entity_token_mean_t(const T_* feat, const int64_t* mask, float* out, int64_t pairs, int T, int D,
                                      hipStream_t st) {
  if (pairs <= 0) return DRIN_OK;
  if (D % 4 != 0 || T <= 0) {
    set_error("entity_token_mean: D=%d must be a multiple of 4, T=%d positive", D, T);
    return DRIN_E_SHAPE;
  }
  const int D4 = D / 4;
  int threads = D4 >= 192 ? 192 : (D4 > 64 ? 128 : 64);  // D = 768 -> one float4 column per thread, 3 waves
  KernelTimer timer(DRIN_KC_POOL, st);
  hipLaunchKernelGGL(k_entity_token_mean<T_>, dim3((unsigned)pairs), dim3(threads), 0, st, feat, mask, out, T, D4);
  DRIN_CHECK_LAUNCH("k_entity_token_mean");
  return DRIN_OK;
}

int launch_entity_token_mean(const float* feat, const int64_t* mask, float* out, int64_t pairs, int T, int D,
                             hipStream_t st) {
  return launch_entity_token_mean_t<float>(feat, mask, out, pairs, T, D, st);
}
int launch_entity_token_mean_bf16(const void* feat, const int64_t* mask, float* out, int64_t pairs, int T, int D,
                                  hipStream_t st) {
  return launch_entity_token_mean_t<__bf16>(static_cast<const __bf16*>(feat), mask, out, pairs, T, D, st);
}

// ------------------------------------------------------------------------------------------------
// out[b*N+n] = scale * cos(x[b], y[pair])  -- one wave per pair.  Used for the tt edge (model.py:71-76,
// y = entity pooler vector or token 0 of the token block) and for the final score (model.py:207-209).
__global__ void __launch_bounds__(256) k_cosine_rows(const float* __restrict__ x, const float* __restrict__ y,
                                                     int64_t y_stride, float* __restrict__ out, int64_t pairs, int N,
                                                     int D4, float eps, float scale, const int64_t* __restrict__ y_index,
                                                     const float* __restrict__ sim1, const float* __restrict__ sim2,
                                                     float* __restrict__ out1, float* __restrict__ out2, float div) {
  int64_t p, b;
  if (!wave_pair(pairs, N, p, b)) return;
  const int lane = threadIdx.x & 63;
  // the two CLIP edges of model.py:203 ride along (ti = mtei / 100, it = miet / 100: true divisions)
  if (sim1 != nullptr && lane == 1) out1[p] = sim1[p] / div;
  if (sim2 != nullptr && lane == 2) out2[p] = sim2[p] / div;
  const float* xr = x + b * (int64_t)D4 * 4;
  const float* yr = y + (y_index != nullptr ? y_index[p] : p) * y_stride;  // y_index: y is a table, pair p reads row y_index[p]
  float xy = 0.f, xx = 0.f, yy = 0.f;
  for (int c4 = lane; c4 < D4; c4 += 64) {
    const float4 a = ld4(xr + c4 * 4), b = ld4(yr + c4 * 4);
    xy += dot4(a, b);
    xx += dot4(a, a);
    yy += dot4(b, b);
  }
  xy = wave_sum(xy);
  xx = wave_sum(xx);
  yy = wave_sum(yy);
  if (lane == 0) out[p] = scale * cosine_from_sums(xy, xx, yy, eps);
}

int launch_cosine_rows(const float* x, const float* y, int64_t y_stride, float* out, int B, int N, int D, float eps,
                       float scale, hipStream_t st, const int64_t* y_index, const float* sim1, const float* sim2,
                       float* out1, float* out2, float div) {
  const int64_t pairs = (int64_t)B * N;
  if (pairs <= 0) return DRIN_OK;
  if (D % 4 != 0 || y_stride % 4 != 0) {
    set_error("cosine_rows: D=%d and stride=%lld must be multiples of 4", D, (long long)y_stride);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_EDGE, st);
  hipLaunchKernelGGL(k_cosine_rows, pair_grid(B, N), dim3(256), 0, st, x, y, y_stride, out, pairs, N,
                     D / 4, eps, scale, y_index, sim1, sim2, out1, out2, div);
  DRIN_CHECK_LAUNCH("k_cosine_rows");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// miei (model.py:84-92): sum_{i<Km, j<Ke} cos(mobj[b,i], eobj[b,n,j]) ms[b,i] es[b,n,j] / (sum ms es + 1e-9)
// Fast form (Ke == 1, R <= 2048, Km R floats fit LDS; both datasets): one workgroup per (mention, 32-candidate chunk).
// The Km mention object rows and their squared norms sit in LDS, each entity object row is read ONCE (streaming) into
// registers and met by all Km rows.  (One wave per pair with the mention rows re-read through L1 / L2 moved 32 KB
// through the cache hierarchy per pair for 8 KB of HBM and ran at 2.1 TB/s.)  Generic form: one wave per pair, rows
// walked in memory per (i, j).  Per-lane summation order and the i-then-j order of the sums are the same in both.
constexpr int kMieiChunk = 32;

__global__ void __launch_bounds__(256) k_miei_lds(const float* __restrict__ mobj, const float* __restrict__ mscore,
                                                  const float* __restrict__ eobj, const float* __restrict__ escore,
                                                  float* __restrict__ out, int N, int Km, int R4, float cos_eps,
                                                  float miei_eps, float scale, const int64_t* __restrict__ e_index) {
  extern __shared__ float4 lds4[];  // [Km][R4] rows, then Km squared norms
  float* lds_xx = reinterpret_cast<float*>(lds4 + (size_t)Km * R4);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = blockIdx.y;
  const float* mrow = mobj + b * Km * (int64_t)R4 * 4;
  for (int i = threadIdx.x; i < Km * R4; i += 256) lds4[i] = ld4(mrow + (int64_t)i * 4);
  __syncthreads();
  for (int i = wave; i < Km; i += 4) {
    float xx = 0.f;
    for (int c4 = lane; c4 < R4; c4 += 64) {
      const float4 a = lds4[i * R4 + c4];
      xx += dot4(a, a);
    }
    xx = wave_sum(xx);
    if (lane == 0) lds_xx[i] = xx;
  }
  __syncthreads();
  const int n_end = min(N, ((int)blockIdx.x + 1) * kMieiChunk);
  for (int n = blockIdx.x * kMieiChunk + wave; n < n_end; n += 4) {
    const int64_t p = b * N + n;
    const int64_t er = e_index != nullptr ? e_index[p] : p;  // e_index: eobj / escore are tables
    const float* yr = eobj + er * (int64_t)R4 * 4;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c4 = lane + 64 * k;
      v[k] = c4 < R4 ? ld4_stream(yr + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);  // read once
    }
    float yy = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) yy += dot4(v[k], v[k]);
    yy = wave_sum(yy);
    const float es = escore[er];
    float sim = 0.f, wsum = 0.f;
    for (int i = 0; i < Km; ++i) {
      float xy = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c4 = lane + 64 * k;
        if (c4 < R4) xy += dot4(lds4[i * R4 + c4], v[k]);
      }
      xy = wave_sum(xy);
      const float w = mscore[b * Km + i] * es;
      sim += cosine_from_sums(xy, lds_xx[i], yy, cos_eps) * w;
      wsum += w;
    }
    if (lane == 0) out[p] = scale * (sim / (wsum + miei_eps));
  }
}

__global__ void __launch_bounds__(256) k_miei(const float* __restrict__ mobj, const float* __restrict__ mscore,
                                              const float* __restrict__ eobj, const float* __restrict__ escore,
                                              float* __restrict__ out, int64_t pairs, int N, int Km, int Ke, int R4,
                                              float cos_eps, float miei_eps, float scale) {
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= pairs) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = p / N;
  float sim = 0.f, wsum = 0.f;
  for (int i = 0; i < Km; ++i) {
    const float* xr = mobj + (b * Km + i) * (int64_t)R4 * 4;
    const float ms = mscore[b * Km + i];
    for (int j = 0; j < Ke; ++j) {
      const float* yr = eobj + (p * Ke + j) * (int64_t)R4 * 4;
      float xy = 0.f, xx = 0.f, yy = 0.f;
      for (int c4 = lane; c4 < R4; c4 += 64) {
        const float4 a = ld4(xr + c4 * 4), v = ld4(yr + c4 * 4);
        xy += dot4(a, v);
        xx += dot4(a, a);
        yy += dot4(v, v);
      }
      xy = wave_sum(xy);
      xx = wave_sum(xx);
      yy = wave_sum(yy);
      const float w = ms * escore[p * Ke + j];
      sim += cosine_from_sums(xy, xx, yy, cos_eps) * w;
      wsum += w;
    }
  }
  if (lane == 0) out[p] = scale * (sim / (wsum + miei_eps));
}

int launch_miei(const float* mobj, const float* mscore, const float* eobj, const float* escore, float* out, int B,
                int N, int Km, int Ke, int R, float cos_eps, float miei_eps, float scale, hipStream_t st,
                const int64_t* e_index) {
  const int64_t pairs = (int64_t)B * N;
  if (pairs <= 0) return DRIN_OK;
  if (R % 4 != 0) {
    set_error("miei: R=%d must be a multiple of 4", R);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_EDGE, st);
  const size_t lds = ((size_t)Km * R + Km) * sizeof(float);
  if (Ke == 1 && R <= 2048 && lds <= 64 * 1024 && B <= 65535) {
    hipLaunchKernelGGL(k_miei_lds, dim3((unsigned)cdiv(N, kMieiChunk), (unsigned)B), dim3(256), lds, st, mobj, mscore,
                       eobj, escore, out, N, Km, R / 4, cos_eps, miei_eps, scale, e_index);
  } else if (e_index != nullptr) {
    set_error("miei: table form (entity_index) is built for Ke == 1, R <= 2048 and at most 65535 mentions per call");
    return DRIN_E_UNSUPPORTED;
  } else {
    hipLaunchKernelGGL(k_miei, dim3((unsigned)cdiv(pairs, 4)), dim3(256), 0, st, mobj, mscore, eobj, escore, out, pairs,
                       N, Km, Ke, R / 4, cos_eps, miei_eps, scale);
  }
  DRIN_CHECK_LAUNCH("k_miei");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// out[i] = in[i] * mul / div   (model.py:203 divides the CLIP logits by 100: a true division, since
// x / 100 and x * 0.01f round differently; model.py:122 multiplies by the 0/1 edge switch)
__global__ void __launch_bounds__(256) k_scale_div(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                   float mul, float div) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (in[i] * mul) / div;
}

int launch_scale_div(const float* in, float* out, int64_t n, float mul, float div, hipStream_t st) {
  if (n <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_EDGE, st);
  hipLaunchKernelGGL(k_scale_div, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, in, out, n, mul, div);
  DRIN_CHECK_LAUNCH("k_scale_div");
  return DRIN_OK;
}

}  // namespace drin
