// Caller-side loss and metric of one batch on the device:
//   TripletLoss   common/utils.py:26-43   - every mention's positive distance against the WHOLE batch's
//                                           [B, N-1] distance matrix (utils.py:41-42), mean, / B
//   TopkAccuracy  common/utils.py:60-66   - sum(y * (y_hat >= k-th largest of the row)), ties count
// plus d loss / d scores, in three small launches with no host round trip: the reference reads the loss
// back every step (`train.py:35`) and walks the batch in a Python loop (utils.py:40-42).
//
//   k_triplet_rows   one wave per mention i: pos_i = sum_n -y_hat[i,n] y[i,n]; for every k the number of
//                    gold candidates inside the row's top k (a gold is inside iff fewer than k candidates
//                    score strictly higher - the tie rule of `y_pred >= lower`; NaN orders as torch.topk
//                    does: larger than everything, and a NaN gold is never counted)
//   k_triplet_pairs  one block per mention r: (1) for its N-1 candidates the hinge sum and the number of
//                    active hinges over all i  (2) for i = r the number of active hinges over the whole
//                    matrix; both give d loss / d y_hat[r, :] without atomics
//   k_triplet_finish fixed-order reduction of the per-mention partial sums and of the top-k hits
// The hinge is active where pos_i - p + margin >= 0 for the gradient (torch.clamp's backward mask) and
// NaN propagates into the loss exactly as torch.clamp(min=0) lets it.
#include "device_utils.h"
#include "internal.h"

namespace drin {

constexpr int kMaxTopk = 8;

struct TopkList {
  int k[kMaxTopk];
  int n;
};

__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void __launch_bounds__(64) k_triplet_rows(const float* __restrict__ scores, const uint8_t* __restrict__ answer,
                                                     float* __restrict__ pos, int* __restrict__ hits, int N, TopkList tk) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int C = N - 1;  // candidates without the answer slot (utils.py:36-37)
  const float* s = scores + (int64_t)i * N;
  const uint8_t* y = answer + (int64_t)i * C;
  float p = 0.f;
  for (int n = lane; n < C; n += 64) p += -s[n] * (float)y[n];
  p = wave_sum(p);
  int h[kMaxTopk];
#pragma unroll
  for (int q = 0; q < kMaxTopk; ++q) h[q] = 0;
  for (int g = 0; g < C; ++g) {  // uniform loop; gold entries are rare (one per row)
    const int w = y[g];
    if (w == 0) continue;
    const float sg = s[g];
    if (sg != sg) continue;  // NaN >= lower is false
    int greater = 0;
    for (int n = lane; n < C; n += 64) {
      const float v = s[n];
      greater += (v > sg || v != v) ? 1 : 0;
    }
    greater = wave_sum_int(greater);
#pragma unroll
    for (int q = 0; q < kMaxTopk; ++q)
      if (q < tk.n && greater < tk.k[q]) h[q] += w;
  }
  if (lane == 0) {
    pos[i] = p;
#pragma unroll
    for (int q = 0; q < kMaxTopk; ++q)
      if (q < tk.n) hits[(int64_t)i * kMaxTopk + q] = h[q];
  }
}

constexpr int kPosChunk = 4096;

__global__ void __launch_bounds__(256) k_triplet_pairs(const float* __restrict__ scores, const uint8_t* __restrict__ answer,
                                                       const float* __restrict__ pos, float* __restrict__ partial,
                                                       float* __restrict__ dscores, int B, int N, float margin,
                                                       float scale) {
  __shared__ float l_pos[kPosChunk];
  __shared__ float l_sum[4];
  __shared__ int l_cnt[4];
  const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int C = N - 1;
  const float* s = scores + (int64_t)r * N;

  // (1) hinge sums / active counts of this mention's candidates against every mention's positive distance
  float sum = 0.f;
  for (int n0 = 0; n0 < C; n0 += 256) {  // uniform trip count: the LDS refills below are block-wide
    const int n = n0 + t;
    const bool live = n < C;
    const float p = live ? -s[n] : 0.f;
    int cnt = 0;
    float acc = 0.f;
    for (int c0 = 0; c0 < B; c0 += kPosChunk) {
      const int len = min(kPosChunk, B - c0);
      __syncthreads();
      for (int j = t; j < len; j += 256) l_pos[j] = pos[c0 + j];
      __syncthreads();
      if (live)
        for (int j = 0; j < len; ++j) {
          const float v = l_pos[j] - p + margin;
          acc += (v <= 0.f) ? 0.f : v;  // NaN falls through, as in torch.clamp(min=0)
          cnt += (v >= 0.f) ? 1 : 0;
        }
    }
    if (live) {
      sum += acc;
      if (dscores != nullptr) dscores[(int64_t)r * N + n] = scale * (float)cnt;
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) l_sum[wave] = sum;

  // (2) active hinges of mention r's positive distance over the whole [B, N-1] matrix
  if (dscores != nullptr) {
    const float pr = pos[r] + margin;
    int cnt = 0;
    // rows split over the 4 waves, candidates over the lanes: no division in the B * N loop
    // (four rows per trip: the loads of a trip are independent, so their L2 latencies overlap - 89 -> ~30 us at B = 512)
    int bb = wave;
    for (; bb + 12 < B; bb += 16) {
      const float* s0 = scores + (int64_t)bb * N;
      const float* s1 = s0 + 4 * (int64_t)N;
      const float* s2 = s0 + 8 * (int64_t)N;
      const float* s3 = s0 + 12 * (int64_t)N;
      for (int n = lane; n < C; n += 64) {
        const float v0 = s0[n], v1 = s1[n], v2 = s2[n], v3 = s3[n];
        cnt += ((pr + v0 >= 0.f) ? 1 : 0) + ((pr + v1 >= 0.f) ? 1 : 0) + ((pr + v2 >= 0.f) ? 1 : 0) + ((pr + v3 >= 0.f) ? 1 : 0);
      }
    }
    for (; bb < B; bb += 4) {
      const float* sr = scores + (int64_t)bb * N;
      for (int n = lane; n < C; n += 64) cnt += (pr + sr[n] >= 0.f) ? 1 : 0;  // pos_r - p[b,n] + margin with p = -y_hat
    }
    cnt = wave_sum_int(cnt);
    if (lane == 0) l_cnt[wave] = cnt;
  }
  __syncthreads();
  if (t == 0) partial[r] = (l_sum[0] + l_sum[1]) + (l_sum[2] + l_sum[3]);
  if (dscores != nullptr) {
    const float w = scale * (float)(l_cnt[0] + l_cnt[1] + l_cnt[2] + l_cnt[3]);
    const uint8_t* y = answer + (int64_t)r * C;
    for (int n = t; n < C; n += 256) {
      const int yy = y[n];
      if (yy != 0) dscores[(int64_t)r * N + n] -= w * (float)yy;  // d pos_r / d y_hat[r,n] = -y[r,n]; same thread wrote it
    }
    if (t == 0) dscores[(int64_t)r * N + C] = 0.f;  // the answer slot never reaches the loss
  }
}

__global__ void __launch_bounds__(256) k_triplet_finish(const float* __restrict__ partial, const int* __restrict__ hits,
                                                        float* __restrict__ loss, int64_t* __restrict__ correct, int B,
                                                        float scale, int ntop) {
  __shared__ double l_sum[256];
  __shared__ int64_t l_hit[256];
  const int t = threadIdx.x;
  double acc = 0.0;
  for (int i = t; i < B; i += 256) acc += (double)partial[i];
  l_sum[t] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) l_sum[t] += l_sum[t + o];
    __syncthreads();
  }
  if (t == 0) loss[0] = (float)(l_sum[0] * (double)scale);
  if (correct == nullptr) return;
  for (int q = 0; q < ntop; ++q) {
    int64_t h = 0;
    for (int i = t; i < B; i += 256) h += hits[(int64_t)i * kMaxTopk + q];
    l_hit[t] = h;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) l_hit[t] += l_hit[t + o];
      __syncthreads();
    }
    if (t == 0) correct[q] += l_hit[0];
    __syncthreads();
  }
}

}  // namespace drin

using namespace drin;

extern "C" {

DRIN_API size_t drin_loss_workspace_bytes(int32_t batch) {
  if (batch <= 0) return 0;
  // pos [B] f32 | partial [B] f32 | hits [B, 8] i32
  return (size_t)batch * (4 + 4 + 4 * kMaxTopk);
}

DRIN_API int drin_triplet_topk(const float* scores, const uint8_t* answer, int32_t batch, int32_t num_candidates, float margin,
                      const int32_t* topk, int32_t num_topk, float* loss, float* d_scores, int64_t* correct,
                      void* workspace, size_t workspace_bytes, void* stream) {
  DRIN_BIND_DEVICE(stream, scores, "drin_triplet_topk");
  RoctxRange range("drin_triplet_topk");
  if (batch <= 0 || num_candidates < 2) {
    set_error("drin_triplet_topk: batch %d, num_candidates %d (need >= 1 mention and the answer slot + 1 candidate)", batch,
              num_candidates);
    return DRIN_E_SHAPE;
  }
  if (num_topk < 0 || num_topk > kMaxTopk || (num_topk > 0 && topk == nullptr)) {
    set_error("drin_triplet_topk: num_topk %d outside 0..%d", num_topk, kMaxTopk);
    return DRIN_E_SHAPE;
  }
  for (int q = 0; q < num_topk; ++q)
    if (topk[q] < 1 || topk[q] > num_candidates - 1) {  // torch.topk raises beyond the row length
      set_error("drin_triplet_topk: top-k %d outside 1..%d", topk[q], num_candidates - 1);
      return DRIN_E_SHAPE;
    }
  if (scores == nullptr || answer == nullptr || loss == nullptr || workspace == nullptr ||
      (num_topk > 0 && correct == nullptr)) {
    set_error("drin_triplet_topk: null scores / answer / loss / workspace / correct");
    return DRIN_E_NULL;
  }
  if (workspace_bytes < drin_loss_workspace_bytes(batch)) {
    set_error("drin_triplet_topk: workspace %zu < %zu bytes", workspace_bytes, drin_loss_workspace_bytes(batch));
    return DRIN_E_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* pos = static_cast<float*>(workspace);
  float* partial = pos + batch;
  int* hits = reinterpret_cast<int*>(partial + batch);
  TopkList tk{};
  tk.n = num_topk;
  for (int q = 0; q < num_topk; ++q) tk.k[q] = topk[q];
  // utils.py:42-43: mean over the B (N-1) matrix per mention, then / B
  const float scale = (float)(1.0 / ((double)batch * (double)batch * (double)(num_candidates - 1)));
  KernelTimer timer(DRIN_KC_EDGE, st);
  hipLaunchKernelGGL(k_triplet_rows, dim3(batch), dim3(64), 0, st, scores, answer, pos, hits, num_candidates, tk);
  DRIN_CHECK_LAUNCH("k_triplet_rows");
  hipLaunchKernelGGL(k_triplet_pairs, dim3(batch), dim3(256), 0, st, scores, answer, pos, partial, d_scores, batch,
                     num_candidates, margin, scale);
  DRIN_CHECK_LAUNCH("k_triplet_pairs");
  hipLaunchKernelGGL(k_triplet_finish, dim3(1), dim3(256), 0, st, partial, hits, loss, num_topk > 0 ? correct : nullptr,
                     batch, scale, num_topk);
  DRIN_CHECK_LAUNCH("k_triplet_finish");
  return DRIN_OK;
}

}  // extern "C"
