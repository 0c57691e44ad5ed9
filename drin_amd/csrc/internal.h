// Host-side internals of libdrin_hip.so: launch wrappers, error plumbing, workspace layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "../../include/drin_hip.h"

namespace drin {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define DRIN_CHECK_LAUNCH(what)                                  \
  do {                                                           \
    hipError_t _e = hipGetLastError();                           \
    if (_e != hipSuccess) return ::drin::hip_fail(_e, what);     \
  } while (0)

#define DRIN_TRY(expr)              \
  do {                              \
    int _s = (expr);                \
    if (_s != DRIN_OK) return _s;   \
  } while (0)

// Brackets the launches issued during its lifetime with hipEvents when a profile is open on this
// thread (drin_profile_begin); otherwise free.
struct KernelTimer {
  KernelTimer(int kernel_class, hipStream_t st);
  ~KernelTimer();
  int slot;
  hipStream_t stream;
};

// The device a call runs on is the device of the caller's STREAM, not whatever device happens to be current on the calling
// thread: every entry point that launches opens a DeviceScope first.  It asks the stream (hipStreamGetDevice) - or, for the NULL
// stream, the device that owns `device_pointer` (hipPointerGetAttributes; the NULL stream of THAT device is then used) - makes that
// device current for the duration of the call and restores the previous one on the way out.  A process whose current device is 0
// may therefore score tensors that live on device 1 through a stream of device 1: kernel attributes, launches, events and memsets
// all land on device 1.  Scopes nest (an entry point calling another one); the innermost one answers call_device().
struct DeviceScope {
  DeviceScope(void* stream, const void* device_pointer, const char* entry_point);
  ~DeviceScope();
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
  int status;        // DRIN_OK, or DRIN_E_HIP with the message set
  int prev, dev, outer;
  bool switched;
};
// device of the innermost DeviceScope of this thread (outside any scope: the current device)
int call_device();
#define DRIN_BIND_DEVICE(stream, device_pointer, entry_point)               \
  ::drin::DeviceScope _device_scope((stream), (device_pointer), (entry_point)); \
  if (_device_scope.status != DRIN_OK) return _device_scope.status

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: one high-water mark per (call site,
// device), so a process that runs on cuda:0 and later on cuda:1 opts in on both, and a later call that needs more LDS
// than the first one raises it.  The device is the CALL's (DeviceScope), which the entry point has made current.
// Atomics: the caller's thread and autograd's may race here harmlessly (idempotent call).
struct DynLdsOptIn {
  std::atomic<int> bytes[64] = {};
};
inline int ensure_dynamic_lds(DynLdsOptIn& s, const void* kernel, int bytes, const char* what) {
  const int dev = call_device();
  std::atomic<int>& mark = s.bytes[dev & 63];
  if (mark.load(std::memory_order_relaxed) >= bytes) return DRIN_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return hip_fail(e, what);
  mark.store(bytes, std::memory_order_relaxed);
  return DRIN_OK;
}

// roctx range over one entry point (rocprofv3 --marker-trace shows the calls of the path as named ranges above their
// kernels).  Inert unless DRIN_ROCTX is set in the environment when the first entry point runs; the roctx library is
// dlopen'ed then (librocprofiler-sdk-roctx.so, else libroctx64.so) - the library has no link-time dependency on it.
struct RoctxRange {
  explicit RoctxRange(const char* name);
  ~RoctxRange();
  bool pushed;
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// grid of the one-wave-per-pair kernels (device_utils.h: wave_pair)
inline dim3 pair_grid(int B, int N) {
  return (B > 1 && B <= 65535) ? dim3((unsigned)cdiv(N, 4), (unsigned)B) : dim3((unsigned)cdiv((int64_t)B * N, 4));
}

// ---- streaming / pooling kernels (stream_kernels.hip) -------------------------------------------
// out[g, c] = mean_s in[g, s, c]
int launch_axis_mean(const float* in, float* out, int64_t groups, int inner, int cols, hipStream_t st);
// Avg.avg (ghmfc.py:54-60): out[b, :] = mean(seq[b, start[b]:end[b], :])
int launch_span_mean(const float* seq, const int64_t* start, const int64_t* end, float* out, int B, int L, int D,
                     hipStream_t st);
// the same two for features stored as bf16 (drin_config.feature_dtype = DRIN_FEAT_BF16)
int launch_axis_mean_bf16(const void* in, float* out, int64_t groups, int inner, int cols, hipStream_t st);
int launch_span_mean_bf16(const void* seq, const int64_t* start, const int64_t* end, float* out, int B, int L, int D,
                          hipStream_t st);
// ghmfc.py:245-249: out[p, :] = mean(feat[p, 1:ntok-1, :]), ntok = sum(mask[p, :])
int launch_entity_token_mean(const float* feat, const int64_t* mask, float* out, int64_t pairs, int T, int D,
                             hipStream_t st);
int launch_entity_token_mean_bf16(const void* feat, const int64_t* mask, float* out, int64_t pairs, int T, int D,
                                  hipStream_t st);
// out[b*N + n] = scale * cos(x[b, :], y[(b*N + n) * y_stride : +D])
// (sim1 / sim2 -> out1 / out2: optional [B*N] arrays divided by `div` in the same pass: the CLIP edges of model.py:203)
int launch_cosine_rows(const float* x, const float* y, int64_t y_stride, float* out, int B, int N, int D, float eps,
                       float scale, hipStream_t st, const int64_t* y_index = nullptr, const float* sim1 = nullptr,
                       const float* sim2 = nullptr, float* out1 = nullptr, float* out2 = nullptr, float div = 1.0f);
// model.py:84-92: weighted object-pair similarity
int launch_miei(const float* mobj, const float* mscore, const float* eobj, const float* escore, float* out, int B,
                int N, int Km, int Ke, int R, float cos_eps, float miei_eps, float scale, hipStream_t st,
                const int64_t* e_index = nullptr);
// out[i] = in[i] * mul / div
int launch_scale_div(const float* in, float* out, int64_t n, float mul, float div, hipStream_t st);

// ---- ordered slice sums (gemm_f32.hip) -----------------------------------------------------------
// y[r, c] += sum over the destination's segments (in the order they were added), over each segment's slices (in order), of
// partial[slice][r][c].  Every split reduction of the backward pass - the weight-gradient products split over the B N pairs,
// their bias sums, the column sums - stores its slices plainly and lands in its destination through this ONE kernel: no
// fp32 atomics anywhere, the same bits every run.  Two products that add to the same destination (dW_h takes the mention
// and the entity rows) become two segments of one entry, summed by one thread in a fixed order.
struct SliceSum {
  static constexpr int MAX_DST = 32, MAX_SEG = 40;
  struct Dst {
    float* y;
    int64_t ldy;
    int rows, c4;          // the destination is rows x 4 c4 floats
    unsigned first_block;
    int seg_head, seg_tail;
  } dst[MAX_DST];
  struct Seg {
    const float* partial;  // [slices][rows][4 c4], contiguous
    int slices, next;
  } seg[MAX_SEG];
  int n = 0, n_seg = 0;
  // DRIN_E_ALIGN / DRIN_E_SHAPE outside the contract (cols % 4, 16-byte alignment of y / partial, ldy % 4 unless rows == 1,
  // the same y with another shape, more than MAX_DST destinations / MAX_SEG segments)
  int add(float* y, int64_t ldy, int rows, int cols, const float* partial, int slices);
};
int launch_slice_sum(SliceSum& s, hipStream_t st);   // no-op for an empty one

// ---- GEMM (gemm_f32.hip) ------------------------------------------------------------------------
// y[m, n] (+)= sum_k x[m, k] * w[n, k] + bias[n];  x row stride ldx, w row stride ldw, y row stride ldy
// (splitk: optional scratch; mention-sized exact-fp32 products with K >= 512 then split K over workgroups into it
//  and reduce in order - deterministic - instead of walking K serially in 24 tiles)
// (w_planes: optional bf16 hi / lo planes of w - launch_split_planes_batch - taken when the product runs split-bf16)
int launch_gemm_nt(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                   int64_t M, int N, int K, bool accumulate, int precision, hipStream_t st, float* splitk = nullptr,
                   size_t splitk_floats = 0, const float* w_planes = nullptr);
// same contraction, operands split into bf16 hi + lo, three bf16 MFMAs, fp32 accumulate (gemm_bf16x3.hip)
// (w_hi, w_lo: optional pre-split planes of w - then w itself is not read and the weights stream by LDS-DMA)
int launch_gemm_nt_bf16x3(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y,
                          int64_t ldy, int64_t M, int N, int K, hipStream_t st, const void* w_hi = nullptr,
                          const void* w_lo = nullptr, bool accumulate = false,
                          float* tail = nullptr, size_t tail_floats = 0, const int64_t* a_index = nullptr);
// (a_index: row m of x is row a_index[m] of a table)
// one fp16 plane of x under ONE power-of-two scale: out = fp16(x / s), s -> scale[0] (scale: two floats, [1] is scratch); n % 4 == 0
int launch_to_f16_scaled(const float* x, void* out, int64_t n, float* scale, hipStream_t st);
// (tail: optional scratch; a partly filled last round of 256 x 256 tiles is then split along K over the idle CUs)
// (accumulate: y += ... instead of y = ...)
// weight-gradient contraction y[n, k] += sum_m a[m, n] b[m, k] in split-bf16 (gemm_tn_bf16x3.hip)
bool gemm_tn_bf16x3_fits(int64_t lda, int64_t ldb, int64_t M, int N, int K, const void* a, const void* b);
// (scratch: [slices][N][K] floats, at least N K of them, 16-byte aligned - the slices store plainly and are added to y in
//  order: DRIN_E_WORKSPACE without it)
int launch_gemm_tn_bf16x3(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M,
                          int N, int K, hipStream_t st, float* scratch, size_t scratch_floats,
                          const int64_t* b_index = nullptr);
// what launch_gemm_tn_bf16x3 / launch_gemm_tn_group need of y and the scratch besides gemm_tn_bf16x3_fits
bool gemm_tn_bf16x3_scratch_ok(const float* y, int64_t ldy, int N, int K, const float* scratch, size_t scratch_floats);
// (b_index: reduction row m of b is row b_index[m] of a table - the gathered form never materialised)
// up to 8 such products in ONE launch (+ one for the slice reduction): the chip's workgroups are dealt over all of them
struct TnGroup {
  static constexpr int MAX = 8;
  struct Item {
    const float* a;
    int64_t lda;
    const float* b;
    int64_t ldb;
    float* y;
    int64_t ldy, M;
    int N, K;
    const int64_t* b_index;
    float* colsum;
  } item[MAX];
  int n = 0;
  // no-op for y == NULL; DRIN_E_SHAPE outside gemm_tn_bf16x3_fits or past MAX
  // (colsum: optional [N], += the column sums of a - the bias gradient that belongs to dW = dY^T X, from the rows the
  //  product stages anyway)
  int add(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int N, int K,
          const int64_t* b_index = nullptr, float* colsum = nullptr);
};
// (defer: the slice sums are appended to it instead of being launched - the caller launches it, after which the scratch
//  may be reused; NULL: launched here)
// (target_rows: the slice length to deal the products by, 0 = tn_group_target of this group - a caller that launches one
//  group in two parts passes the whole group's target to both, so that every product keeps the slices, hence the bits, of
//  the single launch)
int launch_gemm_tn_group(const TnGroup& g, hipStream_t st, float* scratch, size_t scratch_floats, SliceSum* defer = nullptr,
                         int64_t target_rows = 0);
int64_t tn_group_target(const TnGroup& g, size_t scratch_floats);
// same, operands pre-split into bf16 hi / lo planes (gemm_x3_planes.hip); K % 32 == 0
// (a_lo NULL: A exact in bf16, two MFMAs per tile pair)
int launch_gemm_x3_planes(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo,
                          int64_t ldb, const float* bias, float* y, int64_t ldy, int64_t M, int N, int K,
                          hipStream_t st, float* splitk = nullptr, size_t splitk_floats = 0);
// y = a b^T in ONE fp16 MFMA pass on single fp16 planes (DRIN_PREC_BF16X3_IF16; gemm_x3_planes.hip): a_f16 [M][K] holds row m of the
// activation divided by row_scale[m], b_f16 [N][K] the weight divided by *b_scale (powers of two: exact); output row m is multiplied
// by row_scale[m] * *b_scale.  Whole 256 x 256 tiles, K % 64 == 0, 16-byte aligned: gemm_f16_planes_fits, DRIN_E_UNSUPPORTED otherwise.
bool gemm_f16_planes_fits(const void* a_f16, int64_t lda, const void* b_f16, int64_t ldb, const float* y, int64_t ldy, int64_t M, int N, int K);
int launch_gemm_f16_planes(const void* a_f16, int64_t lda, const void* b_f16, int64_t ldb, const float* row_scale, const float* b_scale,
                           float* y, int64_t ldy, int64_t M, int N, int K, hipStream_t st, float* tail = nullptr, size_t tail_floats = 0);
// the four-phase pipeline (gemm_x3_planes.hip) on an fp32 x against pre-split weight planes: whole 256 x 256 tiles
int launch_gemm_nt_bf16x3_p4(const float* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw, const float* bias,
                             float* y, int64_t ldy, int64_t M, int N, int K, hipStream_t st, bool accumulate, float* tail = nullptr,
                             size_t tail_floats = 0);
// y[tail tile] += its ksplit - 1 partial 256 x 256 tiles, in order (tail split of the two split-bf16 NT kernels)
int launch_tail_add_256(const float* tail, float* y, int64_t ldy, int64_t M, int N, unsigned col_tiles, unsigned full,
                        unsigned tail_tiles, int ksplit, hipStream_t st);
// y[m, n] (+)= bias[n] + sum_z partial[z][m][n] in order (the reduction step of every split-K product; gemm_f32.hip)
int launch_splitk_reduce(const float* partial, int splits, int64_t part_stride, const float* bias, float* y, int64_t ldy,
                         int64_t M, int N, bool accumulate, hipStream_t st);
// fp32 -> bf16 hi / lo planes (n % 4 == 0)
int launch_split_planes(const float* x, void* hi, void* lo, int64_t n, hipStream_t st);
// up to 32 tensors split in ONE launch; planes_out holds numel bf16 of hi followed by numel bf16 of lo (= numel floats).
// launch_transpose_split_batch: the same for the TRANSPOSES of equally shaped [rows][cols] matrices.
struct SplitBatch {
  const float* src[32];
  float* planes[32];
  int64_t n4[32];
  int n = 0;
  int add(const float* w, float* planes_out, int64_t numel);   // no-op for w == NULL
};
int launch_split_planes_batch(const SplitBatch& b, hipStream_t st);
int launch_transpose_split_batch(const SplitBatch& b, int rows, int cols, hipStream_t st);
// y[m, n] (+)= sum_k x[m, k] * w[k, n]        (used by backward: dX = dY * W)
int launch_gemm_nn(const float* x, int64_t ldx, const float* w, int64_t ldw, float* y, int64_t ldy, int64_t M, int N,
                   int K, bool accumulate, int precision, hipStream_t st, float* splitk = nullptr, size_t splitk_floats = 0);
// up to 8 MENTION-sized (M <= 2048 reduction rows) exact-fp32 products y[n, k] += sum_m a[m, n] b[m, k] in one launch
struct F32GemmGroup {
  static constexpr int MAX = 8;
  struct Item {
    const float* a;
    int64_t lda;
    const float* b;
    int64_t ldb;
    float* y;
    int64_t ldy, M;
    int N, K;
  } item[MAX];
  const float* bias_of[MAX];
  int n = 0;
  int add_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int N, int K);
  // y = x w^T + bias (item fields: a = x, b = w, M = rows, N = outputs, K = reduction); only for products that passed
  // gemm_nt_f32_group_fits - the ones launch_gemm_nt would run as the exact-fp32 split-K kernel + slice reduction
  int add_nt(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy, int64_t M, int N, int K);
};
// (a product alone on its destination with one slice adds to it in place; any other stores its slices to the scratch and
//  goes through the slice sum - defer as for launch_gemm_tn_group; floats needed: small_tn_scratch_floats per item)
int launch_gemm_tn_f32_group(const F32GemmGroup& g, hipStream_t st, float* scratch, size_t scratch_floats, SliceSum* defer = nullptr);
inline int small_tn_slices(int64_t reduction_rows) {   // slices of a mention-sized (<= 2048 rows) weight-gradient product
  const int64_t s = (reduction_rows + 127) / 128;
  return (int)(s > 4 ? 4 : (s < 1 ? 1 : s));
}
bool gemm_nt_f32_group_fits(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y, int64_t ldy, int64_t M, int N,
                            int K, int precision);
int launch_gemm_nt_f32_group(const F32GemmGroup& g, hipStream_t st, float* scratch, size_t scratch_floats);
// Two NT products that do not depend on each other.  When both would run as the exact-fp32 split-K kernel + slice reduction
// (a few hundred rows: 6-9 us + 5 us each, nearly all fill and drain) the two kernels share a launch and so do the two
// reductions - same slices, same order, same bits; otherwise two launch_gemm_nt calls (planes: as its w_planes).
struct NtProduct {
  const float *x, *w, *bias;
  float* y;
  int64_t ldx, ldw, ldy, rows;
  int n_out, k_red;
  const float* planes;
};
int launch_gemm_nt_pair(const NtProduct& a, const NtProduct& b, int precision, hipStream_t st, float* splitk, size_t splitk_floats);
// y[n, k] += sum_m a[m, n] * b[m, k]          (used by backward: dW = dY^T * X), split over m
int launch_gemm_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int N,
                   int K, int precision, hipStream_t st, float* scratch = nullptr, size_t scratch_floats = 0);

// ---- GCN elementwise / reduction kernels (gcn_kernels.hip) --------------------------------------
// model.py:143-144 + :128 input: out[b,:] = mean_n(e1[b,n] v1[b,n,:]) + mean_n(e2[b,n] v2[b,n,:]) + u[b,:]
int launch_mention_aggregate(const float* e1, const float* v1, const float* e2, const float* v2, const float* u,
                             float* out, int B, int N, int D, hipStream_t st);
// model.py:146 + :128 input: out[b,n,:] = e1[b,n] m1[b,:] + e2[b,n] m2[b,:] + v[b,n,:]
int launch_entity_aggregate(const float* e1, const float* m1, const float* e2, const float* m2, const float* v,
                            float* out, int B, int N, int D, hipStream_t st);
// both aggregations of a scalar-edge layer in one pass; e = [4][edge_stride], vm = [2][B][D], ve = [2][B*N][D]
int launch_layer_aggregate(const float* e, int64_t edge_stride, const float* vm, const float* ve, float* agg_m,
                           float* agg_e, int B, int N, int D, bool live_image, hipStream_t st);
// model.py:128: y = gelu(layer_norm(h)) row-wise; optionally keeps mean / rstd for backward
// (act: drin_activation of the vertices, resolved - DRIN_ACT_GELU by default; the name keeps the reference's default)
int launch_layernorm_gelu(const float* h, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          int64_t rows, int D, float eps, hipStream_t st, int act = DRIN_ACT_GELU);
// two row segments (rows of h, then rows2 of h2) through the same LayerNorm in one launch
int launch_layernorm_gelu2(const float* h, float* y, float* mean, float* rstd, int64_t rows, const float* h2, float* y2,
                           float* mean2, float* rstd2, int64_t rows2, const float* gamma, const float* beta, int D, float eps,
                           hipStream_t st, int act = DRIN_ACT_GELU);
// model.py:148-153 + :133: out[b,n] = sigmoid(mean_d(fu[b,:] fv[b,n,:]) + e[b,n])
// (z_out: optional [4][B*N], the pre-activations - kept for backward when the activation has no derivative-from-output)
int launch_edge_update4(const float* fu, const float* fv, const float* e, float* out, int B, int N, int D,
                        hipStream_t st, int act = DRIN_ACT_SIGMOID, float* z_out = nullptr);

// ---- backward row kernels (backward_kernels.hip) ------------------------------------------------
// d cos(x[b], y[p]) : dx [B, D], dy [B*N, D]; scratch3 holds 3 * B*N floats
int launch_cosine_bwd(const float* x, const float* y, const float* g, float* dx, float* dy, float* scratch3, int B,
                      int N, int D, float eps, hipStream_t st);
// in place: g (dL/dy) -> dL/dh for y = gelu(LN(h)); accumulates dgamma, dbeta and dbias (= column sum of dL/dh);
// partial: (1024 + 16) * 3 * D floats of scratch for the per-block column sums
int launch_layernorm_gelu_bwd(const float* h, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, float* g, float* dgamma, float* dbeta, float* dbias, float* partial,
                              int64_t rows, int D, hipStream_t st, int act = DRIN_ACT_GELU);
// the same over two row segments that share gamma / beta and the column sums (mention + entity vertices of one layer)
int launch_layernorm_gelu_bwd2(const float* h, const float* mean, const float* rstd, float* g, int64_t rows, const float* h2,
                               const float* mean2, const float* rstd2, float* g2, int64_t rows2, const float* gamma,
                               const float* beta, float* dgamma, float* dbeta, float* dbias, float* partial, int D,
                               hipStream_t st, int act = DRIN_ACT_GELU, SliceSum* defer = nullptr, float* level1_own = nullptr);
// (defer + level1_own [3][16][D] floats: the second level of the column-sum reduction joins the caller's slice sum)
// out[c] += sum_rows x[row, c].  The rows are dealt over up to kColsumMaxSlices workgroups per 256 columns, whose sums go to
// the scratch ([slices][C] floats) and from there to out in order (SliceSum); without scratch one workgroup per 256 columns
// walks all rows and adds to out itself - slower, the same kind of result: no atomics either way.
constexpr int kColsumMaxSlices = 64;
int launch_colsum(const float* x, float* out, int64_t rows, int C, hipStream_t st, float* scratch = nullptr,
                  size_t scratch_floats = 0);
// up to 8 such column sums in ONE launch
struct ColsumBatch {
  const float* x[8];
  float* out[8];
  int64_t rows[8];
  int c4[8];
  int by[8];
  int n = 0;
  int add(const float* src, float* dst, int64_t nrows, int C);   // no-op for dst == NULL or nrows <= 0
};
// (scratch: sum over the entries of by * C floats <= 8 * kColsumMaxSlices * C; defer as for launch_gemm_tn_group)
int launch_colsum_batch(const ColsumBatch& b, hipStream_t st, float* scratch, size_t scratch_floats, SliceSum* defer = nullptr);
// sigmoid backward of the scalar edge update + both entity-side gradients dfv_t, dfv_i in one pass (see the kernel)
// (act | kActFromPre (device_utils.h: 0x100): e_new holds the pre-activation instead of the stored edge; also launch_sigmoid_bwd)
int launch_edge_update_bwd(const float* g, const float* e_new, const float* fu, float* dpre, float* dfv, int B, int N, int D,
                           float scale, hipStream_t st, int act = DRIN_ACT_SIGMOID);
// dpre = g * e' * (1 - e')
int launch_sigmoid_bwd(const float* g, const float* e_new, float* dpre, int64_t n, hipStream_t st, int act = DRIN_ACT_SIGMOID);
// out[b,:] = scale (sum_n w1 v1 + sum_n w2 v2) + u      (v2, u optional)
int launch_mention_reduce(const float* w1, const float* v1, const float* w2, const float* v2, const float* u,
                          float* out, int B, int N, int D, float scale, hipStream_t st);
// two such reductions over the same rows in one pass: w = [4][B*N]; outA takes planes 0 / 1, outB planes 2 / 3
int launch_mention_reduce2(const float* w, const float* v1, const float* v2, const float* uA, const float* uB,
                           float* outA, float* outB, int B, int N, int D, float scale, hipStream_t st);
// two independent launch_mention_reduce2 jobs (a, b) over the same B x N x D geometry in one launch; same bits
int launch_mention_reduce2_pair(const float* wa, const float* va1, const float* va2, const float* uaA, const float* uaB, float* outaA,
                                float* outaB, float scale_a, const float* wb, const float* vb1, const float* vb2, const float* ubA,
                                const float* ubB, float* outbA, float* outbB, float scale_b, int B, int N, int D, hipStream_t st);
// out[p,:] = scale (w1[p] m1[b,:] + w2[p] m2[b,:])       (m2 optional)
int launch_entity_combine(const float* w1, const float* m1, const float* w2, const float* m2, float* out, int B, int N,
                          int D, float scale, hipStream_t st);
// entity side of the aggregation backward + edge gradients (see kernel comment); mask = edge_enabled[4]
int launch_entity_side_bwd(const float* dA_mt, const float* dA_mi, const float* dA_et, const float* dA_ei,
                           const float* mt, const float* mi, const float* et, const float* ei, const float* e,
                           const float* de_extra, float* d_et, float* d_ei, float* de, int B, int N, int D,
                           const float* mask, bool accumulate, hipStream_t st);  // accumulate: d_et, d_ei += ...

// ---- vector-edge ablation (vector_kernels.hip) --------------------------------------------------
int launch_expand_edges(const float* es, float* out, int64_t pairs4, int D, hipStream_t st);
int launch_mention_reduce_vec(const float* w1, const float* v1, const float* w2, const float* v2, const float* u,
                              float* out, int B, int N, int D, float scale, bool mean_style, hipStream_t st);
int launch_entity_aggregate_vec(const float* e1, const float* m1, const float* e2, const float* m2, const float* v,
                                float* out, int B, int N, int D, hipStream_t st);
int launch_edge_pre_vec(const float* fu, const float* fv, const float* e, float* pre, int B, int N, int D,
                        hipStream_t st);
int launch_sigmoid_inplace(float* x, int64_t n, hipStream_t st, int act = DRIN_ACT_SIGMOID, float* z_out = nullptr);
int launch_edge_pre_vec_bwd(const float* dpre, float* dfu, float* dfv, int B, int N, int D, hipStream_t st);
int launch_entity_side_bwd_vec(const float* dA_mt, const float* dA_mi, const float* dA_et, const float* dA_ei,
                               const float* mt, const float* mi, const float* et, const float* ei, const float* e,
                               const float* de_extra, float* d_et, float* d_ei, float* de, int B, int N, int D,
                               const float* mask, hipStream_t st);

}  // namespace drin
