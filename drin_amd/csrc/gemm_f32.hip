// Generic fp32 GEMM on the CDNA4 matrix cores: C[m, n] (+)= sum_k a(m, k) * b(n, k) (+ bias[n]).
//
// The contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain, the fp32
// matrix rate 157 TF/s on MI355X), so results carry fp32 rounding only and parity with the
// reference's fp32 `nn.Linear` (<= 1e-4 on the final scores) holds without a reduced-precision
// caveat.  This is the any-shape kernel behind every Linear of the path and its backward; the
// D = 768 fused row-complete kernels build on the same tile code.
//
// Tile: 128 x 128 x 32 per 256-thread workgroup (4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA
// tiles, 64 accumulator VGPRs), two LDS buffers, register staging (global -> VGPR -> LDS) so the
// next K-block's loads are in flight under the current block's 64 MFMAs per wave.  Problems with
// at most 2048 rows (the mention side: B or 2B rows) use a 64 x 64 tile of the same code so that a
// [512 x 768] output is 96 workgroups instead of 24.
//
// Operand layouts.  Each operand may be given "k-contiguous" (src[row * ld + k], the nn.Linear
// forward case for both x and W) or "k-major" (src[k * ld + row], needed by the backward products
// dX = dY W and dW = dY^T X).  k-contiguous tiles sit in LDS as [row][32 + 4] and a lane fetches FOUR
// k-steps with one ds_read_b128; to make that possible the k index inside a group of 8 is permuted:
// MFMA step kk of group kg reads actual k = 8 kg + 4 h + kk in lane-half h (a sum over k does not
// care about the order, and both operands use the same map).  The 16-byte row pad keeps the
// ds_read_b128 lane groups on distinct banks.  k-major tiles sit as [32][128 + 4] and are read with
// conflict-free ds_read_b32 (consecutive lanes -> consecutive rows).
#include "device_utils.h"
#include "internal.h"

namespace drin {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_KC = BK + 4;    // k-contiguous row stride (floats)
// one operand tile of T rows: k-contiguous [T][36] or k-major [32][T + 4]; T * 36 covers both (T >= 32)
template <int T>
struct Tile {
  static constexpr int LDS_KM = T + 4;   // k-major row stride (floats)
  static constexpr int FLOATS = T * LDS_KC;
  static constexpr int PASSES = T / 32;  // float4 loads per thread
  static_assert(BK * LDS_KM <= FLOATS, "tile buffer too small");
};

template <int T>
struct Staged {
  float4 v[Tile<T>::PASSES];
};

// global -> registers.  rows: operand rows (m or n) of this tile start at row0, limit `rows`;
// k range of this block [k0, k0 + BK) clipped to k_end.
template <int T, bool KMAJOR>
__device__ __forceinline__ void load_tile(Staged<T>& s, const float* __restrict__ src, int64_t ld, int64_t row0,
                                          int64_t rows, int k0, int k_end) {
  const int t = threadIdx.x;
  if (!KMAJOR) {
    const int c4 = t & 7, r = t >> 3;
    const int k = k0 + c4 * 4;
#pragma unroll
    for (int i = 0; i < Tile<T>::PASSES; ++i) {
      const int64_t row = row0 + r + 32 * i;
      s.v[i] = (row < rows && k < k_end) ? ld4(src + row * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    constexpr int C4 = T / 4, KR = 256 / C4;  // float4 per k-row, k-rows per pass
    const int c4 = t % C4, kr = t / C4;
    const int64_t row = row0 + c4 * 4;
#pragma unroll
    for (int i = 0; i < Tile<T>::PASSES; ++i) {
      const int k = k0 + kr + KR * i;
      s.v[i] = (row < rows && k < k_end) ? ld4(src + (int64_t)k * ld + row) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int T, bool KMAJOR>
__device__ __forceinline__ void store_tile(const Staged<T>& s, float* __restrict__ lds) {
  const int t = threadIdx.x;
  if (!KMAJOR) {
    const int c4 = t & 7, r = t >> 3;
#pragma unroll
    for (int i = 0; i < Tile<T>::PASSES; ++i) st4(lds + (r + 32 * i) * LDS_KC + c4 * 4, s.v[i]);
  } else {
    constexpr int C4 = T / 4, KR = 256 / C4;
    const int c4 = t % C4, kr = t / C4;
#pragma unroll
    for (int i = 0; i < Tile<T>::PASSES; ++i) st4(lds + (kr + KR * i) * Tile<T>::LDS_KM + c4 * 4, s.v[i]);
  }
}

// fragment of 4 k-steps for the 32-row sub-tile starting at `row` (lane r = l & 31, half h = l >> 5)
template <int T, bool KMAJOR>
__device__ __forceinline__ float4 read_frag(const float* __restrict__ lds, int row, int kg, int r, int h) {
  if (!KMAJOR) {
    return ld4(lds + (row + r) * LDS_KC + kg * 8 + h * 4);
  } else {
    constexpr int S = Tile<T>::LDS_KM;
    const float* p = lds + (kg * 8 + h * 4) * S + row + r;
    return make_float4(p[0], p[S], p[2 * S], p[3 * S]);
  }
}

// PARTIAL: split z writes its tile to C + z * part_stride (a slice reduction adds them in order; there is no atomic epilogue:
// every split reduction of this library is reproducible bit for bit)
enum : int { GEMM_ACCUMULATE = 1, GEMM_PARTIAL = 4 };

// BM x BN output tile (128 or 64 each): 4 waves as 2 x 2, each wave (BM/2) x (BN/2)
// (tile_x / tile_y / split: the workgroup's column tile, row tile and K slice - blockIdx of the one-problem kernel)
template <int BM, int BN, bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void gemm_f32_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
                                              const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int64_t M,
                                              int N, int K, int k_per_split, int flags, int64_t part_stride, unsigned tile_x,
                                              unsigned tile_y, unsigned split) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TA = Tile<BM>::FLOATS, TB = Tile<BN>::FLOATS, TBUF = TA + TB;  // buffer c: A at c*TBUF, B after it
  constexpr int MI = BM / 64, NI = BN / 64;                                   // MFMA tiles per wave
  constexpr int WM = BM / 2, WN = BN / 2;

  const int64_t m0 = (int64_t)tile_y * BM;
  const int n0 = tile_x * BN;
  const int k_begin = split * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  const int nkb = (k_end - k_begin + BK - 1) / BK;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  Staged<BM> sa;
  Staged<BN> sb;
  if (nkb > 0) {
    load_tile<BM, A_KMAJOR>(sa, A, lda, m0, M, k_begin, k_end);
    load_tile<BN, B_KMAJOR>(sb, Bm, ldb, n0, N, k_begin, k_end);
    store_tile<BM, A_KMAJOR>(sa, smem);
    store_tile<BN, B_KMAJOR>(sb, smem + TA);
  }
  __syncthreads();

  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    const bool more = kb + 1 < nkb;
    if (more) {  // next block's global loads fly under this block's MFMAs
      load_tile<BM, A_KMAJOR>(sa, A, lda, m0, M, k_begin + (kb + 1) * BK, k_end);
      load_tile<BN, B_KMAJOR>(sb, Bm, ldb, n0, N, k_begin + (kb + 1) * BK, k_end);
    }
    const float* as = smem + cur * TBUF;
    const float* bs = as + TA;
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = read_frag<BM, A_KMAJOR>(as, wm * WM + i * 32, kg, r, h);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = read_frag<BN, B_KMAJOR>(bs, wn * WN + j * 32, kg, r, h);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (more) {
      store_tile<BM, A_KMAJOR>(sa, smem + (cur ^ 1) * TBUF);
      store_tile<BN, B_KMAJOR>(sb, smem + (cur ^ 1) * TBUF + TA);
    }
    __syncthreads();
  }

  // epilogue: C/D register v of a 32x32 tile is row (v & 3) + 8 (v >> 2) + 4 h, column r
  const bool first_split = split == 0;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = n0 + wn * WN + j * 32 + r;
      if (col >= N) continue;
      const float bv = (bias != nullptr && first_split) ? bias[col] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t row = m0 + wm * WM + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (row >= M) continue;
        float* dst = C + row * ldc + col;
        const float val = acc[i][j][v] + bv;
        if (flags & GEMM_PARTIAL) {
          dst[(int64_t)split * part_stride] = acc[i][j][v];  // bias is added by the reduction
        } else if (flags & GEMM_ACCUMULATE) {
          *dst += val;
        } else {
          *dst = val;
        }
      }
    }
}

template <int BM, int BN, bool A_KMAJOR, bool B_KMAJOR>
__global__ void __launch_bounds__(256, 2)
    k_gemm_f32(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
               const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
               int k_per_split, int flags, int64_t part_stride) {
  gemm_f32_tile<BM, BN, A_KMAJOR, B_KMAJOR>(A, lda, Bm, ldb, bias, C, ldc, M, N, K, k_per_split, flags, part_stride, blockIdx.x,
                                            blockIdx.y, blockIdx.z);
}

// Several mention-sized products of ONE operand layout in one launch (F32GemmGroup, internal.h): a 1-D grid, a problem's
// work items are (K slice, row tile, column tile), column tile fastest.  Each product alone is a launch of a few
// microseconds of work between a fill and a drain; the chain of them is what a B = 64 training step spends 0.2 ms on.
struct GroupItem {
  const float* a;
  const float* b;
  float* c;
  int64_t lda, ldb, ldc, M, part_stride;
  int N, K, k_per_split, flags;
  unsigned col_tiles, row_tiles, first;
};
struct GroupArgs {
  GroupItem p[F32GemmGroup::MAX];
  int n;
};
template <bool A_KMAJOR, bool B_KMAJOR>
__global__ void __launch_bounds__(256, 2) k_gemm_f32_group(const GroupArgs g) {
  unsigned t = blockIdx.x;
  int pi = 0;
  for (int i = 1; i < g.n; ++i) pi = t >= g.p[i].first ? i : pi;
  const GroupItem& P = g.p[pi];
  t -= P.first;
  const unsigned per_split = P.col_tiles * P.row_tiles;
  const unsigned split = t / per_split, rest = t - split * per_split;
  gemm_f32_tile<64, 64, A_KMAJOR, B_KMAJOR>(P.a, P.lda, P.b, P.ldb, nullptr, P.c, P.ldc, P.M, P.N, P.K, P.k_per_split, P.flags,
                                            P.part_stride, rest % P.col_tiles, rest / P.col_tiles, split);
}

template <int BM, int BN, bool A_KMAJOR, bool B_KMAJOR>
static int launch(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc,
                  int64_t M, int N, int K, int splits, int flags, hipStream_t st, const char* what,
                  int64_t part_stride = 0, int* splits_used = nullptr) {
  if (M <= 0 || N <= 0) return DRIN_OK;
  const int64_t mt = cdiv(M, BM);
  if (mt > 65535) {
    set_error("%s: %lld row tiles exceed the grid limit; split the batch", what, (long long)mt);
    return DRIN_E_SHAPE;
  }
  if (!aligned16(A) || !aligned16(B) || (lda % 4) || (ldb % 4)) {
    set_error("%s: operands must be 16-byte aligned with leading dimensions that are multiples of 4", what);
    return DRIN_E_ALIGN;
  }
  int kps = K;
  if (splits > 1 && K > 0) {   // (K <= 0: one empty slice - never a slice length of 0 to divide by)
    kps = (int)(cdiv(cdiv(K, splits), BK) * BK);
    splits = (int)cdiv(K, kps);
  } else {
    splits = 1;
  }
  if (splits_used) *splits_used = splits;
  const size_t lds = sizeof(float) * 2 * (Tile<BM>::FLOATS + Tile<BN>::FLOATS);
  static DynLdsOptIn opt_in;  // one per template instantiation; > 64 KiB of dynamic LDS needs the opt-in
  auto kern = k_gemm_f32<BM, BN, A_KMAJOR, B_KMAJOR>;
  DRIN_TRY(ensure_dynamic_lds(opt_in, reinterpret_cast<const void*>(kern), (int)lds, "hipFuncSetAttribute(gemm)"));
  dim3 grid((unsigned)cdiv(N, BN), (unsigned)mt, (unsigned)splits);
  KernelTimer timer(DRIN_KC_GEMM, st);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, A, lda, B, ldb, bias, C, ldc, M, N, K, kps, flags, part_stride);
  DRIN_CHECK_LAUNCH(what);
  return DRIN_OK;
}

static int check_precision(int precision, const char* what) {
  if (precision != DRIN_PREC_F32) {
    set_error("%s: precision %d is not built in the generic GEMM (fp32 MFMA only)", what, precision);
    return DRIN_E_UNSUPPORTED;
  }
  return DRIN_OK;
}

// y[m, n] (+)= bias[n] + sum_z partial[z][m][n], slices added in order (deterministic split-K of the mention-sized
// products: a 128-row problem has 24 output tiles and a serial K loop of 24 .. 64 stages - latency, not work)
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ partial, int splits, int64_t part_stride,
                                                       const float* __restrict__ bias, float* __restrict__ y, int64_t ldy,
                                                       int64_t M, int N4, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N4) return;
  const int64_t m = i / N4;
  const int c4 = (int)(i - m * N4);
  const int64_t off = m * (int64_t)N4 * 4 + (int64_t)c4 * 4;
  float4 s = bias != nullptr ? ld4(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < splits; ++z) s = s + ld4(partial + z * part_stride + off);
  float* dst = y + m * ldy + c4 * 4;
  if (accumulate) s = s + ld4(dst);
  st4(dst, s);
}

int launch_splitk_reduce(const float* partial, int splits, int64_t part_stride, const float* bias, float* y, int64_t ldy,
                         int64_t M, int N, bool accumulate, hipStream_t st) {
  KernelTimer timer(DRIN_KC_GEMM, st);
  hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)cdiv(M * (N / 4), 256)), dim3(256), 0, st, partial, splits, part_stride,
                     bias, y, ldy, M, N / 4, accumulate ? 1 : 0);
  DRIN_CHECK_LAUNCH("k_splitk_reduce");
  return DRIN_OK;
}

// ---- SliceSum (internal.h): y += its segments' slices, in order --------------------------------------------------------
__global__ void __launch_bounds__(256) k_slice_sum(const SliceSum g) {
  int di = 0;
  for (int i = 1; i < g.n; ++i) di = blockIdx.x >= g.dst[i].first_block ? i : di;
  const SliceSum::Dst& d = g.dst[di];
  const int64_t i = (int64_t)(blockIdx.x - d.first_block) * 256 + threadIdx.x;
  const int64_t elems = (int64_t)d.rows * d.c4;
  if (i >= elems) return;
  const int r = (int)(i / d.c4), c4 = (int)(i - (int64_t)r * d.c4);
  const int64_t stride = elems * 4;
  float* dst = d.y + (int64_t)r * d.ldy + c4 * 4;
  float4 total = ld4(dst);
  for (int si = d.seg_head; si >= 0; si = g.seg[si].next) {
    const float* p = g.seg[si].partial + i * 4;
    const int slices = g.seg[si].slices;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    int z = 0;
    for (; z + 4 <= slices; z += 4) {  // four chains (the loads of a trip are independent), fixed association
      s0 = s0 + ld4(p + z * stride);
      s1 = s1 + ld4(p + (z + 1) * stride);
      s2 = s2 + ld4(p + (z + 2) * stride);
      s3 = s3 + ld4(p + (z + 3) * stride);
    }
    for (; z < slices; ++z) s0 = s0 + ld4(p + z * stride);
    total = total + ((s0 + s1) + (s2 + s3));
  }
  st4(dst, total);
}

int SliceSum::add(float* y, int64_t ldy, int rows, int cols, const float* partial, int slices) {
  if (y == nullptr || partial == nullptr || rows <= 0 || cols <= 0 || slices <= 0) return DRIN_OK;
  if ((cols % 4) || !aligned16(y) || !aligned16(partial) || (rows > 1 && (ldy % 4))) {
    set_error("slice sum: %d x %d destination (ld %lld) outside the kernel's contract (16-byte rows)", rows, cols, (long long)ldy);
    return DRIN_E_ALIGN;
  }
  int di = -1;
  for (int i = 0; i < n; ++i)
    if (dst[i].y == y) di = i;
  if (di >= 0 && (dst[di].rows != rows || dst[di].c4 != cols / 4 || (rows > 1 && dst[di].ldy != ldy))) {
    set_error("slice sum: one destination given as %d x %d and as %d x %d", dst[di].rows, dst[di].c4 * 4, rows, cols);
    return DRIN_E_SHAPE;
  }
  if ((di < 0 && n == MAX_DST) || n_seg == MAX_SEG) {
    set_error("internal: slice sum over more than %d destinations / %d segments", MAX_DST, MAX_SEG);
    return DRIN_E_SHAPE;
  }
  const int si = n_seg++;
  seg[si] = {partial, slices, -1};
  if (di < 0) {
    dst[n++] = {y, ldy, rows, cols / 4, 0u, si, si};
  } else {
    seg[dst[di].seg_tail].next = si;
    dst[di].seg_tail = si;
  }
  return DRIN_OK;
}

int launch_slice_sum(SliceSum& s, hipStream_t st) {
  if (s.n == 0) return DRIN_OK;
  int64_t blocks = 0;
  for (int i = 0; i < s.n; ++i) {
    s.dst[i].first_block = (unsigned)blocks;
    blocks += cdiv((int64_t)s.dst[i].rows * s.dst[i].c4, 256);
  }
  KernelTimer timer(DRIN_KC_GEMM, st);
  hipLaunchKernelGGL(k_slice_sum, dim3((unsigned)blocks), dim3(256), 0, st, s);
  DRIN_CHECK_LAUNCH("k_slice_sum");
  return DRIN_OK;
}

// mention-sized fp32 products with a long reduction: split K over workgroups into `scratch`, then reduce in order
template <bool B_KMAJOR>
static int launch_small_splitk(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y,
                               int64_t ldy, int64_t M, int N, int K, bool accumulate, float* scratch, size_t scratch_floats,
                               hipStream_t st, const char* what) {
  int splits = K / 128;  // >= 4 stages of 32 per slice
  if (splits > 8) splits = 8;
  while (splits > 1 && (size_t)splits * M * N > scratch_floats) --splits;
  const int64_t part_stride = M * (int64_t)N;
  int used = 1;
  DRIN_TRY((launch<64, 64, false, B_KMAJOR>(x, ldx, w, ldw, nullptr, scratch, N, M, N, K, splits, GEMM_PARTIAL, st, what,
                                            part_stride, &used)));
  return launch_splitk_reduce(scratch, used, part_stride, bias, y, ldy, M, N, accumulate, st);
}

// y[m, n] (+)= bias[n] + sum_k x[m, k] w[n, k] for ONE mention's rows (M <= 2: its text and image vertex).  A GEMV per row: one
// wave per output column, the lanes stride over k in float4 steps (the weight row is read once, coalesced, and meets both
// activation rows), wave reduction, lane 0 stores.  The MFMA tile kernel spends 12 + 4 us on such a product (24 tiles of
// 64 x 64 walking K in slices, then the slice reduction); this is one launch of ~7 us - the scoring call of a single mention
// is a chain of nine of them: 0.263 -> 0.210 ms (WikiMEL-shaped), 0.188 -> 0.155 ms (WikiDiverse-shaped).  Exact fp32 (fma
// chain per lane, fixed reduction order).  Kept to the single-mention case on purpose: from two mentions up the host's
// launch rate, not these products, bounds the call (measured at 4 mentions: kernel time -30 us, wall time unchanged), and
// batches of any other size keep ONE summation order for their mention-sized products (a sub-batch scores bit-identically).
template <int MR>
__global__ void __launch_bounds__(256) k_gemv_rows(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                   int64_t ldw, const float* __restrict__ bias, float* __restrict__ y,
                                                   int64_t ldy, int M, int N, int K4, int accumulate) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  const float* wr = w + (int64_t)n * ldw;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  for (int c4 = lane; c4 < K4; c4 += 64) {
    const float4 wv = ld4(wr + c4 * 4);
#pragma unroll
    for (int m = 0; m < MR; ++m)
      if (m < M) acc[m] += dot4(ld4(x + (int64_t)m * ldx + c4 * 4), wv);
  }
  const float bv = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (m >= M) break;
    const float s = wave_sum(acc[m]);
    if (lane == 0) {
      float* dst = y + (int64_t)m * ldy + n;
      *dst = s + bv + (accumulate ? *dst : 0.f);
    }
  }
}

static bool gemv_fits(const float* x, int64_t ldx, const float* w, int64_t ldw, int64_t M, int K) {
  return M >= 1 && M <= 2 && (K % 4) == 0 && (ldx % 4) == 0 && (ldw % 4) == 0 && aligned16(x) && aligned16(w);
}

static int launch_gemv(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                       int64_t M, int N, int K, bool accumulate, hipStream_t st) {
  KernelTimer timer(DRIN_KC_GEMM, st);
  const dim3 grid((unsigned)cdiv(N, 4));
  hipLaunchKernelGGL(k_gemv_rows<2>, grid, dim3(256), 0, st, x, ldx, w, ldw, bias, y, ldy, (int)M, N, K / 4, accumulate ? 1 : 0);
  DRIN_CHECK_LAUNCH("k_gemv_rows");
  return DRIN_OK;
}

static bool small_splitk_fits(int64_t M, int N, int K, int64_t ldy, const float* y, float* scratch, size_t scratch_floats) {
  return scratch != nullptr && M <= 512 && K >= 512 && (N % 4) == 0 && (ldy % 4) == 0 && aligned16(y) && aligned16(scratch) &&
         scratch_floats >= (size_t)2 * M * N;
}

// the slice reductions of a group of split-K products in one launch (k_splitk_reduce per item)
struct ReduceGroupArgs {
  const float* partial[F32GemmGroup::MAX];
  const float* bias[F32GemmGroup::MAX];
  float* y[F32GemmGroup::MAX];
  int64_t ldy[F32GemmGroup::MAX], M[F32GemmGroup::MAX], part_stride[F32GemmGroup::MAX];
  int N4[F32GemmGroup::MAX], splits[F32GemmGroup::MAX];
  unsigned first[F32GemmGroup::MAX];
  int n;
};
__global__ void __launch_bounds__(256) k_splitk_reduce_group(const ReduceGroupArgs g) {
  int pi = 0;
  for (int i = 1; i < g.n; ++i) pi = blockIdx.x >= g.first[i] ? i : pi;
  const int64_t i = (int64_t)(blockIdx.x - g.first[pi]) * 256 + threadIdx.x;
  const int N4 = g.N4[pi];
  if (i >= g.M[pi] * N4) return;
  const int64_t m = i / N4;
  const int c4 = (int)(i - m * N4);
  const int64_t off = m * (int64_t)N4 * 4 + (int64_t)c4 * 4;
  float4 s = g.bias[pi] != nullptr ? ld4(g.bias[pi] + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < g.splits[pi]; ++z) s = s + ld4(g.partial[pi] + z * g.part_stride[pi] + off);
  st4(g.y[pi] + m * g.ldy[pi] + c4 * 4, s);
}

bool gemm_nt_f32_group_fits(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y, int64_t ldy, int64_t M, int N,
                            int K, int precision) {
  // exactly the products launch_gemm_nt sends to the exact-fp32 split-K pair (kernel + slice reduction)
  if (precision == DRIN_PREC_BF16X3_ALL || (precision == DRIN_PREC_BF16X3 && M >= 256)) return false;
  if (precision != DRIN_PREC_F32 && precision != DRIN_PREC_BF16X3) return false;
  if ((K % 4) || gemv_fits(x, ldx, w, ldw, M, K)) return false;
  return M >= 1 && M <= 512 && K >= 512 && (N % 4) == 0 && (ldy % 4) == 0 && (ldx % 4) == 0 && (ldw % 4) == 0 && aligned16(y) &&
         aligned16(x) && aligned16(w);
}

int F32GemmGroup::add_nt(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy, int64_t M,
                         int N, int K) {
  if (n == MAX) {
    set_error("internal: more than %d products in one exact-fp32 group", MAX);
    return DRIN_E_SHAPE;
  }
  item[n] = {x, ldx, w, ldw, y, ldy, M, N, K};
  bias_of[n++] = bias;
  return DRIN_OK;
}

// y = x w^T + bias for every item (each one passed gemm_nt_f32_group_fits): the split-K kernel of all items in one launch,
// their slice reductions in a second - the same slices in the same order as launch_gemm_nt's own pair of launches
int launch_gemm_nt_f32_group(const F32GemmGroup& grp, hipStream_t st, float* scratch, size_t scratch_floats) {
  if (grp.n == 0) return DRIN_OK;
  GroupArgs ga;
  ReduceGroupArgs ra;
  ga.n = ra.n = grp.n;
  unsigned items = 0, blocks = 0;
  size_t used = 0;
  for (int i = 0; i < grp.n; ++i) {
    const auto& it = grp.item[i];
    if (it.M <= 0 || it.N <= 0 || it.K <= 0) {
      set_error("gemm_nt group: item %d has M=%lld N=%d K=%d", i, (long long)it.M, it.N, it.K);
      return DRIN_E_SHAPE;
    }
    int splits = it.K / 128;   // as launch_small_splitk
    if (splits > 8) splits = 8;
    if (splits < 1) splits = 1;
    const int kps = (int)(cdiv(cdiv(it.K, splits), BK) * BK);
    splits = (int)cdiv(it.K, kps);
    const int64_t part_stride = it.M * (int64_t)it.N;
    if (scratch == nullptr || !aligned16(scratch) || used + (size_t)splits * part_stride > scratch_floats) {
      set_error("gemm_nt group: split-K scratch of %zu floats is too small", scratch_floats);
      return DRIN_E_WORKSPACE;
    }
    auto& P = ga.p[i];
    P.a = it.a, P.b = it.b, P.c = scratch + used, P.lda = it.lda, P.ldb = it.ldb, P.ldc = it.N;
    P.M = it.M, P.N = it.N, P.K = it.K, P.k_per_split = kps, P.flags = GEMM_PARTIAL, P.part_stride = part_stride;
    P.col_tiles = (unsigned)cdiv(it.N, 64), P.row_tiles = (unsigned)cdiv(it.M, 64), P.first = items;
    items += P.col_tiles * P.row_tiles * (unsigned)splits;
    ra.partial[i] = scratch + used, ra.bias[i] = grp.bias_of[i], ra.y[i] = it.y, ra.ldy[i] = it.ldy, ra.M[i] = it.M;
    ra.part_stride[i] = part_stride, ra.N4[i] = it.N / 4, ra.splits[i] = splits, ra.first[i] = blocks;
    blocks += (unsigned)cdiv(it.M * (it.N / 4), 256);
    used += (size_t)splits * part_stride;
    used = (used + 3) & ~(size_t)3;
  }
  const size_t lds = sizeof(float) * 2 * (Tile<64>::FLOATS + Tile<64>::FLOATS);
  static DynLdsOptIn opt_in;
  auto kern = k_gemm_f32_group<false, false>;
  DRIN_TRY(ensure_dynamic_lds(opt_in, reinterpret_cast<const void*>(kern), (int)lds, "hipFuncSetAttribute(gemm group)"));
  KernelTimer timer(DRIN_KC_GEMM, st);
  hipLaunchKernelGGL(kern, dim3(items), dim3(256), lds, st, ga);
  DRIN_CHECK_LAUNCH("k_gemm_f32_group");
  hipLaunchKernelGGL(k_splitk_reduce_group, dim3(blocks), dim3(256), 0, st, ra);
  DRIN_CHECK_LAUNCH("k_splitk_reduce_group");
  return DRIN_OK;
}

int launch_gemm_nt(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                   int64_t M, int N, int K, bool accumulate, int precision, hipStream_t st, float* splitk,
                   size_t splitk_floats, const float* w_planes) {
  // split-bf16 contraction for the pair-sized GEMMs; the mention-sized ones (a few hundred rows) stay on
  // the exact fp32 kernel: they are latency-bound, not rate-bound
  // (the 256-row threshold measured again in round 2, same box, 16 / 64 rows instead: a 64-mention scoring call 0.551 -> 0.576 ms,
  //  the B = 64 training step 1.406 -> 1.424 - below a tile row of the 256-wide kernel the exact one is as fast and exact)
  if (!accumulate && ((precision == DRIN_PREC_BF16X3 && M >= 256) || precision == DRIN_PREC_BF16X3_ALL)) {
    // (pre-split weight planes: contiguous [N][K] weights only, K a multiple of 32 - what the LDS-DMA path reads)
    const bool planes = w_planes != nullptr && ldw == K && (K % 32) == 0;
    const __bf16* hi = reinterpret_cast<const __bf16*>(w_planes);
    return launch_gemm_nt_bf16x3(x, ldx, w, ldw, bias, y, ldy, M, N, K, st, planes ? (const void*)hi : nullptr,
                                 planes ? (const void*)(hi + (int64_t)N * K) : nullptr, false, splitk, splitk_floats);
  }
  if (precision == DRIN_PREC_BF16X3 || precision == DRIN_PREC_BF16X3_ALL) precision = DRIN_PREC_F32;
  DRIN_TRY(check_precision(precision, "gemm_nt"));
  if (K % 4) {
    set_error("gemm_nt: K=%d must be a multiple of 4", K);
    return DRIN_E_SHAPE;
  }
  if (gemv_fits(x, ldx, w, ldw, M, K)) return launch_gemv(x, ldx, w, ldw, bias, y, ldy, M, N, K, accumulate, st);
  if (small_splitk_fits(M, N, K, ldy, y, splitk, splitk_floats))
    return launch_small_splitk<false>(x, ldx, w, ldw, bias, y, ldy, M, N, K, accumulate, splitk, splitk_floats, st, "gemm_nt");
  // mention-sized problems (a few hundred rows): 64 x 64 tiles give 4x the workgroups of 128 x 128 and
  // keep more of the chip busy on what is a latency-bound launch
  if (M <= 2048)
    return launch<64, 64, false, false>(x, ldx, w, ldw, bias, y, ldy, M, N, K, 1, accumulate ? GEMM_ACCUMULATE : 0, st,
                                        "gemm_nt");
  return launch<128, 128, false, false>(x, ldx, w, ldw, bias, y, ldy, M, N, K, 1, accumulate ? GEMM_ACCUMULATE : 0, st,
                                        "gemm_nt");
}

int launch_gemm_nt_pair(const NtProduct& a, const NtProduct& b, int precision, hipStream_t st, float* splitk, size_t splitk_floats) {
  const NtProduct* two[2] = {&a, &b};
  bool grouped = splitk != nullptr;
  size_t need = 8;
  for (const NtProduct* q : two) {
    grouped = grouped && gemm_nt_f32_group_fits(q->x, q->ldx, q->w, q->ldw, q->y, q->ldy, q->rows, q->n_out, q->k_red, precision);
    need += (size_t)8 * q->rows * q->n_out;
  }
  if (grouped && need <= splitk_floats) {
    F32GemmGroup g;
    for (const NtProduct* q : two) DRIN_TRY(g.add_nt(q->x, q->ldx, q->w, q->ldw, q->bias, q->y, q->ldy, q->rows, q->n_out, q->k_red));
    return launch_gemm_nt_f32_group(g, st, splitk, splitk_floats);
  }
  for (const NtProduct* q : two)
    DRIN_TRY(launch_gemm_nt(q->x, q->ldx, q->w, q->ldw, q->bias, q->y, q->ldy, q->rows, q->n_out, q->k_red, false, precision, st, splitk,
                            splitk_floats, q->planes));
  return DRIN_OK;
}

int launch_gemm_nn(const float* x, int64_t ldx, const float* w, int64_t ldw, float* y, int64_t ldy, int64_t M, int N,
                   int K, bool accumulate, int precision, hipStream_t st, float* splitk, size_t splitk_floats) {
  // y[m, n] = sum_k x[m, k] * w[k, n]: b(n, k) = w[k * ldw + n] is k-major
  if (precision == DRIN_PREC_BF16X3 || precision == DRIN_PREC_BF16X3_ALL) precision = DRIN_PREC_F32;  // backward stays exact fp32
  DRIN_TRY(check_precision(precision, "gemm_nn"));
  if ((K % 4) || (N % 4)) {
    set_error("gemm_nn: K=%d and N=%d must be multiples of 4", K, N);
    return DRIN_E_SHAPE;
  }
  if (small_splitk_fits(M, N, K, ldy, y, splitk, splitk_floats))
    return launch_small_splitk<true>(x, ldx, w, ldw, nullptr, y, ldy, M, N, K, accumulate, splitk, splitk_floats, st, "gemm_nn");
  if (M <= 2048)
    return launch<64, 64, false, true>(x, ldx, w, ldw, nullptr, y, ldy, M, N, K, 1, accumulate ? GEMM_ACCUMULATE : 0, st,
                                       "gemm_nn");
  return launch<128, 128, false, true>(x, ldx, w, ldw, nullptr, y, ldy, M, N, K, 1, accumulate ? GEMM_ACCUMULATE : 0, st,
                                       "gemm_nn");
}

int F32GemmGroup::add_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int N, int K) {
  if (y == nullptr || M <= 0 || N <= 0 || K <= 0) return DRIN_OK;
  if ((N % 4) || (K % 4) || M > 2048 || !aligned16(a) || !aligned16(b) || (lda % 4) || (ldb % 4)) {
    set_error("gemm_tn group: M=%lld N=%d K=%d outside the mention-sized kernel's contract", (long long)M, N, K);
    return DRIN_E_SHAPE;
  }
  if (n == MAX) {
    set_error("internal: more than %d products in one exact-fp32 group", MAX);
    return DRIN_E_SHAPE;
  }
  item[n++] = {a, lda, b, ldb, y, ldy, M, N, K};
  return DRIN_OK;
}

// y[n, k] += sum_m a[m, n] b[m, k] for every item: the mention-sized branch of launch_gemm_tn (64 x 64 tiles, up to four
// slices of the reduction), all items in one launch.  A product that is alone on its destination and has one slice adds
// to it in place (one workgroup per output tile: no race); any other stores its slices to the scratch and the slice sum
// adds them in order - two products of one destination as two segments of one entry.
int launch_gemm_tn_f32_group(const F32GemmGroup& grp, hipStream_t st, float* scratch, size_t scratch_floats, SliceSum* defer) {
  if (grp.n == 0) return DRIN_OK;
  GroupArgs ga;
  ga.n = grp.n;
  SliceSum local;
  SliceSum& sums = defer != nullptr ? *defer : local;
  unsigned items = 0;
  size_t used = 0;
  for (int i = 0; i < grp.n; ++i) {
    const auto& it = grp.item[i];
    // add_tn never stores an empty product; an item with a non-positive extent can only be a caller's slip (round 3's staged
    // backward once handed over value-initialised items past a group it had already flushed: M == 0 made kps 0 and the next
    // line's division raised SIGFPE on the host) - refused, never divided by
    if (it.M <= 0 || it.N <= 0 || it.K <= 0 || it.M > 2048 || it.a == nullptr || it.b == nullptr || it.y == nullptr) {
      set_error("gemm_tn group: item %d of %d has M=%lld N=%d K=%d (or a NULL operand) - outside [1, 2048] reduction rows", i,
                grp.n, (long long)it.M, it.N, it.K);
      return DRIN_E_SHAPE;
    }
    int splits = small_tn_slices(it.M);
    const int kps = (int)(cdiv(cdiv(it.M, splits), BK) * BK);
    splits = (int)cdiv(it.M, kps);
    bool shared = false;   // another product of this launch adds to the same destination
    for (int j = 0; j < grp.n; ++j) shared = shared || (j != i && grp.item[j].y == it.y);
    const bool in_place = splits == 1 && !shared;
    auto& P = ga.p[i];
    // the kernel's (M, N, K) are (output rows, output columns, reduction length) = (N, K, M) of the product
    P.a = it.a, P.b = it.b, P.lda = it.lda, P.ldb = it.ldb;
    P.M = it.N, P.N = it.K, P.K = (int)it.M, P.k_per_split = kps;
    if (in_place) {
      P.c = it.y, P.ldc = it.ldy, P.flags = GEMM_ACCUMULATE, P.part_stride = 0;
    } else {
      const size_t need = (size_t)splits * it.N * it.K;
      if (scratch == nullptr || !aligned16(scratch) || used + need > scratch_floats) {
        set_error("gemm_tn group: %zu floats of slice scratch, %zu needed", scratch_floats, used + need);
        return DRIN_E_WORKSPACE;
      }
      P.c = scratch + used, P.ldc = it.K, P.flags = GEMM_PARTIAL, P.part_stride = (int64_t)it.N * it.K;
      DRIN_TRY(sums.add(it.y, it.ldy, it.N, it.K, scratch + used, splits));
      used += (need + 3) & ~(size_t)3;
    }
    P.col_tiles = (unsigned)cdiv(it.K, 64), P.row_tiles = (unsigned)cdiv(it.N, 64), P.first = items;
    items += P.col_tiles * P.row_tiles * (unsigned)splits;
  }
  const size_t lds = sizeof(float) * 2 * (Tile<64>::FLOATS + Tile<64>::FLOATS);
  static DynLdsOptIn opt_in;
  auto kern = k_gemm_f32_group<true, true>;
  DRIN_TRY(ensure_dynamic_lds(opt_in, reinterpret_cast<const void*>(kern), (int)lds, "hipFuncSetAttribute(gemm group)"));
  {
    KernelTimer timer(DRIN_KC_GEMM, st);
    hipLaunchKernelGGL(kern, dim3(items), dim3(256), lds, st, ga);
    DRIN_CHECK_LAUNCH("k_gemm_f32_group");
  }
  return defer != nullptr ? DRIN_OK : launch_slice_sum(local, st);
}

int launch_gemm_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int N,
                   int K, int precision, hipStream_t st, float* scratch, size_t scratch_floats) {
  // y[n, k] += sum_m a[m, n] * b[m, k]: both operands k-major over the reduction index m.
  // The reduction runs over all B*N pairs while the output is one weight matrix, so it is split over m: the slices
  // store their tiles to the scratch and are added to y in order (launch_splitk_reduce) - reproducible bit for bit.
  // Without scratch one workgroup per output tile walks the whole reduction and adds to y in place.
  if (precision == DRIN_PREC_BF16X3 || precision == DRIN_PREC_BF16X3_ALL) {
    // pair-sized weight gradients: split-bf16 MFMA (gemm_tn_bf16x3.hip); mention-sized ones stay exact fp32
    if (gemm_tn_bf16x3_fits(lda, ldb, M, N, K, a, b) && gemm_tn_bf16x3_scratch_ok(y, ldy, N, K, scratch, scratch_floats))
      return launch_gemm_tn_bf16x3(a, lda, b, ldb, y, ldy, M, N, K, st, scratch, scratch_floats);
    precision = DRIN_PREC_F32;
  }
  DRIN_TRY(check_precision(precision, "gemm_tn"));
  if ((N % 4) || (K % 4)) {
    set_error("gemm_tn: N=%d and K=%d must be multiples of 4", N, K);
    return DRIN_E_SHAPE;
  }
  if (M > 0x7fffffff || M < 0) {
    set_error("gemm_tn: reduction length %lld outside [0, 2^31)", (long long)M);
    return DRIN_E_SHAPE;
  }
  if (M == 0) return DRIN_OK;   // an empty sum adds nothing (and no slice length is derived from it)
  const bool small = M <= 2048;   // mention-sized reductions (a few hundred rows): 64 x 64 tiles, 4x the workgroups
  int splits;
  if (small) {
    splits = small_tn_slices(M);
  } else {
    const int64_t tiles = cdiv(N, 128) * cdiv(K, 128);
    splits = (int)cdiv(1024, tiles);                      // aim at ~4 workgroups per CU
    const int max_splits = (int)cdiv(M, 4 * BK);          // at least 4 K-blocks per slice
    if (splits > max_splits) splits = max_splits;
  }
  const size_t tile_floats = (size_t)N * K;
  const bool can_slice = scratch != nullptr && aligned16(scratch) && aligned16(y) && (ldy % 4) == 0;
  if (!can_slice) splits = 1;
  while (splits > 1 && (size_t)splits * tile_floats > scratch_floats) --splits;
  if (splits < 1) splits = 1;
  float* c = splits > 1 ? scratch : y;
  const int64_t ldc = splits > 1 ? K : ldy;
  const int flags = splits > 1 ? GEMM_PARTIAL : GEMM_ACCUMULATE;
  int used = 1;
  if (small) {
    DRIN_TRY((launch<64, 64, true, true>(a, lda, b, ldb, nullptr, c, ldc, /*M=*/N, /*N=*/K, /*K=*/(int)M, splits, flags, st,
                                         "gemm_tn", (int64_t)tile_floats, &used)));
  } else {
    DRIN_TRY((launch<128, 128, true, true>(a, lda, b, ldb, nullptr, c, ldc, /*M=*/N, /*N=*/K, /*K=*/(int)M, splits, flags, st,
                                           "gemm_tn", (int64_t)tile_floats, &used)));
  }
  if (splits > 1) return launch_splitk_reduce(scratch, used, (int64_t)tile_floats, nullptr, y, ldy, N, K, true, st);
  return DRIN_OK;
}

}  // namespace drin
