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
// next K-block's loads are in flight under the current block's 64 MFMAs per wave.
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

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_KC = BK + 4;    // k-contiguous row stride (floats)
constexpr int LDS_KM = BM + 4;    // k-major row stride (floats)
constexpr int TILE_FLOATS = BM * LDS_KC;  // 4608 >= 32 * 132 = 4224
static_assert(BK * LDS_KM <= TILE_FLOATS, "tile buffer too small");

struct Staged {
  float4 v[4];
};

// global -> registers.  rows: operand rows (m or n) of this tile start at row0, limit `rows`;
// k range of this block [k0, k0 + BK) clipped to k_end.
template <bool KMAJOR>
__device__ __forceinline__ void load_tile(Staged& s, const float* __restrict__ src, int64_t ld, int64_t row0,
                                          int64_t rows, int k0, int k_end) {
  const int t = threadIdx.x;
  if (!KMAJOR) {
    const int c4 = t & 7, r = t >> 3;
    const int k = k0 + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = row0 + r + 32 * i;
      s.v[i] = (row < rows && k < k_end) ? ld4(src + row * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    const int c4 = t & 31, kr = t >> 5;
    const int64_t row = row0 + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + kr + 8 * i;
      s.v[i] = (row < rows && k < k_end) ? ld4(src + (int64_t)k * ld + row) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void store_tile(const Staged& s, float* __restrict__ lds) {
  const int t = threadIdx.x;
  if (!KMAJOR) {
    const int c4 = t & 7, r = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) st4(lds + (r + 32 * i) * LDS_KC + c4 * 4, s.v[i]);
  } else {
    const int c4 = t & 31, kr = t >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) st4(lds + (kr + 8 * i) * LDS_KM + c4 * 4, s.v[i]);
  }
}

// fragment of 4 k-steps for the 32-row sub-tile starting at `row` (lane r = l & 31, half h = l >> 5)
template <bool KMAJOR>
__device__ __forceinline__ float4 read_frag(const float* __restrict__ lds, int row, int kg, int r, int h) {
  if (!KMAJOR) {
    return ld4(lds + (row + r) * LDS_KC + kg * 8 + h * 4);
  } else {
    const float* p = lds + (kg * 8 + h * 4) * LDS_KM + row + r;
    return make_float4(p[0], p[LDS_KM], p[2 * LDS_KM], p[3 * LDS_KM]);
  }
}

enum : int { GEMM_ACCUMULATE = 1, GEMM_ATOMIC = 2 };

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ void __launch_bounds__(256, 2)
    k_gemm_f32(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
               const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
               int k_per_split, int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // buffer c of operand A at smem + 2c tiles, of operand B at smem + (2c + 1) tiles

  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int k_begin = blockIdx.z * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  const int nkb = (k_end - k_begin + BK - 1) / BK;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  Staged sa, sb;
  if (nkb > 0) {
    load_tile<A_KMAJOR>(sa, A, lda, m0, M, k_begin, k_end);
    load_tile<B_KMAJOR>(sb, Bm, ldb, n0, N, k_begin, k_end);
    store_tile<A_KMAJOR>(sa, smem);
    store_tile<B_KMAJOR>(sb, smem + TILE_FLOATS);
  }
  __syncthreads();

  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    const bool more = kb + 1 < nkb;
    if (more) {  // next block's global loads fly under this block's MFMAs
      load_tile<A_KMAJOR>(sa, A, lda, m0, M, k_begin + (kb + 1) * BK, k_end);
      load_tile<B_KMAJOR>(sb, Bm, ldb, n0, N, k_begin + (kb + 1) * BK, k_end);
    }
    const float* as = smem + (2 * cur) * TILE_FLOATS;
    const float* bs = smem + (2 * cur + 1) * TILE_FLOATS;
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = read_frag<A_KMAJOR>(as, wm * 64 + i * 32, kg, r, h);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = read_frag<B_KMAJOR>(bs, wn * 64 + j * 32, kg, r, h);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (more) {
      store_tile<A_KMAJOR>(sa, smem + (2 * (cur ^ 1)) * TILE_FLOATS);
      store_tile<B_KMAJOR>(sb, smem + (2 * (cur ^ 1) + 1) * TILE_FLOATS);
    }
    __syncthreads();
  }

  // epilogue: C/D register v of a 32x32 tile is row (v & 3) + 8 (v >> 2) + 4 h, column r
  const bool first_split = blockIdx.z == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + r;
      if (col >= N) continue;
      const float bv = (bias != nullptr && first_split) ? bias[col] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t row = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (row >= M) continue;
        float* dst = C + row * ldc + col;
        const float val = acc[i][j][v] + bv;
        if (flags & GEMM_ATOMIC) {
          atomicAdd(dst, val);
        } else if (flags & GEMM_ACCUMULATE) {
          *dst += val;
        } else {
          *dst = val;
        }
      }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR>
static int launch(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc,
                  int64_t M, int N, int K, int splits, int flags, hipStream_t st, const char* what) {
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
  if (splits > 1) {
    kps = (int)(cdiv(cdiv(K, splits), BK) * BK);
    splits = (int)cdiv(K, kps);
  } else {
    splits = 1;
  }
  if (K <= 0) splits = 1;
  const size_t lds = sizeof(float) * 4 * TILE_FLOATS;
  static bool attr_done[4] = {false, false, false, false};
  const int idx = (A_KMAJOR ? 2 : 0) + (B_KMAJOR ? 1 : 0);
  auto kern = k_gemm_f32<A_KMAJOR, B_KMAJOR>;
  if (!attr_done[idx]) {  // > 64 KiB of dynamic LDS needs the opt-in; idempotent, so a race is harmless
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm)");
    attr_done[idx] = true;
  }
  dim3 grid((unsigned)cdiv(N, BN), (unsigned)mt, (unsigned)splits);
  KernelTimer timer(DRIN_KC_GEMM, st);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, A, lda, B, ldb, bias, C, ldc, M, N, K, kps, flags);
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

int launch_gemm_nt(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                   int64_t M, int N, int K, bool accumulate, int precision, hipStream_t st) {
  DRIN_TRY(check_precision(precision, "gemm_nt"));
  if (K % 4) {
    set_error("gemm_nt: K=%d must be a multiple of 4", K);
    return DRIN_E_SHAPE;
  }
  return launch<false, false>(x, ldx, w, ldw, bias, y, ldy, M, N, K, 1, accumulate ? GEMM_ACCUMULATE : 0, st,
                              "gemm_nt");
}

int launch_gemm_nn(const float* x, int64_t ldx, const float* w, int64_t ldw, float* y, int64_t ldy, int64_t M, int N,
                   int K, bool accumulate, int precision, hipStream_t st) {
  // y[m, n] = sum_k x[m, k] * w[k, n]: b(n, k) = w[k * ldw + n] is k-major
  DRIN_TRY(check_precision(precision, "gemm_nn"));
  if ((K % 4) || (N % 4)) {
    set_error("gemm_nn: K=%d and N=%d must be multiples of 4", K, N);
    return DRIN_E_SHAPE;
  }
  return launch<false, true>(x, ldx, w, ldw, nullptr, y, ldy, M, N, K, 1, accumulate ? GEMM_ACCUMULATE : 0, st,
                             "gemm_nn");
}

int launch_gemm_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int N,
                   int K, int precision, hipStream_t st) {
  // y[n, k] += sum_m a[m, n] * b[m, k]: both operands k-major over the reduction index m.
  // The reduction runs over all B*N pairs while the output is one weight matrix, so it is split over
  // m and the slices are combined with fp32 atomics (the caller's gradient buffer is the accumulator).
  DRIN_TRY(check_precision(precision, "gemm_tn"));
  if ((N % 4) || (K % 4)) {
    set_error("gemm_tn: N=%d and K=%d must be multiples of 4", N, K);
    return DRIN_E_SHAPE;
  }
  if (M > 0x7fffffff) {
    set_error("gemm_tn: reduction length %lld too large", (long long)M);
    return DRIN_E_SHAPE;
  }
  const int64_t tiles = cdiv(N, BM) * cdiv(K, BN);
  int splits = (int)cdiv(1024, tiles);                 // aim at ~4 workgroups per CU
  const int max_splits = (int)cdiv(M, 4 * BK);          // at least 4 K-blocks per slice
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return launch<true, true>(a, lda, b, ldb, nullptr, y, ldy, /*M=*/N, /*N=*/K, /*K=*/(int)M, splits, GEMM_ATOMIC, st,
                            "gemm_tn");
}

}  // namespace drin
