// Split-bf16 GEMM on pre-split operands: y[m, n] = sum_k x[m, k] w[n, k] (+ bias[n]) where both operands
// arrive as two bf16 planes, v = hi + lo (the producers of the activations - k_entity_stream,
// k_pair_layer1 - and drin_prepare for the weights write the planes; see gemm_bf16x3.hip for the
// arithmetic and its error).  With nothing left to convert, the kernel is pure LDS-DMA + MFMA:
// every tile plane goes global -> LDS by global_load_lds_dwordx4 (no VGPR staging, no VALU, no
// ds_write), 16 bytes per lane, one 1 KiB piece (16 rows x 64 B) per wave-instruction.
//
// Tile 256 x 256 x 32, 512 threads = 8 waves (2 x 4, wave tile 128 x 64, 128 accumulator VGPRs), two
// LDS buffers of four planes (128 KiB).  LDS rows are 64 B; LDS-DMA writes linearly (wave base + lane *
// 16), so the XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free is applied to the
// SOURCE address: the lane that fills physical chunk c of row r fetches logical chunk c ^ ((r >> 2) & 3).
#include "device_utils.h"
#include "internal.h"

namespace drin {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace x3p {

#ifdef X3_STAMPS  // diagnostic build only (tools/x3_stamps.py)
__device__ unsigned long long g_stamps[8];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(i)                           \
  do {                                     \
    const unsigned long long _t = stamp(); \
    seg[i] += _t - tprev;                  \
    tprev = _t;                            \
  } while (0)
#else
#define STAMP(i)
#endif

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int THREADS = 512;
constexpr int PLANE_BYTES = 256 * 64;
constexpr int BUF_BYTES = 4 * PLANE_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;

__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with its own L2.
// The column tiles of one row tile all stream the same A rows, so they should run on ONE XCD at the same
// time: XCD x takes a contiguous range of the tile sequence (column index fastest).  Without this the
// A operand is fetched from HBM once per column tile (3x for N = 768).  Placement only affects speed.
__device__ __forceinline__ void tile_of_block(int& tile_x, int64_t& tile_y, int nx) {
  const unsigned nwg = gridDim.x * gridDim.y;
  const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned xcd = id & 7, k = id >> 3;
  const unsigned q = nwg >> 3, rem = nwg & 7;
  const unsigned t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  tile_x = (int)(t % (unsigned)nx);
  tile_y = (int64_t)(t / (unsigned)nx);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Wave w fills plane (w >> 1) of the tile, rows (w & 1) * 128 .. + 127: eight 16-row pieces.
struct Pieces {
  const char* src[8];  // per-lane source of each piece at k-block 0
};

__device__ __forceinline__ Pieces make_pieces(const __bf16* a_hi, const __bf16* a_lo, int64_t lda, int64_t m0, int64_t M,
                                              const __bf16* b_hi, const __bf16* b_lo, int64_t ldb, int64_t n0,
                                              int64_t N) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int plane = wave >> 1;
  const __bf16* base = plane == 0 ? a_hi : plane == 1 ? a_lo : plane == 2 ? b_hi : b_lo;
  const int64_t ld = plane < 2 ? lda : ldb, r0 = plane < 2 ? m0 : n0, lim = plane < 2 ? M : N;
  Pieces p;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (wave & 1) * 128 + i * 16 + (lane >> 2);  // row inside the tile
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);          // logical chunk this lane's 16 bytes hold
    int64_t g = r0 + row;
    g = g < lim ? g : lim - 1;                                // clamp: rows past the end are never stored
    p.src[i] = reinterpret_cast<const char*>(base + g * ld) + chunk * 16;
  }
  return p;
}

__device__ __forceinline__ void issue_piece(const Pieces& p, char* buf, int kb, int i) {
  const int wave = threadIdx.x >> 6;
  char* plane_base = buf + (wave >> 1) * PLANE_BYTES + (wave & 1) * 128 * 64;
  __builtin_amdgcn_global_load_lds((gptr_t)(p.src[i] + (int64_t)kb * (BK * 2)), (lptr_t)(plane_base + i * 1024), 16, 0, 0);
}
__device__ __forceinline__ void issue_tile(const Pieces& p, char* buf, int kb) {
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_piece(p, buf, kb, i);
}

__global__ void __launch_bounds__(THREADS, 2)
    k_gemm_x3_planes(const __bf16* __restrict__ a_hi, const __bf16* __restrict__ a_lo, int64_t lda,
                     const __bf16* __restrict__ b_hi, const __bf16* __restrict__ b_lo, int64_t ldb,
                     const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tx;
  int64_t ty;
  tile_of_block(tx, ty, (int)gridDim.x);
  const int64_t m0 = ty * BM;
  const int n0 = tx * BN;
  const int nkb = K / BK;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const Pieces pieces = make_pieces(a_hi, a_lo, lda, m0, M, b_hi, b_lo, ldb, n0, N);
  issue_tile(pieces, smem, 0);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and makes every wave's pieces visible

#ifdef X3_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = stamp();
#endif
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    const char* buf = smem + cur * BUF_BYTES;
    // The other buffer was last read before the barrier that ended iteration kb - 1: free to refill.
    // Its eight LDS-DMA pieces are issued ONE per group of six MFMAs, not all up front: 8 waves x 8 KiB
    // issued together back the load path up (64 B/clk per CU) and every wave stalls ~1000 cycles in the
    // issue queue with the matrix pipe idle; spread out they ride under the MFMAs.
    const bool more = kb + 1 < nkb;
    char* nbuf = smem + (cur ^ 1) * BUF_BYTES;
    STAMP(0);
    // Eight stages (k16 step s, row tile i) of six MFMAs.  The fragments of stage t + 1 are read from LDS
    // before the MFMAs of stage t are issued, so an LDS round trip (~200 cycles under load) hides behind
    // 192 cycles of matrix work instead of stalling the wave in front of every group.
    bf16x8 bh[2][2], bl[2][2], ah[2], al[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int off = swz(wn * 64 + j * 32 + r, h);
      bh[0][j] = *reinterpret_cast<const bf16x8*>(buf + 2 * PLANE_BYTES + off);
      bl[0][j] = *reinterpret_cast<const bf16x8*>(buf + 3 * PLANE_BYTES + off);
    }
    {
      const int off = swz(wm * 128 + r, h);
      ah[0] = *reinterpret_cast<const bf16x8*>(buf + off);
      al[0] = *reinterpret_cast<const bf16x8*>(buf + PLANE_BYTES + off);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int s = t >> 2, i = t & 3;
#ifndef P_NO_DMA
      if (more) issue_piece(pieces, nbuf, kb + 1, t);
#endif
#ifndef P_NO_LDSREAD
      if (t + 1 < 8) {
        const int s1 = (t + 1) >> 2, i1 = (t + 1) & 3;
        const int off = swz(wm * 128 + i1 * 32 + r, 2 * s1 + h);
        ah[(t + 1) & 1] = *reinterpret_cast<const bf16x8*>(buf + off);
        al[(t + 1) & 1] = *reinterpret_cast<const bf16x8*>(buf + PLANE_BYTES + off);
      }
      if (t == 2) {  // B fragments of the second k16 step, one stage ahead of their first use
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int off = swz(wn * 64 + j * 32 + r, 2 + h);
          bh[1][j] = *reinterpret_cast<const bf16x8*>(buf + 2 * PLANE_BYTES + off);
          bl[1][j] = *reinterpret_cast<const bf16x8*>(buf + 3 * PLANE_BYTES + off);
        }
      }
#else
      if (t == 0) { ah[1] = ah[0]; al[1] = al[0]; bh[1][0] = bh[0][0]; bh[1][1] = bh[0][1]; bl[1][0] = bl[0][0]; bl[1][1] = bl[0][1]; }
#endif
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t & 1], bh[s][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t & 1], bl[s][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t & 1], bh[s][j], acc[i][j], 0, 0, 0);
      }
      if (t == 3) STAMP(1);
    }
    STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(3);
#ifndef P_NO_BARRIER
    __syncthreads();  // next tile landed (vmcnt(0)) and this buffer is no longer read
#endif
    STAMP(4);
  }
#ifdef X3_STAMPS
  if (blockIdx.x == 0 && blockIdx.y == 1 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == (int)(K & 7))
    for (int i = 0; i < 8; ++i) g_stamps[i] = seg[i];
#endif

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + r;
      if (col >= N) continue;
      const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t row = m0 + wm * 128 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (row < M) C[row * ldc + col] = acc[i][j][v] + bv;
      }
    }
}

// fp32 -> (hi, lo) bf16 planes, 4 values per thread
__global__ void __launch_bounds__(256) k_split_planes(const float* __restrict__ x, __bf16* __restrict__ hi,
                                                      __bf16* __restrict__ lo, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = ld4(x + i * 4);
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 h, l;
  h[0] = (__bf16)v.x;
  h[1] = (__bf16)v.y;
  h[2] = (__bf16)v.z;
  h[3] = (__bf16)v.w;
  l[0] = (__bf16)(v.x - (float)h[0]);
  l[1] = (__bf16)(v.y - (float)h[1]);
  l[2] = (__bf16)(v.z - (float)h[2]);
  l[3] = (__bf16)(v.w - (float)h[3]);
  *reinterpret_cast<bf16x4*>(hi + i * 4) = h;
  *reinterpret_cast<bf16x4*>(lo + i * 4) = l;
}

}  // namespace x3p

#ifdef X3_STAMPS
extern "C" __attribute__((visibility("default"))) int drin_debug_x3p_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(x3p::g_stamps), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif

int launch_split_planes(const float* x, void* hi, void* lo, int64_t n, hipStream_t st) {
  if (n <= 0) return DRIN_OK;
  if (n % 4) {
    set_error("split_planes: element count %lld must be a multiple of 4", (long long)n);
    return DRIN_E_SHAPE;
  }
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(x3p::k_split_planes, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, st, x, (__bf16*)hi,
                     (__bf16*)lo, n / 4);
  DRIN_CHECK_LAUNCH("k_split_planes");
  return DRIN_OK;
}

int launch_gemm_x3_planes(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo,
                          int64_t ldb, const float* bias, float* y, int64_t ldy, int64_t M, int N, int K,
                          hipStream_t st) {
  if (M <= 0 || N <= 0) return DRIN_OK;
  if (K <= 0 || (K % x3p::BK) || (lda % 8) || (ldb % 8) || !aligned16(a_hi) || !aligned16(a_lo) || !aligned16(b_hi) ||
      !aligned16(b_lo)) {
    set_error("gemm_x3_planes: K=%d must be a multiple of 32, leading dimensions multiples of 8, planes 16-byte aligned", K);
    return DRIN_E_ALIGN;
  }
  const int64_t mt = cdiv(M, x3p::BM);
  if (mt > 65535) {
    set_error("gemm_x3_planes: %lld row tiles exceed the grid limit; split the batch", (long long)mt);
    return DRIN_E_SHAPE;
  }
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(x3p::k_gemm_x3_planes),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, x3p::LDS_BYTES);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm_x3_planes)");
    attr_done = true;
  }
  dim3 grid((unsigned)cdiv(N, x3p::BN), (unsigned)mt);
  KernelTimer timer(DRIN_KC_GEMM_PLANES, st);
  hipLaunchKernelGGL(x3p::k_gemm_x3_planes, grid, dim3(x3p::THREADS), x3p::LDS_BYTES, st, (const __bf16*)a_hi,
                     (const __bf16*)a_lo, lda, (const __bf16*)b_hi, (const __bf16*)b_lo, ldb, bias, y, ldy, M, N, K);
  DRIN_CHECK_LAUNCH("k_gemm_x3_planes");
  return DRIN_OK;
}

}  // namespace drin
