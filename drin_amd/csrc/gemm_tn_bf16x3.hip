// Weight-gradient contraction on the bf16 matrix cores, fp32-equivalent ("bf16x3", see gemm_bf16x3.hip):
//   y[n, k] += sum_m a[m, n] * b[m, k]          (dW = dY^T X: both operands are m-major, the reduction index
//                                                is the ROW of both matrices; M = pairs, y = one weight matrix)
//
// The bf16 MFMAs want 8 consecutive values of the reduction index per lane, but in memory consecutive m
// are a whole row apart.  Each thread therefore loads a 4 (m) x 4 (columns) fp32 block - one float4 from
// each of four consecutive rows - transposes it in registers, splits it into bf16 hi / lo and writes four
// 8-byte pieces (4 consecutive m of one column) to LDS.  LDS holds the tile as [column][32 m] rows of 64 B,
// exactly what the MFMA fragments read with ds_read_b128.
//
// LDS layout: the four columns of a thread's block (4 cg + j) would land 256 B apart - the bank period -
// so logical row p lives at physical row (p & 3) * ROWS/4 + (p >> 2): the pieces a wave writes with one
// instruction (j fixed, 8 neighbouring column groups x 8 m-groups) then form 512 contiguous bytes.  The
// 16-byte chunk index is XORed with (p & 3) so that the 16 rows of a fragment read (4 row residues x 4
// physical neighbours) fall on 16 distinct bank quads.
//
// The reduction is split over workgroups (M is 10^4 .. 10^5, the output has 9 .. 24 tiles of 256 x 256).  With
// caller scratch each slice stores its partial tile plainly and k_tn_reduce adds the slices to y in order
// (deterministic); without it the tiles are added to y with fp32 atomics like the exact-fp32 kernel of
// gemm_f32.hip - 65 536 atomics per workgroup cost ~50 us whatever the slice length (MI355X_MICROARCH.md: one
// 256-byte atomic wave-instruction per ~50 ns per CU), more than the MFMA work at M ~ 10^4.
#include "device_utils.h"
#include "internal.h"

namespace drin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace x3tn {

constexpr int BK = 32;      // reduction rows per LDS stage
constexpr int TILE = 256;   // output tile: 256 (n) x 256 (k)
constexpr int THREADS = 512;
constexpr int PLANE = TILE * 64;             // one bf16 plane of one operand: 256 rows x 64 B
constexpr int BUF_BYTES = 4 * PLANE;         // A hi, A lo, B hi, B lo
constexpr int LDS_BYTES = 2 * BUF_BYTES;     // double buffered: 128 KiB
// (Measured and not adopted, same box: 128 x 128 tiles with four waves of 64 x 64 and TWO workgroups per CU - a quarter of
//  the slices, half the partial-tile traffic: B = 64 step 0.845 against 0.835 ms of split-bf16 GEMM time, B = 512 4.90
//  against 4.47.)

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return ((((row & 3) * (TILE / 4)) + (row >> 2)) << 6) + ((chunk ^ (row & 3)) << 4);
}

// 32 (m) x 256 (columns) fp32 staged by 512 threads: lane = (m-group mg = lane >> 3, column group cgl = lane & 7),
// wave w covers column groups 8 w .. 8 w + 7; a column group is 4 adjacent columns.
// index != NULL: reduction row m of the operand is row index[m] of a TABLE (the entity tables of table-form training: the
// vertex-encoder inputs are gathered here instead of being materialised per step).  The indices of the NEXT stage are
// fetched one call ahead, so the dependent address -> data chain never sits inside a stage.
struct TransposeStager {
  const float* p;   // first row of the reduction range (or of the table), this thread's column
  int64_t ld;
  int mg, cg;
  float4 v[4];
  const int64_t* index;
  int64_t next_row[4];

  __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld_, int col0, int ncols,
                                       const int64_t* __restrict__ index_ = nullptr, int64_t m_first = 0, int64_t m_end = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    mg = lane >> 3;
    cg = wave * 8 + (lane & 7);
    int col = col0 + 4 * cg;
    col = col + 4 <= ncols ? col : ncols - 4;  // clamped columns feed output rows / columns that are never stored
    p = src + col;
    ld = ld_;
    index = index_;
    if (index != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = m_first + 4 * mg + i;
        next_row[i] = index[m < m_end ? m : m_end - 1];
      }
    }
  }
  // rows m0 + 4 mg + i; rows at or past m_end contribute zero (the loads stay unconditional: clamped row, then masked).
  // Calls walk m0 in steps of BK (the pipeline's stage order), which is what the index prefetch relies on.
  __device__ __forceinline__ void load(int64_t m0, int64_t m_end) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + 4 * mg + i;
      const bool live = m < m_end;
      const int64_t row = index != nullptr ? next_row[i] : (live ? m : m_end - 1);
      const float4 x = ld4(p + row * ld);
      v[i] = live ? x : make_float4(0.f, 0.f, 0.f, 0.f);
      if (index != nullptr) {
        const int64_t mn = m + BK;
        next_row[i] = index[mn < m_end ? mn : m_end - 1];
      }
    }
  }
  // the sum over m of this thread's four columns, of everything staged so far (rows past m_end were loaded as zeros)
  __device__ __forceinline__ void add_to(float4& s) const { s = s + ((v[0] + v[1]) + (v[2] + v[3])); }
  __device__ __forceinline__ void store(char* __restrict__ hi_plane, char* __restrict__ lo_plane) const {
    const float col[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x},
                             {v[0].y, v[1].y, v[2].y, v[3].y},
                             {v[0].z, v[1].z, v[2].z, v[3].z},
                             {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 hi, lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hi[i] = (__bf16)col[j][i];
        lo[i] = (__bf16)(col[j][i] - (float)hi[i]);
      }
      const int off = lds_off(4 * cg + j, mg >> 1) + ((mg & 1) << 3);
      *reinterpret_cast<bf16x4*>(hi_plane + off) = hi;
      *reinterpret_cast<bf16x4*>(lo_plane + off) = lo;
    }
  }
};

// One launch runs a GROUP of such products (TnGroup, internal.h: the pair-sized weight gradients of a backward pass).
// grid: 1-D; a problem's work items are (slice of the reduction, output tile n-tile major), tile fastest; problems in order.
struct Problem {
  const float* a;
  const float* b;
  float* y;
  float* partial;            // [slices][N][K] partial tiles, or NULL: fp32 atomics onto y
  const int64_t* b_index;
  int64_t lda, ldb, ldy, M, rows_per_slice;
  int N, K, k_tiles, tiles;
  unsigned first;            // first work item
  int slices;
  float* colsum;             // optional: column sums of a (the bias gradient that goes with dW = dY^T X) ...
  float* colsum_partial;     // ... as [slices][N] partial rows, or NULL: fp32 atomics onto colsum
};
struct GroupArgs {
  Problem p[TnGroup::MAX];
  int n;
};

__global__ void __launch_bounds__(THREADS, 1) k_gemm_tn_bf16x3(const GroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware order (as in gemm_x3_planes.hip): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  All
  // tiles of one slice read the same rows of both operands (fp32: a slice is megabytes), so XCD x takes a contiguous range
  // of the work-item sequence - a slice's rows then come through ONE L2 instead of up to eight.
  unsigned t = blockIdx.x;
  {
    const unsigned total = gridDim.x, xcd = t & 7, k = t >> 3, q = total >> 3, rem = total & 7;
    t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  int pi = 0;
  for (int i = 1; i < g.n; ++i) pi = t >= g.p[i].first ? i : pi;
  const Problem& P = g.p[pi];
  const float* __restrict__ A = P.a;
  const float* __restrict__ Bm = P.b;
  float* __restrict__ Y = P.y;
  float* __restrict__ partial = P.partial;
  const int64_t lda = P.lda, ldb = P.ldb, ldy = P.ldy, M = P.M, rows_per_slice = P.rows_per_slice;
  const int N = P.N, K = P.K, k_tiles = P.k_tiles, tiles = P.tiles;
  t -= P.first;
  const unsigned slice = t / (unsigned)tiles, tile = t - slice * (unsigned)tiles;
  const int n0 = (int)(tile / (unsigned)k_tiles) * TILE, k0 = (int)(tile % (unsigned)k_tiles) * TILE;
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  const int64_t m_end = m_begin + rows_per_slice < M ? m_begin + rows_per_slice : M;
  const int nkb = (int)((m_end - m_begin + BK - 1) / BK);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, wave tile 128 (n) x 64 (k)
  const int r = lane & 15, c = lane >> 4;
  constexpr int MI = 8, NI = 4;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[i][j][v] = 0.f;

  TransposeStager sa, sb;
  sa.init(A, lda, n0, N);
  sb.init(Bm, ldb, k0, K, P.b_index, m_begin, m_end);
  sa.load(m_begin, m_end);
  sb.load(m_begin, m_end);
  // the workgroups of the first k-tile also sum the columns of a (they stage every row of their n-tile exactly once)
  const bool sums = P.colsum != nullptr && k0 == 0;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sums) sa.add_to(csum);
  sa.store(smem, smem + PLANE);
  sb.store(smem + 2 * PLANE, smem + 3 * PLANE);
  if (nkb > 1) {
    sa.load(m_begin + BK, m_end);
    sb.load(m_begin + BK, m_end);
  }
  __syncthreads();

  bf16x8 bh[NI], bl[NI];
  auto load_b = [&](const char* buf) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int off = lds_off(wn * 64 + j * 16 + r, c);
      bh[j] = *reinterpret_cast<const bf16x8*>(buf + 2 * PLANE + off);
      bl[j] = *reinterpret_cast<const bf16x8*>(buf + 3 * PLANE + off);
    }
  };
  auto row_tiles = [&](const char* buf, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (i < i0 || i >= i1) continue;
      const int off = lds_off(wm * 128 + i * 16 + r, c);
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(buf + off);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(buf + PLANE + off);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        // the k-side fragment is the FIRST operand: a lane then holds four consecutive k of one n - one 16-byte store
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
      }
    }
  };

  // Same software pipeline as k_gemm_bf16x3: stage kb in LDS buffer kb & 1, stage kb + 1 in the staging
  // registers; between the two halves of the MFMA work the staged tile goes to the other buffer and the loads
  // of stage kb + 2 are issued.
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    const char* buf = smem + cur * BUF_BYTES;
    char* nb = smem + (cur ^ 1) * BUF_BYTES;
    const bool more = kb + 1 < nkb;
    load_b(buf);
    row_tiles(buf, 0, MI / 2);
    if (more) {
      if (sums) sa.add_to(csum);
      sa.store(nb, nb + PLANE);
      sb.store(nb + 2 * PLANE, nb + 3 * PLANE);
      if (kb + 2 < nkb) {
        sa.load(m_begin + (int64_t)(kb + 2) * BK, m_end);
        sb.load(m_begin + (int64_t)(kb + 2) * BK, m_end);
      }
    }
    row_tiles(buf, MI / 2, MI);
    __syncthreads();
  }

  if (sums) {  // the eight m-groups of a column group sit 8 lanes apart
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
      csum.x += __shfl_xor(csum.x, d);
      csum.y += __shfl_xor(csum.y, d);
      csum.z += __shfl_xor(csum.z, d);
      csum.w += __shfl_xor(csum.w, d);
    }
    const int col = n0 + 4 * sa.cg;
    if (sa.mg == 0 && col + 4 <= N) {
      if (P.colsum_partial != nullptr) {
        st4(P.colsum_partial + (int64_t)slice * N + col, csum);
      } else {
        unsafeAtomicAdd(P.colsum + col + 0, csum.x);
        unsafeAtomicAdd(P.colsum + col + 1, csum.y);
        unsafeAtomicAdd(P.colsum + col + 2, csum.z);
        unsafeAtomicAdd(P.colsum + col + 3, csum.w);
      }
    }
  }
  // C / D of a 16 x 16 tile (operands swapped above): n = lane & 15, k = 4 (lane >> 4) + v
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int nrow = n0 + wm * 128 + i * 16 + r;
    if (nrow >= N) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int kcol = k0 + wn * 64 + j * 16 + c * 4;
      if (kcol >= K) continue;  // K % 4 == 0: a group of four is inside or outside as a whole
      if (partial != nullptr) {
        st4(partial + ((int64_t)slice * N + nrow) * K + kcol, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) unsafeAtomicAdd(Y + (int64_t)nrow * ldy + kcol + v, acc[i][j][v]);
      }
    }
  }
}

// y[n, k] += sum_slices partial[slice][n][k], slices in order; every problem of the group in one launch
struct ReduceArgs {
  static constexpr int MAX = 2 * TnGroup::MAX;   // a product's tiles and, as a one-row matrix, its column sums
  const float* partial[MAX];
  float* y[MAX];
  int64_t ldy[MAX];
  int N[MAX], K4[MAX], slices[MAX];
  unsigned first[MAX];   // first block
  int n;
};
__global__ void __launch_bounds__(256) k_tn_reduce(const ReduceArgs g) {
  int pi = 0;
  for (int i = 1; i < g.n; ++i) pi = blockIdx.x >= g.first[i] ? i : pi;
  const int N = g.N[pi], K4 = g.K4[pi], slices = g.slices[pi];
  const int64_t i = (int64_t)(blockIdx.x - g.first[pi]) * 256 + threadIdx.x;
  if (i >= (int64_t)N * K4) return;
  const int n = (int)(i / K4), c4 = (int)(i - (int64_t)n * K4);
  const int64_t stride = (int64_t)N * K4 * 4;
  const float* p = g.partial[pi] + (int64_t)n * K4 * 4 + (int64_t)c4 * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  int z = 0;
  for (; z + 2 <= slices; z += 2) {  // two chains, fixed association
    s0 = s0 + ld4(p + z * stride);
    s1 = s1 + ld4(p + (z + 1) * stride);
  }
  if (z < slices) s0 = s0 + ld4(p + z * stride);
  float* dst = g.y[pi] + (int64_t)n * g.ldy[pi] + c4 * 4;
  st4(dst, ld4(dst) + (s0 + s1));
}

}  // namespace x3tn

bool gemm_tn_bf16x3_fits(int64_t lda, int64_t ldb, int64_t M, int N, int K, const void* a, const void* b) {
  return M >= 1024 && N >= 128 && K >= 128 && (N % 4) == 0 && (K % 4) == 0 && (lda % 4) == 0 && (ldb % 4) == 0 &&
         aligned16(a) && aligned16(b);
}

int TnGroup::add(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int N, int K,
                 const int64_t* b_index, float* colsum) {
  if (y == nullptr || M <= 0 || N <= 0 || K <= 0) return DRIN_OK;
  if (!gemm_tn_bf16x3_fits(lda, ldb, M, N, K, a, b)) {
    set_error("gemm_tn_bf16x3: shape M=%lld N=%d K=%d / alignment outside the kernel's contract", (long long)M, N, K);
    return DRIN_E_SHAPE;
  }
  if (n == MAX) {
    set_error("internal: more than %d weight-gradient products in one group", MAX);
    return DRIN_E_SHAPE;
  }
  item[n++] = {a, lda, b, ldb, y, ldy, M, N, K, b_index, colsum};
  return DRIN_OK;
}

int launch_gemm_tn_group(const TnGroup& grp, hipStream_t st, float* scratch, size_t scratch_floats) {
  if (grp.n == 0) return DRIN_OK;
  // One workgroup per CU in all (the 128 KiB tile buffers allow one per CU at a time): every extra slice adds a pipeline
  // fill and a 256 KiB partial tile (or 65 536 fp32 atomics), which at M ~ 10^4 cost more than the MFMA work itself
  // (measured at M = 12 928: 86 slices 162 us, 28 slices 92 us).  So the chip's 256 workgroups are dealt over ALL the
  // products of the group in proportion to their work: one target slice length, a whole number of 32-row stages,
  // grown until the group fits.  (Five products of one B = 64 backward pass: slices of ~62 stages instead of 15, same box
  // 0.835 -> 0.74 ms of split-bf16 GEMM time per step; at B = 512, 58 .. 160 stages per slice already, neither better nor worse.)
  int tiles[TnGroup::MAX];
  int64_t work = 0, longest = 0;
  for (int i = 0; i < grp.n; ++i) {
    const auto& it = grp.item[i];
    tiles[i] = (int)(cdiv(it.N, x3tn::TILE) * cdiv(it.K, x3tn::TILE));
    work += (int64_t)tiles[i] * it.M;
    longest = it.M > longest ? it.M : longest;
  }
  int64_t target = cdiv(cdiv(work, 256), x3tn::BK) * x3tn::BK;
  if (target < 4 * x3tn::BK) target = 4 * x3tn::BK;
  for (;; target += x3tn::BK) {
    int64_t wgs = 0;
    for (int i = 0; i < grp.n; ++i) wgs += tiles[i] * cdiv(grp.item[i].M, target);
    if (wgs <= 256 || target >= longest) break;
  }
  x3tn::GroupArgs ga;
  x3tn::ReduceArgs ra;
  ga.n = grp.n;
  ra.n = 0;
  int64_t items = 0, blocks = 0;
  size_t part = 0;
  bool two_stage = scratch != nullptr && aligned16(scratch);
  for (int i = 0; i < grp.n; ++i) {
    const auto& it = grp.item[i];
    int64_t slices = cdiv(it.M, target);
    const int64_t rows = cdiv(cdiv(it.M, slices), x3tn::BK) * x3tn::BK;   // equal slices of this product
    slices = cdiv(it.M, rows);
    auto& P = ga.p[i];
    P.a = it.a, P.b = it.b, P.y = it.y, P.b_index = it.b_index;
    P.lda = it.lda, P.ldb = it.ldb, P.ldy = it.ldy, P.M = it.M, P.rows_per_slice = rows;
    P.N = it.N, P.K = it.K, P.k_tiles = (int)cdiv(it.K, x3tn::TILE), P.tiles = tiles[i];
    P.first = (unsigned)items, P.slices = (int)slices;
    P.partial = scratch != nullptr ? scratch + part : nullptr;
    items += slices * tiles[i];
    part += (size_t)slices * it.N * it.K;
    two_stage = two_stage && (it.ldy % 4) == 0 && aligned16(it.y);
    two_stage = two_stage && (it.colsum == nullptr || aligned16(it.colsum));
    auto reduce_entry = [&](const float* partial, float* y, int64_t ldy, int n_rows, int k4) {
      const int j = ra.n++;
      ra.partial[j] = partial, ra.y[j] = y, ra.ldy[j] = ldy, ra.N[j] = n_rows, ra.K4[j] = k4, ra.slices[j] = (int)slices;
      ra.first[j] = (unsigned)blocks;
      blocks += cdiv((int64_t)n_rows * k4, 256);
    };
    reduce_entry(P.partial, it.y, it.ldy, it.N, it.K / 4);
    P.colsum = it.colsum;
    P.colsum_partial = nullptr;
    if (it.colsum != nullptr) {
      P.colsum_partial = scratch != nullptr ? scratch + part : nullptr;
      reduce_entry(P.colsum_partial, it.colsum, it.N, 1, it.N / 4);
      part += (size_t)slices * it.N;
    }
  }
  two_stage = two_stage && part <= scratch_floats;
  if (!two_stage)
    for (int i = 0; i < grp.n; ++i) ga.p[i].partial = ga.p[i].colsum_partial = nullptr;
  if (items > (int64_t)1 << 30) {
    set_error("gemm_tn_bf16x3: %lld work items exceed the grid limit", (long long)items);
    return DRIN_E_SHAPE;
  }
  {
    static DynLdsOptIn opt_in;
    DRIN_TRY(ensure_dynamic_lds(opt_in, reinterpret_cast<const void*>(x3tn::k_gemm_tn_bf16x3), x3tn::LDS_BYTES,
                                "hipFuncSetAttribute(gemm_tn_bf16x3)"));
  }
  KernelTimer timer(DRIN_KC_GEMM_X3, st);
  hipLaunchKernelGGL(x3tn::k_gemm_tn_bf16x3, dim3((unsigned)items), dim3(x3tn::THREADS), x3tn::LDS_BYTES, st, ga);
  DRIN_CHECK_LAUNCH("k_gemm_tn_bf16x3");
  if (two_stage) {
    hipLaunchKernelGGL(x3tn::k_tn_reduce, dim3((unsigned)blocks), dim3(256), 0, st, ra);
    DRIN_CHECK_LAUNCH("k_tn_reduce");
  }
  return DRIN_OK;
}

int launch_gemm_tn_bf16x3(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M,
                          int N, int K, hipStream_t st, float* scratch, size_t scratch_floats, const int64_t* b_index) {
  if (M <= 0 || N <= 0 || K <= 0) return DRIN_OK;
  TnGroup one;
  DRIN_TRY(one.add(a, lda, b, ldb, y, ldy, M, N, K, b_index));
  return launch_gemm_tn_group(one, st, scratch, scratch_floats);
}

}  // namespace drin
