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
// SOURCE address: the lane that fills physical chunk c of row r fetches logical chunk c ^ f(r).
#include <stdlib.h>

#include "device_utils.h"
#include "internal.h"

namespace drin {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace x3p {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int THREADS = 512;
constexpr int PLANE_BYTES = 256 * 64;
constexpr int BUF_BYTES = 4 * PLANE_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with its own L2.
// The column tiles of one row tile all stream the same A rows, so they should run on ONE XCD at the same
// time: XCD x takes a contiguous range of the tile sequence (column index fastest).  Without this the
// A operand is fetched from HBM once per column tile (3x for N = 768).  Placement only affects speed.
// Tail split (as in gemm_bf16x3.hip): the grid is 1-D; ids below `full` are whole tiles in the XCD order above, the
// `tail_tiles` of a partly filled last round follow as `ksplit` work items each, one per slice of K.
__device__ __forceinline__ void tile_of_block(int& tile_x, int64_t& tile_y, int nx, unsigned full, int ksplit, int& kpart,
                                              unsigned& tail_slot) {
  const unsigned id = blockIdx.x;
  unsigned t;
  kpart = 0;
  tail_slot = 0;
  if (id < full) {
    const unsigned xcd = id & 7, k = id >> 3;
    const unsigned q = full >> 3, rem = full & 7;
    t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  } else {
    const unsigned u = id - full;
    t = full + u / (unsigned)ksplit;
    kpart = (int)(u % (unsigned)ksplit);
    tail_slot = (u / (unsigned)ksplit) * (unsigned)(ksplit - 1) + (unsigned)(kpart - 1);
  }
  tile_x = (int)(t % (unsigned)nx);
  tile_y = (int64_t)(t / (unsigned)nx);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Wave w fills plane (w >> 1) of the tile, rows (w & 1) * 128 .. + 127: eight 16-row pieces.  The per-lane source of
// piece i is recomputed when it is issued (a handful of integer ops) instead of living in 16 VGPRs.
struct Pieces {
  const char* base;   // plane base + this lane's logical 16-byte chunk
  int64_t ld_bytes;   // row stride of the plane in bytes
  int64_t row0, lim;  // first row of this lane's piece 0, number of valid rows
};

template <bool A_LO, bool B_LO>
__device__ __forceinline__ void issue_piece(const Pieces& p, char* buf, int kb, int i) {
  const int wave = threadIdx.x >> 6;
  if (!A_LO && (wave >> 1) == 1) return;
  if (!B_LO && (wave >> 1) == 3) return;
  char* plane_base = buf + (wave >> 1) * PLANE_BYTES + (wave & 1) * 128 * 64;
  int64_t g = p.row0 + 16 * i;
  g = g < p.lim ? g : p.lim - 1;  // rows past the end re-read the last row: their products are never stored
  __builtin_amdgcn_global_load_lds((gptr_t)(p.base + g * p.ld_bytes + (int64_t)kb * (BK * 2)), (lptr_t)(plane_base + i * 1024),
                                   16, 0, 0);
}
template <bool A_LO, bool B_LO>
__device__ __forceinline__ void issue_tile(const Pieces& p, char* buf, int kb) {
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_piece<A_LO, B_LO>(p, buf, kb, i);
}

// MFMA shape: v_mfma_f32_16x16x32_bf16.  The wave tile 128 x 64 is 8 x 4 tiles of 16 x 16 and a whole K-block
// (32) is ONE MFMA k-step.  Measured against the same kernel on 32x32x16 (3 x 2 x 16 MFMAs per K-block, same
// cycles per FLOP): +4.5 ... +6 % - the chip holds a higher clock on this shape (MI355X_MICROARCH.md "DVFS
// give-back" item 7).  A lane's fragment is row (lane & 15), 16-byte chunk (lane >> 4): the four chunks of
// a row are read by four different lane quarters, and the swizzle that keeps every ds_read_b128 lane group
// on 16 distinct bank quads is the row-quad permutation f = (0, 2, 3, 1).
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz16(int row, int c) {
  const int f = (0x78 >> (((row >> 2) & 3) << 1)) & 3;
  return row * 64 + ((c ^ f) << 4);
}

__device__ __forceinline__ Pieces make_pieces(const __bf16* a_hi, const __bf16* a_lo, int64_t lda, int64_t m0,
                                                int64_t M, const __bf16* b_hi, const __bf16* b_lo, int64_t ldb,
                                                int64_t n0, int64_t N) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int plane = wave >> 1;
  const __bf16* base = plane == 0 ? a_hi : plane == 1 ? a_lo : plane == 2 ? b_hi : b_lo;
  const int64_t ld = plane < 2 ? lda : ldb, r0 = plane < 2 ? m0 : n0, lim = plane < 2 ? M : N;
  // the swizzle of row r is f((r >> 2) & 3); pieces are 16 rows apart, so it is the same for all eight: f(lane >> 4)
  const int f = (0x78 >> (((lane >> 4) & 3) << 1)) & 3;
  const int chunk = (lane & 3) ^ f;  // logical chunk whose bytes land in physical chunk lane & 3
  Pieces p;
  p.base = reinterpret_cast<const char*>(base) + chunk * 16;
  p.ld_bytes = ld * 2;
  p.row0 = r0 + (wave & 1) * 128 + (lane >> 2);
  p.lim = lim;
  return p;
}

// (Tried: weight fragments fetched straight into registers by global_load_dwordx4 - a lane's fragment of the
// K-contiguous planes is 16 contiguous bytes - so that the weights never touch LDS: 291 vs 346 TF/s on the same box.
// The K-loop is not LDS-bound; the extra vector-memory traffic costs more than the LDS traffic it removes.
// Cycle stamps (-DDRIN_STAMPS, tools/stamps_probe.py) of one tile at K = 768: prologue 6 k cycles, K-loop 109 k (4 400
// per K-block against 3 072 of pure MFMA issue: ~300 of fragment reads before the first MFMA, ~700 of barrier skew),
// epilogue 24 k.  Tried against that: a mid-block barrier that publishes the next block early so that its first
// fragments are prefetched across the block boundary (-8 %: the second barrier costs more than the bubble), a
// staggered start of the first round of workgroups to de-synchronise the epilogue bursts (-5 %).
// Also tried: persistent workgroups (one per CU walking a tile list, the next tile's first K-block requested before
// the epilogue of the finished one): 1411 vs 1365 us at 413 696 x 768 x 768 - the hardware's dynamic dispatch of
// 4 848 independent tiles balances better than a static list, and the fill it would hide is small.)
#ifdef DRIN_STAMPS
__device__ unsigned long long g_stamps[8 * 64 * 4 + 64];  // [wave][kb][4]: loop top, first MFMA issued, last MFMA issued, after barrier
#endif
template <bool A_LO, bool B_LO>
__global__ void __launch_bounds__(THREADS, 2)
    k_gemm_x3_planes(const __bf16* __restrict__ a_hi, const __bf16* __restrict__ a_lo, int64_t lda,
                       const __bf16* __restrict__ b_hi, const __bf16* __restrict__ b_lo, int64_t ldb,
                       const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                       int64_t part_stride, int nx, unsigned full, int ksplit, float* __restrict__ tail) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (gridDim.z > 1) {  // split K (small problems): slice blockIdx.z of the reduction, raw partial tile to scratch
    K /= (int)gridDim.z;
    const int64_t k0 = (int64_t)blockIdx.z * K;
    a_hi += k0;
    if (A_LO) a_lo += k0;
    b_hi += k0;
    if (B_LO) b_lo += k0;
    C += (int64_t)blockIdx.z * part_stride;
    ldc = N;
    bias = nullptr;
  }
#ifdef DRIN_STAMPS
  const unsigned long long t_start = __builtin_readcyclecounter();
#endif
  int tx, kpart;
  int64_t ty;
  unsigned tail_slot;
  tile_of_block(tx, ty, nx, full, ksplit, kpart, tail_slot);
  const int64_t m0 = ty * BM;
  const int n0 = tx * BN;
  if (blockIdx.x >= full) {  // a K-slice of a tail tile
    K /= ksplit;
    const int64_t k0 = (int64_t)kpart * K;
    a_hi += k0;
    if (A_LO) a_lo += k0;
    b_hi += k0;
    if (B_LO) b_lo += k0;
  }
  const int nkb = K / BK;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, c = lane >> 4;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[i][j][v] = 0.f;

  const Pieces pieces = make_pieces(a_hi, A_LO ? a_lo : a_hi, lda, m0, M, b_hi, B_LO ? b_lo : b_hi, ldb, n0, N);
  issue_tile<A_LO, B_LO>(pieces, smem, 0);
  __syncthreads();
#ifdef DRIN_STAMPS
  const unsigned long long t_prologue = __builtin_readcyclecounter();
#endif

  for (int kb = 0; kb < nkb; ++kb) {
#ifdef DRIN_STAMPS
    const bool stamp = blockIdx.x == 121 && lane == 0 && kb < 64;
    if (stamp) g_stamps[(wave * 64 + kb) * 4 + 0] = __builtin_readcyclecounter();
#endif
    const int cur = kb & 1;
    const char* buf = smem + cur * BUF_BYTES;
    const bool more = kb + 1 < nkb;
    char* nbuf = smem + (cur ^ 1) * BUF_BYTES;
    bf16x8 bh[4], bl[4], ah[2], al[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int off = swz16(wn * 64 + j * 16 + r, c);
      bh[j] = *reinterpret_cast<const bf16x8*>(buf + 2 * PLANE_BYTES + off);
      if (B_LO) bl[j] = *reinterpret_cast<const bf16x8*>(buf + 3 * PLANE_BYTES + off);
    }
    {
      const int off = swz16(wm * 128 + r, c);
      ah[0] = *reinterpret_cast<const bf16x8*>(buf + off);
      if (A_LO) al[0] = *reinterpret_cast<const bf16x8*>(buf + PLANE_BYTES + off);
    }
#ifdef DRIN_STAMPS
    if (stamp) g_stamps[(wave * 64 + kb) * 4 + 1] = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // eight row tiles; the next one's fragments are read one stage ahead
      if (more) issue_piece<A_LO, B_LO>(pieces, nbuf, kb + 1, t);
      if (t + 1 < 8) {
        const int off = swz16(wm * 128 + (t + 1) * 16 + r, c);
        ah[(t + 1) & 1] = *reinterpret_cast<const bf16x8*>(buf + off);
        if (A_LO) al[(t + 1) & 1] = *reinterpret_cast<const bf16x8*>(buf + PLANE_BYTES + off);
      }
      // term-major: consecutive MFMAs write different accumulators, a dependent one is four issues away (+1 %)
      if (A_LO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[t & 1], acc[t][j], 0, 0, 0);
      }
      if (B_LO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[t & 1], acc[t][j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[t & 1], acc[t][j], 0, 0, 0);
    }
#ifdef DRIN_STAMPS
    if (stamp) g_stamps[(wave * 64 + kb) * 4 + 2] = __builtin_readcyclecounter();
#endif
    __syncthreads();
#ifdef DRIN_STAMPS
    if (stamp) g_stamps[(wave * 64 + kb) * 4 + 3] = __builtin_readcyclecounter();
#endif
  }

  // The MFMAs take the WEIGHT fragment as their first operand, so a lane's four accumulator registers of a 16 x 16
  // tile are four consecutive output COLUMNS of one row (row lane & 15, columns 4 (lane >> 4) + v): one 16-byte store
  // per tile instead of four scattered 4-byte ones (32 instead of 128 store instructions per wave; the epilogue was
  // 24 % of a tile's time at K = 768).
  if (kpart > 0) {  // raw accumulators of a tail slice: [slot][256][256], folded into C by the tail-add launch
    float* part = tail + (size_t)tail_slot * (BM * BN);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        st4(part + (wm * 128 + i * 16 + r) * BN + wn * 64 + j * 16 + c * 4,
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
    return;
  }
  const bool vec_ok = (ldc % 4) == 0 && (reinterpret_cast<uintptr_t>(C) & 15u) == 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t row = m0 + wm * 128 + i * 16 + r;
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + c * 4;
      float* dst = C + row * ldc + col;
      if (vec_ok && col + 3 < N) {
        float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (bias != nullptr) o = o + ld4(bias + col);
        st4(dst, o);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (col + v < N) dst[v] = acc[i][j][v] + (bias != nullptr ? bias[col + v] : 0.f);
      }
    }
  }
#ifdef DRIN_STAMPS
  if (blockIdx.x == 121 && (threadIdx.x & 63) == 0) {
    __builtin_amdgcn_s_waitcnt(0);  // stores issued and acknowledged
    const int wv = threadIdx.x >> 6;
    unsigned long long* o = g_stamps + 8 * 64 * 4 + wv * 4;
    o[0] = t_start;
    o[1] = t_prologue;
    o[2] = g_stamps[(wv * 64 + (K / BK - 1 < 63 ? K / BK - 1 : 63)) * 4 + 3];
    o[3] = __builtin_readcyclecounter();
  }
#endif
}


// ------------------------------------------------------------------------------------------------
// The same product on a finer pipeline ("four phases per K-block", after the 8-phase bf16 template of
// cdna_hip_programming.md 5).  What it changes against k_gemm_x3_planes:
//  * the K-block's 64 KiB arrive as FOUR units of 16 KiB - (operand, 128-row half), both planes - one unit issued per
//    phase, and the LDS-DMA stays in flight ACROSS the barriers: a counted s_waitcnt vmcnt(4) (two units outstanding) and
//    raw s_barrier instead of __syncthreads(), whose vmcnt(0) drains every DMA at every K-block;
//  * a wave's 128 x 64 output is two 64-row strips x two 32-column strips, one from each half, so that output quadrant
//    (qi, qj) needs exactly unit A[qi] and unit B[qj]; the quadrants run 00, 01, 11, 10 - each shares an operand with its
//    predecessor, whose fragments stay in registers (12 + 4 + 8 + 4 fragment reads per K-block);
//  * a phase is { fragment reads, DMA issue, counted wait | barrier | 24 MFMAs | barrier } and the two halves of the
//    workgroup (waves 0 - 3, 4 - 7: one wave of each per SIMD) run ONE BARRIER APART, so that on every SIMD one wave
//    issues MFMAs while the other reads LDS - no second fragment set, 176 accumulator + fragment registers.
//   unit      issued in phase   first read in phase        (g = 4 t + q counts phases; the unit belongs to K-block t + 1)
//   A0(t+1)   4 t               4 t + 4
//   B0(t+1)   4 t + 1           4 t + 4 (and 4 t + 7)
//   B1(t+1)   4 t + 2           4 t + 5
//   A1(t+1)   4 t + 3           4 t + 6
// Before the first barrier of phase g every wave has waited for all but its two newest units (vmcnt(4): two DMA
// instructions per unit per wave), which retires what phase g + 1 reads first; the delayed half waits one barrier later, and
// the second barrier of the phase still lies between its wait and anybody's read.  A slot is re-filled two or more phases
// after its last read, whose lgkmcnt(0) precedes the MFMAs of that phase.
// The fp16 form's two scales (row, weight plane) are powers of two: their EXPONENTS are added and applied with one ldexp per
// element, so that an extreme row scale (2^126) next to a weight scale other than 1 never overflows or flushes an intermediate
// product `row_scale * b_scale` that the result itself does not need.
__device__ __forceinline__ int pow2_exponent(float s) { return (int)((__builtin_bit_cast(uint32_t, s) >> 23) & 0xffu) - 127; }
__device__ __forceinline__ float4 scale_pow2(float4 v, int e) { return make_float4(ldexpf(v.x, e), ldexpf(v.y, e), ldexpf(v.z, e), ldexpf(v.w, e)); }

namespace p4 {

#ifndef DRIN_P4_TERM_MAJOR
#define DRIN_P4_TERM_MAJOR 1
#endif
constexpr bool kTermMajor = DRIN_P4_TERM_MAJOR != 0;
#ifndef DRIN_P4_DMA_PLACE
#define DRIN_P4_DMA_PLACE 0
#endif
constexpr int kDmaPlace = DRIN_P4_DMA_PLACE;
#ifndef DRIN_P4_REBALANCE
#define DRIN_P4_REBALANCE 1
#endif
constexpr bool kRebalance = DRIN_P4_REBALANCE != 0 && kDmaPlace == 0;   // the fp32-A kernel's phase schedule (see there)
// fp32-A kernel, rebalanced schedule: the split of a landed activation unit (24 vector instructions + 4 LDS writes per thread) runs
// INSIDE the MFMA half of its phase - one vector instruction per MFMA gap (an MFMA holds the SIMD's vector issue for 8 of its 16
// cycles: MI355X_MICROARCH.md) - instead of between the counted wait and the phase's first barrier, where its ~250 cycles
// delayed all eight waves twice per K-block.
#ifndef DRIN_P4_SPLIT_IN_MMA
#define DRIN_P4_SPLIT_IN_MMA 1
#endif
constexpr bool kSplitInMma = DRIN_P4_SPLIT_IN_MMA != 0 && kRebalance;
// the counted wait before a phase's first barrier (two DMA / load instructions per unit per wave)
__device__ __forceinline__ void wait_units() {
  if (kDmaPlace == 0) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
}
constexpr int UNIT_BYTES = 2 * 128 * 64;       // two planes x 128 rows x 64 B
constexpr int BUF = 4 * UNIT_BYTES;            // A0, A1, B0, B1
constexpr int LDS = 2 * BUF;

struct Src {
  const char* ptr[4][2];   // [unit][piece]: this lane's 16 source bytes of K-block 0 (rows past the end clamped)
};

// unit u: 0 = A0, 1 = A1, 2 = B0, 3 = B1.  Wave w fills pieces 2 w and 2 w + 1 of the unit's 16 (plane w >> 2, rows 32 (w & 3) ..)
// (KB_BYTES: bytes a K-block advances along a plane row - 64, or 128 in the single-plane fp16 kernel, whose K-blocks are 64 wide)
template <int U, int KB_BYTES = BK * 2>
__device__ __forceinline__ void issue_unit(const Src& s, char* buf, int kb) {
  const int wave = threadIdx.x >> 6;
  char* dst = buf + U * UNIT_BYTES + (wave >> 2) * (128 * 64) + (wave & 3) * 2048;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(s.ptr[U][i] + (int64_t)kb * KB_BYTES), (lptr_t)(dst + i * 1024), 16, 0, 0);
}

}  // namespace p4

// A_LO = false: the activation operand is exact in bf16 (features stored as bf16: the image rows read in place) - the hi x lo_a
// term does not exist, 16 MFMAs per phase instead of 24.  The A units keep their shape and their DMA count (the wave group that
// would fetch the lo plane fetches the hi plane once more, into the slot nobody reads), so that every counted wait is unchanged.
// F16 (DRIN_PREC_BF16X3_IF16: the entity-image contraction in ONE pass of v_mfma_f32_16x16x32_f16): each operand is a SINGLE fp16
// plane and a K-block is 64 wide - its two 32-wide sub-blocks take the unit slots the split product gives to the hi and the lo
// plane (a_lo = a_hi + 32, b_lo = b_hi + 32 elements, set here), so units, DMA counts and counted waits are again unchanged; a
// 16 x 16 tile takes two MFMAs per K-block (sub-block 0, sub-block 1: the k order of 32-wide blocks), 16 per phase, for half
// the K-blocks.  Row m of A was divided by row_scale[m] and B by *b_scale when their planes were written (powers of two: exact;
// fp16's range never matters): the epilogue multiplies output row m by row_scale[m] * *b_scale.
// -DDRIN_P4_STAMPS (tools/probes/p4_stamps_probe.py): cycle stamps of one workgroup (block 121 of a launch of more than 100 000 rows) of
// the two four-phase kernels - per K-block and phase: phase start, after the first barrier (MFMAs may start), after the MFMAs - for one
// wave of each wave group.  Stamps sit only where the wave has no LDS read in flight anyway (s_memtime counts in lgkmcnt).
#ifdef DRIN_P4_STAMPS
__device__ unsigned long long g_p4_stamps[2 * 2 * 32 * 4 * 3];   // [kernel: 0 planes, 1 fp32-A][wave group][K-block][phase][point]
#define DRIN_P4_STAMP(KERNEL, KB, PHASE, POINT)                                                                               \
  if (stamp_on && (KB) < 32)                                                                                                  \
    g_p4_stamps[((((KERNEL) * 2 + (int)(threadIdx.x >> 8)) * 32 + (KB)) * 4 + (PHASE)) * 3 + (POINT)] = __builtin_readcyclecounter()
#else
#define DRIN_P4_STAMP(KERNEL, KB, PHASE, POINT)
#endif
template <bool A_LO = true, bool F16 = false>
__global__ void __launch_bounds__(THREADS, 2)
    k_gemm_x3_planes_p4(const __bf16* __restrict__ a_hi, const __bf16* __restrict__ a_lo, int64_t lda,
                        const __bf16* __restrict__ b_hi, const __bf16* __restrict__ b_lo, int64_t ldb,
                        const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, int nx,
                        unsigned full, int ksplit, float* __restrict__ tail, const float* __restrict__ row_scale,
                        const float* __restrict__ b_scale) {
  static_assert(!F16 || A_LO, "the fp16 form uses both slots of every unit");
  constexpr int KSTEP = F16 ? 2 * BK : BK;       // elements of K per K-block
  constexpr int KB_BYTES = KSTEP * 2;
  if (F16) {
    a_lo = a_hi + BK;
    b_lo = b_hi + BK;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tx, kpart;
  int64_t ty;
  unsigned tail_slot;
  tile_of_block(tx, ty, nx, full, ksplit, kpart, tail_slot);   // ids >= full: K-slices of the tail tiles (as in k_gemm_x3_planes)
  const int64_t m0 = ty * BM;
  const int n0 = tx * BN;
  int64_t k0 = 0;
  if (blockIdx.x >= full) {
    K /= ksplit;
    k0 = (int64_t)kpart * K;
  }
  const int nkb = K / KSTEP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, c = lane >> 4;

  p4::Src src;
  {
    const int f = (0x78 >> (((lane >> 4) & 3) << 1)) & 3;
    const int chunk = (lane & 3) ^ f;
    const bool lo = wave >= 4;
    const char* ap = reinterpret_cast<const char*>((lo ? a_lo : a_hi) + k0) + chunk * 16;
    const char* bp = reinterpret_cast<const char*>((lo ? b_lo : b_hi) + k0) + chunk * 16;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool is_b = u >= 2;
        int64_t g = (is_b ? (int64_t)n0 : m0) + (u & 1) * 128 + (wave & 3) * 32 + 16 * i + (lane >> 2);
        const int64_t lim = is_b ? (int64_t)N : M;
        g = g < lim ? g : lim - 1;
        src.ptr[u][i] = (is_b ? bp : ap) + g * (is_b ? ldb : lda) * 2;
      }
  }

  f32x4 acc[2][2][4][2];   // [row half][column half][row tile][column tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[a][b][i][j][v] = 0.f;

  bf16x8 ah[4], al[4], bh[2], bl[2];
  auto read_a = [&](const char* buf, int half) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* p = buf + half * p4::UNIT_BYTES + swz16(wm * 64 + i * 16 + r, c);
      ah[i] = *reinterpret_cast<const bf16x8*>(p);
      if (A_LO) al[i] = *reinterpret_cast<const bf16x8*>(p + 128 * 64);
    }
  };
  auto read_b = [&](const char* buf, int half) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* p = buf + (2 + half) * p4::UNIT_BYTES + swz16(wn * 32 + j * 16 + r, c);
      bh[j] = *reinterpret_cast<const bf16x8*>(p);
      bl[j] = *reinterpret_cast<const bf16x8*>(p + 128 * 64);
    }
  };
  auto mma = [&](f32x4 (&cc)[4][2], auto&& mid) {
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F16) {   // (ah, bh): k sub-block 0 of the 64-wide block, (al, bl): sub-block 1 - fp16 bit patterns in the planes
      typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh[j]), __builtin_bit_cast(f16x8, ah[i]), cc[i][j], 0, 0, 0);
      mid();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bl[j]), __builtin_bit_cast(f16x8, al[i]), cc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      return;
    }
    // term-major over the quadrant's eight tiles: an accumulator's next MFMA is eight issues away (same order of the three
    // terms per accumulator as everywhere: hi lo, lo hi, hi hi - same bits)
    if (p4::kTermMajor) {
      if (A_LO) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], cc[i][j], 0, 0, 0);
      }
      mid();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], cc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], cc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (A_LO) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], cc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], cc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], cc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // prologue: K-block 0 whole, landed and published to everybody
  p4::issue_unit<0, KB_BYTES>(src, smem, 0);
  p4::issue_unit<2, KB_BYTES>(src, smem, 0);
  p4::issue_unit<3, KB_BYTES>(src, smem, 0);
  p4::issue_unit<1, KB_BYTES>(src, smem, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();
  if (wm == 1) barrier();   // the second half of the workgroup runs one barrier behind the first from here on

  // Every K-block issues the units of block min(kb + 1, last): the last one fetches itself again into the idle buffer, so that
  // all blocks are the same straight-line code with the same counted wait (one block's worth of L2 reads per tile more).
  // Where in the phase the unit's two DMA instructions go (p4::kDmaPlace): an LDS-DMA costs its wave 100-185 cycles of issue
  // among LDS reads and ~60 among MFMAs (MI355X_MICROARCH.md), and the reading half of a phase must not outlast the other wave
  // group's 24 MFMAs.  0: with the fragment reads; 1: behind the first barrier, before the MFMAs; 2: between the first and the
  // second third of the MFMAs.  The unit a phase's wait must retire was issued two phases earlier: with the phase's own issue
  // in front of the wait (0) two newer units are outstanding - vmcnt(4) - with it behind the barrier (1, 2) one - vmcnt(2).
  auto none = [] {};
#ifdef DRIN_P4_STAMPS
  const bool stamp_on = blockIdx.x == 121 && M > 100000 && (threadIdx.x & 255) == 0 && !F16;
#endif
#define DRIN_P4_PHASE(READS, UNIT, ACC, PH)                              \
  {                                                                      \
    DRIN_P4_STAMP(0, kb, PH, 0);                                         \
    READS;                                                               \
    if (p4::kDmaPlace == 0) p4::issue_unit<UNIT, KB_BYTES>(src, nbuf, kn);         \
    p4::wait_units();                                                    \
    barrier();                                                           \
    DRIN_P4_STAMP(0, kb, PH, 1);                                         \
    if (p4::kDmaPlace == 1) p4::issue_unit<UNIT, KB_BYTES>(src, nbuf, kn);         \
    if (p4::kDmaPlace == 2)                                              \
      mma(ACC, [&] { p4::issue_unit<UNIT, KB_BYTES>(src, nbuf, kn); });            \
    else                                                                 \
      mma(ACC, none);                                                    \
    DRIN_P4_STAMP(0, kb, PH, 2);                                         \
    barrier();                                                           \
  }
  for (int kb = 0; kb < nkb; ++kb) {
    const char* buf = smem + (kb & 1) * p4::BUF;
    char* nbuf = smem + ((kb + 1) & 1) * p4::BUF;
    const int kn = kb + 1 < nkb ? kb + 1 : kb;
    DRIN_P4_PHASE((read_a(buf, 0), read_b(buf, 0)), 0, acc[0][0], 0)   // quadrant 00
    DRIN_P4_PHASE(read_b(buf, 1), 2, acc[0][1], 1)                      // quadrant 01 (A0 fragments stay)
    DRIN_P4_PHASE(read_a(buf, 1), 3, acc[1][1], 2)                      // quadrant 11 (B1 fragments stay)
    DRIN_P4_PHASE(read_b(buf, 0), 1, acc[1][0], 3)                      // quadrant 10 (A1 fragments stay)
  }
#undef DRIN_P4_PHASE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant units of the last block
  if (wm == 0) barrier();   // pairs with the delayed half's last barrier

  if (kpart > 0) {  // raw accumulators of a tail slice: [slot][256][256], folded into C by the tail-add launch
    float* part = tail + (size_t)tail_slot * (BM * BN);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (fp16 form: a slice carries its rows' scale like the part stored to C - powers of two, so the sum of scaled slices is the scaled sum)
        const int64_t row = m0 + a * 128 + wm * 64 + i * 16 + r;
        const int rs = F16 ? pow2_exponent(row_scale[row < M ? row : M - 1]) + pow2_exponent(*b_scale) : 0;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float4 o = make_float4(acc[a][b][i][j][0], acc[a][b][i][j][1], acc[a][b][i][j][2], acc[a][b][i][j][3]);
            if (F16) o = scale_pow2(o, rs);
            st4(part + (a * 128 + wm * 64 + i * 16 + r) * BN + b * 128 + wn * 32 + j * 16 + c * 4, o);
          }
      }
    return;
  }
  const int bs = F16 ? pow2_exponent(*b_scale) : 0;
  float4 bv[2][2];   // the bias of this lane's four column groups, fetched once (a load per store would serialise the stores)
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + b * 128 + wn * 32 + j * 16 + c * 4;
      bv[b][j] = (bias != nullptr && col < N) ? ld4(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = m0 + a * 128 + wm * 64 + i * 16 + r;
      if (row >= M) continue;
      const int rs = F16 ? pow2_exponent(row_scale[row]) + bs : 0;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + b * 128 + wn * 32 + j * 16 + c * 4;
          if (col >= N) continue;   // N % 4 == 0 (checked by the launcher)
          float4 o = make_float4(acc[a][b][i][j][0], acc[a][b][i][j][1], acc[a][b][i][j][2], acc[a][b][i][j][3]);
          if (F16) o = scale_pow2(o, rs);
#ifdef DRIN_ABLATE_GEMM_STORES   // timing ablation only: the product without its [M, N] fp32 result leaving the chip
          if (ldc < 0) st4(C + row * ldc + col, o + bv[b][j]);
#else
          st4(C + row * ldc + col, o + bv[b][j]);
#endif
        }
    }
}


// The four-phase pipeline for an fp32 A operand (the image rows of the folded path, the activations of the big training
// products): the B units stream by LDS-DMA as above; an A unit (128 rows x 32 k of fp32 = 16 KiB) is fetched into registers
// - two 16-byte loads per thread, issued where the DMA of that unit would be, by inline asm so that the only wait is
// the counted one - and, in the phase whose wait retires it, split into bf16 (hi, lo) and written to the unit's two planes
// (once per workgroup: 24 vector instructions per thread per unit, in the half of the phase in which the other wave group
// holds the matrix pipes).  Two loads per unit per wave like two DMA instructions: the vmcnt arithmetic is unchanged.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4p __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_store(f32x4v v, char* hi_plane, int off) {
  bf16x4p h, l;
#ifdef DRIN_P4_ABLATE_CVT   // timing ablation only (plain-bf16 results): a third of the split arithmetic, the same two LDS writes
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = (__bf16)v[k];
    l[k] = (__bf16)0.0f;
  }
#else
  split_bf16x4(make_float4(v[0], v[1], v[2], v[3]), h, l);
#endif
  *reinterpret_cast<bf16x4p*>(hi_plane + off) = h;
  *reinterpret_cast<bf16x4p*>(hi_plane + 128 * 64 + off) = l;
}

__global__ void __launch_bounds__(THREADS, 2)
    k_gemm_bf16x3_p4(const float* __restrict__ A, int64_t lda, const __bf16* __restrict__ b_hi, const __bf16* __restrict__ b_lo,
                     int64_t ldb, const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                     int accumulate, int nx, unsigned full, int ksplit, float* __restrict__ tail) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tx, kpart;
  int64_t ty;
  unsigned tail_slot;
  tile_of_block(tx, ty, nx, full, ksplit, kpart, tail_slot);
  const int64_t m0 = ty * BM;
  const int n0 = tx * BN;
  if (blockIdx.x >= full) {  // a K-slice of a tail tile
    K /= ksplit;
    const int64_t k0 = (int64_t)kpart * K;
    A += k0;
    b_hi += k0;
    b_lo += k0;
  }
  const int nkb = K / BK;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, c = lane >> 4;

  // B: DMA sources as in k_gemm_x3_planes_p4 (units 2, 3); A: this thread's float4 #(tid & 7) of rows (tid >> 3) + 64 i of a half
  p4::Src src;
  const float* arow[2][2];   // [half][i]
  int a_off[2];              // LDS offset of the 8-byte piece inside an A unit's hi plane
  {
    const int f = (0x78 >> (((lane >> 4) & 3) << 1)) & 3;
    const int chunk = (lane & 3) ^ f;
    const char* bp = reinterpret_cast<const char*>(wave >= 4 ? b_lo : b_hi) + chunk * 16;
#pragma unroll
    for (int u = 2; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int64_t g = (int64_t)n0 + (u & 1) * 128 + (wave & 3) * 32 + 16 * i + (lane >> 2);
        g = g < N ? g : (int64_t)N - 1;
        src.ptr[u][i] = bp + g * ldb * 2;
      }
    src.ptr[0][0] = src.ptr[0][1] = src.ptr[1][0] = src.ptr[1][1] = nullptr;
    const int c4 = threadIdx.x & 7;
    // which two rows of a 128-row unit this thread loads and splits.  kSplitInMma: a wave GROUP splits exactly the 64 rows its own
    // waves read (group g: rows 64 g .. 64 g + 63), so that the split may run inside the group's MFMA half - nobody of the other
    // group, which runs one barrier apart, ever reads what this group has not yet written.  Otherwise rows t / 8 and t / 8 + 64.
    int urow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      urow[i] = p4::kSplitInMma ? (int)(threadIdx.x >> 8) * 64 + (int)((threadIdx.x & 255) >> 3) + 32 * i : (int)(threadIdx.x >> 3) + 64 * i;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int64_t g = m0 + h * 128 + urow[i];
        g = g < M ? g : M - 1;
        arow[h][i] = A + g * lda + c4 * 4;
      }
#pragma unroll
    for (int i = 0; i < 2; ++i) a_off[i] = swz16(urow[i], c4 >> 1) + ((c4 & 1) << 3);
  }

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[a][b][i][j][v] = 0.f;

  bf16x8 ah[4], al[4], bh[2], bl[2];
  auto read_a = [&](const char* buf, int half) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* p = buf + half * p4::UNIT_BYTES + swz16(wm * 64 + i * 16 + r, c);
      ah[i] = *reinterpret_cast<const bf16x8*>(p);
      al[i] = *reinterpret_cast<const bf16x8*>(p + 128 * 64);
    }
  };
  auto read_b = [&](const char* buf, int half) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* p = buf + (2 + half) * p4::UNIT_BYTES + swz16(wn * 32 + j * 16 + r, c);
      bh[j] = *reinterpret_cast<const bf16x8*>(p);
      bl[j] = *reinterpret_cast<const bf16x8*>(p + 128 * 64);
    }
  };
  auto mma = [&](f32x4 (&cc)[4][2], auto&& mid) {
    __builtin_amdgcn_s_setprio(1);
    // term-major over the quadrant's eight tiles: an accumulator's next MFMA is eight issues away (same order of the three
    // terms per accumulator as everywhere: hi lo, lo hi, hi hi - same bits)
    if (p4::kTermMajor) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], cc[i][j], 0, 0, 0);
      mid();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], cc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], cc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], cc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], cc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], cc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // A unit `half` of K-block kb into registers: two loads the compiler does not track (the counted waits below are the only ones)
  auto load_a = [&](int half, int kb, f32x4v& v0, f32x4v& v1) {
    const float* p0 = arow[half][0] + (int64_t)kb * BK;
    const float* p1 = arow[half][1] + (int64_t)kb * BK;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v0) : "v"(p0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v1) : "v"(p1) : "memory");
  };
  // The counted wait of a phase: every unit but this wave's two newest has landed.  The registers of the A unit it retires
  // are named by an EMPTY statement behind it: whatever copies the compiler wants for their consumers come after the wait.
  auto wait4 = [&]() { p4::wait_units(); };
  auto landed = [&](f32x4v& v0, f32x4v& v1) { asm volatile("" : "+v"(v0), "+v"(v1) : : "memory"); };
  auto store_a = [&](char* buf, int half, f32x4v v0, f32x4v v1) {
    split_store(v0, buf + half * p4::UNIT_BYTES, a_off[0]);
    split_store(v1, buf + half * p4::UNIT_BYTES, a_off[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the writes are in LDS before the barrier that publishes them
  };

  f32x4v a0v0, a0v1, a1v0, a1v1;   // the A0 / A1 unit in flight
  // prologue: K-block 0 - A0 converted and published, A1 landed in registers (phase 1 converts it), B0 and B1 by DMA
  load_a(0, 0, a0v0, a0v1);
  p4::issue_unit<2>(src, smem, 0);
  p4::issue_unit<3>(src, smem, 0);
  load_a(1, 0, a1v0, a1v1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  landed(a0v0, a0v1);
  landed(a1v0, a1v1);
  store_a(smem, 0, a0v0, a0v1);
  barrier();
  if (wm == 1) barrier();

  // Every K-block issues the units of block min(kb + 1, last): the last one fetches itself again into the idle buffer, so
  // that all blocks are the same straight-line code with the same counted waits (one block's worth of L2 reads per tile
  // more: 4 % at K = 768).
  if (p4::kRebalance) {
    // The phases' reading halves evened out (an LDS-DMA costs its wave ~150 cycles of issue among LDS reads, a register load
    // ~40, a unit's split ~250 with its two writes: with the DMA of B0 and the split of A1 in one phase that phase's reading
    // half lasted ~600 cycles against the 384 of the other group's MFMAs).  DMA in the phases that read most fragments, loads
    // and splits in the others; the B0 fragments stay in registers from quadrant 00 to quadrant 10, so B0's slot is free for
    // the next block's DMA four phases after its only read:
    //   phase 0 (00): read A0 B0 | DMA  B0(kn)        phase 1 (01): read B1 | load A0(kn) | split A1(kb)
    //   phase 2 (11): read A1    | DMA  B1(kn)        phase 3 (10): -       | load A1(kn) | split A0(kn)
    // Issue order B0', A0', B1', A1': vmcnt(4) before a phase's barrier retires the unit issued two phases earlier - A1(kb)
    // for phase 1's split, A0(kn) for phase 3's, B0(kn) / B1(kn) one or more phases before their first read.
    bf16x8 b0h[2], b0l[2];
    auto mma_b = [&](f32x4 (&cc)[4][2], const bf16x8 (&fh)[2], const bf16x8 (&fl)[2]) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[j], al[i], cc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[j], ah[i], cc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[j], ah[i], cc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    // the same MFMAs with the split of one activation unit between the three groups of eight (kSplitInMma)
    auto mma_split = [&](f32x4 (&cc)[4][2], const bf16x8 (&fh)[2], const bf16x8 (&fl)[2], char* dst_buf, int half, f32x4v v0, f32x4v v1) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[j], al[i], cc[i][j], 0, 0, 0);
      split_store(v0, dst_buf + half * p4::UNIT_BYTES, a_off[0]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[j], ah[i], cc[i][j], 0, 0, 0);
      split_store(v1, dst_buf + half * p4::UNIT_BYTES, a_off[1]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[j], ah[i], cc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the planes are in LDS before the barrier that publishes them to this group
    };
#ifdef DRIN_P4_STAMPS
    const bool stamp_on = blockIdx.x == 121 && M > 100000 && (threadIdx.x & 255) == 0;
#endif
    for (int kb = 0; kb < nkb; ++kb) {
      char* buf = smem + (kb & 1) * p4::BUF;
      char* nbuf = smem + ((kb + 1) & 1) * p4::BUF;
      const int kn = kb + 1 < nkb ? kb + 1 : kb;
      // phase 0: quadrant 00
      DRIN_P4_STAMP(1, kb, 0, 0);
      read_a(buf, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* p = buf + 2 * p4::UNIT_BYTES + swz16(wn * 32 + j * 16 + r, c);
        b0h[j] = *reinterpret_cast<const bf16x8*>(p);
        b0l[j] = *reinterpret_cast<const bf16x8*>(p + 128 * 64);
      }
      p4::issue_unit<2>(src, nbuf, kn);
      wait4();
      barrier();
      DRIN_P4_STAMP(1, kb, 0, 1);
      mma_b(acc[0][0], b0h, b0l);
      DRIN_P4_STAMP(1, kb, 0, 2);
      barrier();
      // phase 1: quadrant 01
      DRIN_P4_STAMP(1, kb, 1, 0);
      read_b(buf, 1);
      load_a(0, kn, a0v0, a0v1);
      wait4();
      landed(a1v0, a1v1);
      if (p4::kSplitInMma) {
        barrier();
        DRIN_P4_STAMP(1, kb, 1, 1);
        mma_split(acc[0][1], bh, bl, buf, 1, a1v0, a1v1);
      } else {
        store_a(buf, 1, a1v0, a1v1);
        barrier();
        DRIN_P4_STAMP(1, kb, 1, 1);
        mma_b(acc[0][1], bh, bl);
      }
      DRIN_P4_STAMP(1, kb, 1, 2);
      barrier();
      // phase 2: quadrant 11
      DRIN_P4_STAMP(1, kb, 2, 0);
      read_a(buf, 1);
      p4::issue_unit<3>(src, nbuf, kn);
      wait4();
      barrier();
      DRIN_P4_STAMP(1, kb, 2, 1);
      mma_b(acc[1][1], bh, bl);
      DRIN_P4_STAMP(1, kb, 2, 2);
      barrier();
      // phase 3: quadrant 10
      DRIN_P4_STAMP(1, kb, 3, 0);
      load_a(1, kn, a1v0, a1v1);
      wait4();
      landed(a0v0, a0v1);
      if (p4::kSplitInMma) {
        barrier();
        DRIN_P4_STAMP(1, kb, 3, 1);
        mma_split(acc[1][0], b0h, b0l, nbuf, 0, a0v0, a0v1);
      } else {
        store_a(nbuf, 0, a0v0, a0v1);
        barrier();
        DRIN_P4_STAMP(1, kb, 3, 1);
        mma_b(acc[1][0], b0h, b0l);
      }
      DRIN_P4_STAMP(1, kb, 3, 2);
      barrier();
    }
  } else
  for (int kb = 0; kb < nkb; ++kb) {
    char* buf = smem + (kb & 1) * p4::BUF;
    char* nbuf = smem + ((kb + 1) & 1) * p4::BUF;
    const int kn = kb + 1 < nkb ? kb + 1 : kb;
    auto none = [] {};
    // (p4::kDmaPlace: where the phase's two loads / DMA instructions go - see k_gemm_x3_planes_p4)
    // phase 0: quadrant 00; A0(kn) leaves for the registers
    read_a(buf, 0);
    read_b(buf, 0);
    if (p4::kDmaPlace == 0) load_a(0, kn, a0v0, a0v1);
    wait4();
    barrier();
    if (p4::kDmaPlace == 1) load_a(0, kn, a0v0, a0v1);
    if (p4::kDmaPlace == 2) mma(acc[0][0], [&] { load_a(0, kn, a0v0, a0v1); }); else mma(acc[0][0], none);
    barrier();
    // phase 1: quadrant 01; B0(kn) leaves; A1(kb) has landed: split and publish it
    read_b(buf, 1);
    if (p4::kDmaPlace == 0) p4::issue_unit<2>(src, nbuf, kn);
    wait4();
    landed(a1v0, a1v1);
    store_a(buf, 1, a1v0, a1v1);
    barrier();
    if (p4::kDmaPlace == 1) p4::issue_unit<2>(src, nbuf, kn);
    if (p4::kDmaPlace == 2) mma(acc[0][1], [&] { p4::issue_unit<2>(src, nbuf, kn); }); else mma(acc[0][1], none);
    barrier();
    // phase 2: quadrant 11; B1(kn) leaves
    read_a(buf, 1);
    if (p4::kDmaPlace == 0) p4::issue_unit<3>(src, nbuf, kn);
    wait4();
    barrier();
    if (p4::kDmaPlace == 1) p4::issue_unit<3>(src, nbuf, kn);
    if (p4::kDmaPlace == 2) mma(acc[1][1], [&] { p4::issue_unit<3>(src, nbuf, kn); }); else mma(acc[1][1], none);
    barrier();
    // phase 3: quadrant 10; A1(kn) leaves; A0(kn) has landed: split and publish it
    read_b(buf, 0);
    if (p4::kDmaPlace == 0) load_a(1, kn, a1v0, a1v1);
    wait4();
    landed(a0v0, a0v1);
    store_a(nbuf, 0, a0v0, a0v1);
    barrier();
    if (p4::kDmaPlace == 1) load_a(1, kn, a1v0, a1v1);
    if (p4::kDmaPlace == 2) mma(acc[1][0], [&] { load_a(1, kn, a1v0, a1v1); }); else mma(acc[1][0], none);
    barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant units of the last block
  if (wm == 0) barrier();

  if (kpart > 0) {
    float* part = tail + (size_t)tail_slot * (BM * BN);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            st4(part + (a * 128 + wm * 64 + i * 16 + r) * BN + b * 128 + wn * 32 + j * 16 + c * 4,
                make_float4(acc[a][b][i][j][0], acc[a][b][i][j][1], acc[a][b][i][j][2], acc[a][b][i][j][3]));
    return;
  }
  float4 bv[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + b * 128 + wn * 32 + j * 16 + c * 4;
      bv[b][j] = (bias != nullptr && col < N) ? ld4(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = m0 + a * 128 + wm * 64 + i * 16 + r;
      if (row >= M) continue;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + b * 128 + wn * 32 + j * 16 + c * 4;
          if (col >= N) continue;
          float* dst = C + row * ldc + col;
          float4 o = make_float4(acc[a][b][i][j][0], acc[a][b][i][j][1], acc[a][b][i][j][2], acc[a][b][i][j][3]) + bv[b][j];
          if (accumulate) o = o + ld4(dst);
#ifdef DRIN_ABLATE_GEMM_STORES
          if (ldc < 0)
#endif
          st4(dst, o);
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
  split_bf16x4(v, h, l);
  *reinterpret_cast<bf16x4*>(hi + i * 4) = h;
  *reinterpret_cast<bf16x4*>(lo + i * 4) = l;
}

// the same for up to 32 tensors in one launch (blockIdx.y = which): the weights a training step's split-bf16 NT products run
// against, split ONCE per forward / backward instead of once per workgroup per K-block by the register stager
__global__ void __launch_bounds__(256) k_split_planes_batch(const SplitBatch b) {
  const int w = blockIdx.y;
  const int64_t n4 = b.n4[w];
  const float* __restrict__ x = b.src[w];
  __bf16* __restrict__ hi = reinterpret_cast<__bf16*>(b.planes[w]);
  __bf16* __restrict__ lo = hi + n4 * 4;
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = ld4(x + i * 4);
    bf16x4 h, l;
    split_bf16x4(v, h, l);
    *reinterpret_cast<bf16x4*>(hi + i * 4) = h;
    *reinterpret_cast<bf16x4*>(lo + i * 4) = l;
  }
}

// planes[w] = (hi, lo) of src[w]^T for equally shaped [rows][cols] matrices: the transposed weights of the backward dX = dY W
// products, transposed AND split once per call
__global__ void __launch_bounds__(256) k_transpose_split_batch(const SplitBatch b, int rows, int cols) {
  __shared__ float tile[32][33];
  const float* __restrict__ in = b.src[blockIdx.z];
  __bf16* __restrict__ hi = reinterpret_cast<__bf16*>(b.planes[blockIdx.z]);
  __bf16* __restrict__ lo = hi + (int64_t)rows * cols;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(int64_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) {
      const float v = tile[tx][i];
      const __bf16 h = (__bf16)v;
      const int64_t o = (int64_t)(c0 + i) * rows + r0 + tx;
      hi[o] = h;
      lo[o] = (__bf16)(v - (float)h);
    }
}

}  // namespace x3p

#ifdef DRIN_P4_STAMPS
extern "C" __attribute__((visibility("default"))) int drin_debug_p4_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(drin::x3p::g_p4_stamps), sizeof(unsigned long long) * 2 * 2 * 32 * 4 * 3);
}
#endif
#ifdef DRIN_STAMPS
extern "C" __attribute__((visibility("default"))) int drin_debug_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(x3p::g_stamps), sizeof(unsigned long long) * (8 * 64 * 4 + 64));
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

int SplitBatch::add(const float* w, float* planes_out, int64_t numel) {
  if (!w || numel <= 0) return DRIN_OK;
  if (n >= 32 || (numel % 4) || !aligned16(w) || !aligned16(planes_out)) {
    set_error("split_planes: at most 32 tensors per batch, element counts multiples of 4, 16-byte aligned");
    return DRIN_E_SHAPE;
  }
  src[n] = w, planes[n] = planes_out, n4[n] = numel / 4;
  ++n;
  return DRIN_OK;
}

int launch_split_planes_batch(const SplitBatch& b, hipStream_t st) {
  if (b.n == 0) return DRIN_OK;
  int64_t most = 0;
  for (int i = 0; i < b.n; ++i) most = most > b.n4[i] ? most : b.n4[i];
  int64_t gx = cdiv(most, 256);
  if (gx > 2048) gx = 2048;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(x3p::k_split_planes_batch, dim3((unsigned)gx, (unsigned)b.n), dim3(256), 0, st, b);
  DRIN_CHECK_LAUNCH("k_split_planes_batch");
  return DRIN_OK;
}

int launch_transpose_split_batch(const SplitBatch& b, int rows, int cols, hipStream_t st) {
  if (b.n == 0 || rows <= 0 || cols <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(x3p::k_transpose_split_batch, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32), (unsigned)b.n), dim3(256),
                     0, st, b, rows, cols);
  DRIN_CHECK_LAUNCH("k_transpose_split_batch");
  return DRIN_OK;
}

// The partly filled last round of 256 x 256 tiles as K-slices over the idle CUs (both split-bf16 NT kernel families)
static void tail_split_256(int64_t tiles, int nkb, const float* tail, size_t tail_floats, unsigned* full, int* ksplit) {
  *full = (unsigned)tiles;
  *ksplit = 1;
  if (tail == nullptr || !aligned16(tail) || tiles > 16 * 256) return;
  const unsigned frac = (unsigned)(tiles % 256);
  for (int s = 4; s >= 2 && frac > 0; --s)
    if ((unsigned)s * frac <= 256 && nkb % s == 0 && nkb / s >= 4 && (size_t)frac * (s - 1) * (x3p::BM * x3p::BN) <= tail_floats) {
      *ksplit = s;
      *full = (unsigned)tiles - frac;
      return;
    }
}

// The four-phase pipeline on an fp32 A operand against pre-split weight planes (256 x 256 tiles; DRIN_E_UNSUPPORTED outside
// its contract: the caller keeps its other kernel).  tail: optional scratch for the tail split, as launch_gemm_nt_bf16x3.
int launch_gemm_nt_bf16x3_p4(const float* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw, const float* bias,
                             float* y, int64_t ldy, int64_t M, int N, int K, hipStream_t st, bool accumulate, float* tail,
                             size_t tail_floats) {
  if (M <= 0 || N <= 0) return DRIN_OK;
  if (K <= 0 || (K % x3p::BK) || (ldx % 4) || (ldw % 8) || (N % 4) || (ldy % 4) || !aligned16(x) || !aligned16(w_hi) || !aligned16(w_lo) ||
      !aligned16(y) || (bias != nullptr && !aligned16(bias))) {
    set_error("gemm_bf16x3_p4: shape / alignment outside the kernel's contract");
    return DRIN_E_UNSUPPORTED;
  }
  const int nx = (int)cdiv(N, x3p::BN);
  const int64_t tiles = cdiv(M, x3p::BM) * nx;
  if (tiles > ((int64_t)1 << 28)) return DRIN_E_UNSUPPORTED;
  unsigned full;
  int ksplit;
  tail_split_256(tiles, K / x3p::BK, tail, tail_floats, &full, &ksplit);
  const unsigned items = full + ((unsigned)tiles - full) * (unsigned)ksplit;
  static DynLdsOptIn opt;
  DRIN_TRY(ensure_dynamic_lds(opt, reinterpret_cast<const void*>(x3p::k_gemm_bf16x3_p4), x3p::p4::LDS, "hipFuncSetAttribute(gemm_bf16x3_p4)"));
  {
    KernelTimer timer(DRIN_KC_GEMM_X3, st);
    hipLaunchKernelGGL(x3p::k_gemm_bf16x3_p4, dim3(items), dim3(x3p::THREADS), x3p::p4::LDS, st, x, ldx, (const __bf16*)w_hi,
                       (const __bf16*)w_lo, ldw, bias, y, ldy, M, N, K, accumulate ? 1 : 0, nx, full, ksplit, tail);
    DRIN_CHECK_LAUNCH("k_gemm_bf16x3_p4");
  }
  if (ksplit > 1) DRIN_TRY(launch_tail_add_256(tail, y, ldy, M, N, (unsigned)nx, full, (unsigned)tiles - full, ksplit, st));
  return DRIN_OK;
}

int launch_gemm_x3_planes(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo,
                          int64_t ldb, const float* bias, float* y, int64_t ldy, int64_t M, int N, int K,
                          hipStream_t st, float* splitk, size_t splitk_floats) {
  if (M <= 0 || N <= 0) return DRIN_OK;
  const bool a_lo_plane = a_lo != nullptr;  // NULL: A is exact in bf16 (one plane, two MFMAs per tile pair)
  if (b_lo == nullptr) {
    set_error("gemm_x3_planes: the weight operand needs both planes");
    return DRIN_E_UNSUPPORTED;
  }
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
  {
    static DynLdsOptIn opt_in[2];
    const void* kernels[2] = {reinterpret_cast<const void*>(x3p::k_gemm_x3_planes<true, true>),
                              reinterpret_cast<const void*>(x3p::k_gemm_x3_planes<false, true>)};
    for (int i = 0; i < 2; ++i)
      DRIN_TRY(ensure_dynamic_lds(opt_in[i], kernels[i], x3p::LDS_BYTES, "hipFuncSetAttribute(gemm_x3_planes)"));
  }
  // A handful of mentions is a handful of tiles walking K serially (101 rows x 768 x 768: 3 workgroups, 50 us): split K
  // over workgroups into scratch and add the slices in order (deterministic), as the mention-sized fp32 products do.
  int splits = 1;
  const int nkb = K / x3p::BK;
  const int64_t tiles = mt * cdiv(N, x3p::BN);
  if (splitk != nullptr && tiles <= 128 && (N % 4) == 0 && (ldy % 4) == 0 && aligned16(y) && aligned16(splitk)) {
    for (int s : {16, 8, 6, 4, 3, 2})
      if (tiles * s <= 256 && nkb % s == 0 && nkb / s >= 3 && (size_t)s * M * N <= splitk_floats) {
        splits = s;
        break;
      }
  }
  // otherwise, a partly filled last round of tiles (fewer than half the CUs) splits K over the idle CUs
  const int nx = (int)cdiv(N, x3p::BN);
  unsigned full = (unsigned)tiles;
  int ksplit = 1;
  if (splits == 1) tail_split_256(tiles, nkb, splitk, splitk_floats, &full, &ksplit);
  // the full split product on whole tiles + tail slices: the four-phase pipeline
  // (an activation operand without a lo plane - exact in bf16 - takes the A_LO = false instantiation from half a round of tiles up;
  //  smaller grids of it keep the two-phase kernel and its split-K forms)
  if ((a_lo_plane || tiles >= 128) && splits == 1 && (N % 4) == 0 && (ldy % 4) == 0 && aligned16(y) && (bias == nullptr || aligned16(bias))) {
    static DynLdsOptIn opt[2];
    auto kern = a_lo_plane ? x3p::k_gemm_x3_planes_p4<true, false> : x3p::k_gemm_x3_planes_p4<false, false>;
    DRIN_TRY(ensure_dynamic_lds(opt[a_lo_plane ? 0 : 1], reinterpret_cast<const void*>(kern), x3p::p4::LDS, "hipFuncSetAttribute(gemm_x3_planes_p4)"));
    const unsigned p4_items = full + ((unsigned)tiles - full) * (unsigned)ksplit;
    {
      KernelTimer timer(DRIN_KC_GEMM_PLANES, st);
      hipLaunchKernelGGL(kern, dim3(p4_items), dim3(x3p::THREADS), x3p::p4::LDS, st, (const __bf16*)a_hi,
                         (const __bf16*)(a_lo_plane ? a_lo : a_hi), lda, (const __bf16*)b_hi, (const __bf16*)b_lo, ldb, bias, y, ldy, M, N, K, nx,
                         full, ksplit, splitk, (const float*)nullptr, (const float*)nullptr);
      DRIN_CHECK_LAUNCH("k_gemm_x3_planes_p4");
    }
    if (ksplit > 1) DRIN_TRY(launch_tail_add_256(splitk, y, ldy, M, N, (unsigned)nx, full, (unsigned)tiles - full, ksplit, st));
    return DRIN_OK;
  }
  const unsigned items = full + ((unsigned)tiles - full) * (unsigned)ksplit;
  dim3 grid(items, 1, (unsigned)splits);
  const int64_t part_stride = M * (int64_t)N;
  float* out = splits > 1 ? splitk : y;
  {
    KernelTimer timer(DRIN_KC_GEMM_PLANES, st);
    const __bf16 *ah = (const __bf16*)a_hi, *al = (const __bf16*)a_lo, *bh = (const __bf16*)b_hi, *bl = (const __bf16*)b_lo;
    if (a_lo_plane)
      hipLaunchKernelGGL((x3p::k_gemm_x3_planes<true, true>), grid, dim3(x3p::THREADS), x3p::LDS_BYTES, st, ah, al, lda, bh,
                         bl, ldb, bias, out, ldy, M, N, K, part_stride, nx, full, ksplit, splitk);
    else
      hipLaunchKernelGGL((x3p::k_gemm_x3_planes<false, true>), grid, dim3(x3p::THREADS), x3p::LDS_BYTES, st, ah, al, lda, bh,
                         bl, ldb, bias, out, ldy, M, N, K, part_stride, nx, full, ksplit, splitk);
    DRIN_CHECK_LAUNCH("k_gemm_x3_planes");
  }
  if (splits > 1) DRIN_TRY(launch_splitk_reduce(splitk, splits, part_stride, bias, y, ldy, M, N, false, st));
  if (ksplit > 1) DRIN_TRY(launch_tail_add_256(splitk, y, ldy, M, N, (unsigned)nx, full, (unsigned)tiles - full, ksplit, st));
  return DRIN_OK;
}

// ONE fp16 MFMA pass on single fp16 planes (the F16 instantiation of the four-phase kernel): see internal.h
bool gemm_f16_planes_fits(const void* a_f16, int64_t lda, const void* b_f16, int64_t ldb, const float* y, int64_t ldy, int64_t M, int N,
                          int K) {
  return cdiv(M, x3p::BM) * cdiv(N, x3p::BN) >= 128 && cdiv(M, x3p::BM) * cdiv(N, x3p::BN) <= ((int64_t)1 << 28) && K > 0 &&
         (K % (2 * x3p::BK)) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (N % 4) == 0 && (ldy % 4) == 0 && aligned16(a_f16) &&
         aligned16(b_f16) && aligned16(y);
}

int launch_gemm_f16_planes(const void* a_f16, int64_t lda, const void* b_f16, int64_t ldb, const float* row_scale, const float* b_scale,
                           float* y, int64_t ldy, int64_t M, int N, int K, hipStream_t st, float* tail, size_t tail_floats) {
  if (M <= 0 || N <= 0) return DRIN_OK;
  if (row_scale == nullptr || b_scale == nullptr || !gemm_f16_planes_fits(a_f16, lda, b_f16, ldb, y, ldy, M, N, K)) {
    set_error("gemm_f16_planes: built for at least half a round of 256 x 256 tiles, K %% 64 == 0, 16-byte aligned planes and both scales");
    return DRIN_E_UNSUPPORTED;
  }
  const int nx = (int)cdiv(N, x3p::BN);
  const int64_t tiles = cdiv(M, x3p::BM) * nx;
  unsigned full;
  int ksplit;
  tail_split_256(tiles, K / (2 * x3p::BK), tail, tail_floats, &full, &ksplit);
  const unsigned items = full + ((unsigned)tiles - full) * (unsigned)ksplit;
  auto kern = x3p::k_gemm_x3_planes_p4<true, true>;
  static DynLdsOptIn opt;
  DRIN_TRY(ensure_dynamic_lds(opt, reinterpret_cast<const void*>(kern), x3p::p4::LDS, "hipFuncSetAttribute(gemm_f16_planes)"));
  {
    KernelTimer timer(DRIN_KC_GEMM_X3, st);   // (the class of the product it replaces: x_i C_i^T)
    hipLaunchKernelGGL(kern, dim3(items), dim3(x3p::THREADS), x3p::p4::LDS, st, (const __bf16*)a_f16, (const __bf16*)a_f16, lda,
                       (const __bf16*)b_f16, (const __bf16*)b_f16, ldb, (const float*)nullptr, y, ldy, M, N, K, nx, full, ksplit, tail, row_scale,
                       b_scale);
    DRIN_CHECK_LAUNCH("k_gemm_f16_planes");
  }
  if (ksplit > 1) DRIN_TRY(launch_tail_add_256(tail, y, ldy, M, N, (unsigned)nx, full, (unsigned)tiles - full, ksplit, st));
  return DRIN_OK;
}

}  // namespace drin
