// Stand-alone probe (not part of the library): how fast can a SUBSET of the CUs stream 3 KB rows from HBM?
// The stream kernel and the MFMA-bound contractions only overlap on disjoint CU sets (tools/cumask_probe.py); whether a partition
// pays depends on the rate a CU can pull when fewer than 256 of them read.  Access pattern of k_entity_stream (a workgroup owns 16
// runs of 34 consecutive 3 KB rows, a wave walks four of them, IN_FLIGHT rows requested ahead), non-temporal register loads, on
// streams created with hipExtStreamCreateWithCUMask (mask bit i lands on XCD i % 8: any prefix is spread evenly over the XCDs).
// Variants: rows in flight per wave x workgroups resident per CU (held down by an LDS pad).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/partition_probe.hip -o partition_probe && ./partition_probe [GiB]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

constexpr int ROW_FLOATS = 768, ROWS_PER_PAIR = 34, PAIRS_PER_WG = 16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int IN_FLIGHT>
__global__ void __launch_bounds__(256) k_regs(const float* __restrict__ src, float* __restrict__ out, int64_t pairs) {
  extern __shared__ char pad[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (pairs < 0) pad[threadIdx.x] = 0;  // keeps the pad allocated
  for (int q = wave; q < PAIRS_PER_WG; q += 4) {
    const int64_t p = (int64_t)blockIdx.x * PAIRS_PER_WG + q;
    if (p >= pairs) return;
    const float* base = src + p * (int64_t)ROWS_PER_PAIR * ROW_FLOATS + lane * 4;
    float4 acc[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    int r = 0;
    for (; r + IN_FLIGHT <= ROWS_PER_PAIR; r += IN_FLIGHT) {
      float4 v[IN_FLIGHT][3];
#pragma unroll
      for (int i = 0; i < IN_FLIGHT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v[i][j] = ldnt(base + (int64_t)(r + i) * ROW_FLOATS + j * 256);
#pragma unroll
      for (int i = 0; i < IN_FLIGHT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = add4(acc[j], v[i][j]);
    }
    for (; r < ROWS_PER_PAIR; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = add4(acc[j], ldnt(base + (int64_t)r * ROW_FLOATS + j * 256));
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(out + p * ROW_FLOATS + lane * 4 + j * 256) = acc[j];
  }
}

static hipStream_t masked_stream(int cus) {
  uint32_t words[8] = {0};
  for (int c = 0; c < cus; ++c) words[c / 32] |= 1u << (c % 32);
  hipStream_t s;
  CHECK(hipExtStreamCreateWithCUMask(&s, 8, words));
  return s;
}

template <typename F>
static double time_ms(F launch, hipStream_t st, int iters) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipStreamSynchronize(st));
  CHECK(hipEventRecord(a, st));
  for (int i = 0; i < iters; ++i) launch();
  CHECK(hipEventRecord(b, st));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

template <int IN_FLIGHT>
static void sweep(const float* src, float* out, int64_t pairs, double bytes, const int* cus, hipStream_t* streams, int n) {
  const unsigned grid = (unsigned)(pairs / PAIRS_PER_WG);
  auto kern = k_regs<IN_FLIGHT>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  for (int wgs : {2, 4, 8}) {  // resident workgroups per CU (4 waves each)
    const size_t lds = (size_t)(160 * 1024 / wgs) - 1024;
    printf("rows in flight %d, %d workgroups/CU (%3d KB requested per CU):", IN_FLIGHT, wgs, IN_FLIGHT * 3 * 4 * wgs);
    for (int i = 0; i < n; ++i) {
      hipStream_t st = streams[i];
      const double t = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, out, pairs); }, st, 3);
      printf("  %3d CUs %6.0f GB/s (%5.1f per CU)", cus[i], bytes / t / 1e6, bytes / t / 1e6 / cus[i]);
    }
    printf("\n");
    fflush(stdout);
  }
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 16.0;
  const int64_t pair_bytes = (int64_t)ROWS_PER_PAIR * ROW_FLOATS * 4;
  int64_t pairs = (int64_t)(gib * (1ll << 30)) / pair_bytes;
  pairs -= pairs % PAIRS_PER_WG;
  float *src = nullptr, *out = nullptr;
  CHECK(hipMalloc((void**)&src, pairs * pair_bytes));
  CHECK(hipMalloc((void**)&out, pairs * ROW_FLOATS * 4));
  CHECK(hipMemset(src, 0, pairs * pair_bytes));
  const int cus[] = {256, 192, 160, 128, 112, 96, 64};
  constexpr int n = sizeof(cus) / sizeof(cus[0]);
  hipStream_t streams[n];
  for (int i = 0; i < n; ++i) streams[i] = masked_stream(cus[i]);
  const double bytes = (double)pairs * pair_bytes;
  printf("%.1f GiB, %lld runs of %d rows x 3 KB\n", bytes / (1ll << 30), (long long)pairs, ROWS_PER_PAIR);
  sweep<2>(src, out, pairs, bytes, cus, streams, n);
  sweep<4>(src, out, pairs, bytes, cus, streams, n);
  sweep<8>(src, out, pairs, bytes, cus, streams, n);
  CHECK(hipFree(src));
  CHECK(hipFree(out));
  return 0;
}
