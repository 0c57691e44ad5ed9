// Caller-side optimiser step of the training configs: `torch.optim.Adam(self.parameters(), lr)` of train.py:55-56 as
// ONE launch over a flat fp32 parameter bucket (torch's default multi-tensor implementation: nine launches, 150 us at
// 6.7 M parameters - a twelfth of the B = 64 training step).
//
// Arithmetic: exactly the op sequence of torch's default (foreach) Adam, each op rounded to fp32 once, so that a loop
// stepped with this kernel tracks a loop stepped with torch.optim.Adam bit for bit (Adam divides by sqrt(v): a different
// rounding is amplified - torch's own fused=True variant drifts 1.4e-3 in the loss after two epochs):
//     exp_avg      = lerp(exp_avg, grad, 1 - beta1)                 _foreach_lerp_     a + w (b - a)        [|w| < 0.5]
//     exp_avg_sq   = exp_avg_sq * beta2                             _foreach_mul_
//     exp_avg_sq   = exp_avg_sq + (1 - beta2) (grad grad)           _foreach_addcmul_  a + s (t1 t2)
//     denom        = sqrt(exp_avg_sq) / sqrt(bias_correction2)      _foreach_sqrt, _foreach_div_
//     denom        = denom + eps                                    _foreach_add_
//     param        = param + (-lr / bias_correction1) (exp_avg / denom)   _foreach_addcdiv_  a + s (t1 / t2)
// The scalars are formed on the host in double precision exactly as torch/optim/adam.py does and rounded to fp32 at the
// call, as the foreach kernels' `scalar.to<opmath_t>()` does.  Which roundings torch's kernels make on this GPU was
// measured (tools/adam_diag.py against float64 emulations rounded once): every `a + s x` above is ONE fma, sqrt and both
// divisions are correctly rounded.  tests/test_gpu_round2.py::test_library_adam_matches_torch_adam_bitwise pins it.
// (HIP's __fsqrt_rn / __fdiv_rn are NOT the rounded operations unless OCML_BASIC_ROUNDED_OPERATIONS is defined -
// __fsqrt_rn is the native approximation - so the kernel uses sqrtf and '/', which hipcc compiles correctly rounded by
// default, with contraction switched off so that exactly the three fmas written below are fused.)
#include "internal.h"

namespace drin {

__device__ __forceinline__ void adam_one(float& p, const float g, float& m, float& v, const float w1, const float beta2,
                                         const float w2, const float bc2_sqrt, const float eps, const float neg_step) {
#pragma clang fp contract(off)
  const float diff = g - m;
  m = __builtin_fmaf(w1, diff, m);
  v = v * beta2;
  const float gg = g * g;
  v = __builtin_fmaf(w2, gg, v);
  float d = sqrtf(v);
  d = d / bc2_sqrt;
  d = d + eps;
  const float q = m / d;
  p = __builtin_fmaf(neg_step, q, p);
}

__global__ void __launch_bounds__(256) k_adam(float* __restrict__ param, const float* __restrict__ grad,
                                              float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, const int64_t n,
                                              const float w1, const float beta2, const float w2, const float bc2_sqrt,
                                              const float eps, const float neg_step) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4*>(param)[i];
    const float4 g = reinterpret_cast<const float4*>(grad)[i];
    float4 m = reinterpret_cast<float4*>(exp_avg)[i];
    float4 v = reinterpret_cast<float4*>(exp_avg_sq)[i];
    adam_one(p.x, g.x, m.x, v.x, w1, beta2, w2, bc2_sqrt, eps, neg_step);
    adam_one(p.y, g.y, m.y, v.y, w1, beta2, w2, bc2_sqrt, eps, neg_step);
    adam_one(p.z, g.z, m.z, v.z, w1, beta2, w2, bc2_sqrt, eps, neg_step);
    adam_one(p.w, g.w, m.w, v.w, w1, beta2, w2, bc2_sqrt, eps, neg_step);
    reinterpret_cast<float4*>(param)[i] = p;
    reinterpret_cast<float4*>(exp_avg)[i] = m;
    reinterpret_cast<float4*>(exp_avg_sq)[i] = v;
  }
  // ragged tail (n % 4 elements): the first threads of block 0
  const int64_t tail = n4 << 2;
  if (blockIdx.x == 0 && tail + threadIdx.x < n) {
    const int64_t i = tail + threadIdx.x;
    float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
    adam_one(p, grad[i], m, v, w1, beta2, w2, bc2_sqrt, eps, neg_step);
    param[i] = p;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
  }
}

static int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float w1, float beta2, float w2, float bc2s,
                       float eps, float neg_step, hipStream_t st) {
  const int64_t n4 = n >> 2;
  int64_t blocks = cdiv(n4 > 0 ? n4 : 1, 256);
  if (blocks > 256 * 8) blocks = 256 * 8;  // 8 workgroups per CU, grid-stride beyond
  KernelTimer timer(DRIN_KC_OPTIM, st);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, w1, beta2, w2, bc2s, eps, neg_step);
  DRIN_CHECK_LAUNCH("k_adam");
  return DRIN_OK;
}

}  // namespace drin

using namespace drin;

extern "C" int drin_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lerp_weight,
                              float beta2, float one_minus_beta2, float bias_correction2_sqrt, float eps, float neg_step_size,
                              void* stream) {
  DRIN_BIND_DEVICE(stream, param, "drin_adam_step");
  RoctxRange range("drin_adam_step");
  if (!param || !grad || !exp_avg || !exp_avg_sq) {
    set_error("drin_adam_step: NULL operand");
    return DRIN_E_NULL;
  }
  if (n < 0) {
    set_error("drin_adam_step: n=%lld", (long long)n);
    return DRIN_E_SHAPE;
  }
  if (n == 0) return DRIN_OK;
  if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq)) {
    set_error("drin_adam_step: operands must be 16-byte aligned");
    return DRIN_E_ALIGN;
  }
  if (!(lerp_weight > -0.5f && lerp_weight < 0.5f)) {
    // torch's lerp switches to b - (b - a)(1 - w) from |w| >= 0.5 (beta1 <= 0.5): not built
    set_error("drin_adam_step: 1 - beta1 = %g outside (-0.5, 0.5) is not built", (double)lerp_weight);
    return DRIN_E_UNSUPPORTED;
  }
  return launch_adam(param, grad, exp_avg, exp_avg_sq, n, lerp_weight, beta2, one_minus_beta2, bias_correction2_sqrt, eps,
                     neg_step_size, (hipStream_t)stream);
}
