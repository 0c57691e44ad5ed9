#!/bin/bash
# Same-box A/B of the GEMM kernels: current build against drin_amd/libdrin_hip_prev.so (tools/gemm_bench.py), 2 alternating runs,
# then the planes-kernel parity tests on the current build.
for i in 1 2; do
  echo "== new $i"; python tools/gemm_bench.py 103424 2>&1 | grep -v amdgpu.ids | grep "planes\|bf16x3"
  echo "== prev $i"; DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_prev.so python tools/gemm_bench.py 103424 2>&1 | grep -v amdgpu.ids | grep "planes\|bf16x3"
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "planes or fused or bf16x3 or golden" 2>&1 | tail -3
