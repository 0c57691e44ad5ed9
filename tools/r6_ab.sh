#!/bin/bash
# Round 6 same-box A/Bs (in process, one resident batch each: tools/probes/inprocess_lib_ab.py):
#  1. the stream kernel's next-pair prefetch (candidate row + mask word) against -DDRIN_STREAM_PREFETCH=0: fp32 rows, bf16-stored rows, WikiDiverse
#  2. the bound of a GEMM + row-kernel fusion: -DDRIN_ABLATE_ROW_TRAFFIC -DDRIN_ABLATE_GEMM_STORES (wrong results, timing only): the row
#     kernels with NO HBM traffic (four cache-resident rows, nothing stored) and the pair-sized contractions WITHOUT their [M, N] stores
O=gpurun_out/r6_ab
rm -rf $O; mkdir -p $O
python tools/probes/inprocess_lib_ab.py drin_amd/libdrin_hip_noprefetch.so 3 > $O/prefetch_f32.txt 2>&1
python tools/probes/inprocess_lib_ab.py drin_amd/libdrin_hip_noprefetch.so 3 --features bf16 > $O/prefetch_bf16.txt 2>&1
python tools/probes/inprocess_lib_ab.py drin_amd/libdrin_hip_noprefetch.so 3 --workload wikidiverse > $O/prefetch_wd.txt 2>&1
python tools/probes/inprocess_lib_ab.py drin_amd/libdrin_hip_ablate.so 3 > $O/ablate_f32.txt 2>&1
for f in prefetch_f32 prefetch_bf16 prefetch_wd ablate_f32; do echo "== $f (shipped = HEAD, other = the variant)"; grep -v amdgpu.ids $O/$f.txt; done
