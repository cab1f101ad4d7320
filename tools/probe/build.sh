#!/bin/bash
# builds tools/probe/libgemm_probe.so (experimental GEMM main loops for tools/bench_gemm.py; not shipped in the product library)
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -I "$root/include" -I "$root/esvit_amd/csrc" \
    -x hip "$here/gemm_probe.hip" -o "$here/libgemm_probe.so"
echo built "$here/libgemm_probe.so"
