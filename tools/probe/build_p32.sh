#!/bin/bash
# builds tools/probe/libp32_probe*.so: the wide fused MLP kernel (esvit_amd/csrc/mlp_fused32p.hip) with per-iteration timestamps and, per
# variant, one ablation of the iteration body (tools/p32_timeline.py --lib)
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
for v in "" STAMPS NO_SIDE_STORE SIDE_LINEAR NO_GELU NO_G1 NO_G2 NO_DMA_WAIT NO_DMA "NO_GELU -DP32_NO_DMA" "NO_G1 -DP32_NO_G2" "NO_G1 -DP32_NO_G2 -DP32_NO_GELU" "NO_G1 -DP32_NO_G2 -DP32_NO_GELU -DP32_NO_DMA" $P32_EXTRA_VARIANTS; do
  n=$(echo "$v" | sed 's/-DP32_//g' | tr -d ' ' | tr 'A-Z' 'a-z')
  f=$( [ -z "$v" ] && echo "" || echo "-DP32_$v" )
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-pass-failed -DESVIT_P32_PROBE $f $P32_CFLAGS -I "$root/include" -I "$root/esvit_amd/csrc" \
      -x hip "$root/esvit_amd/csrc/mlp_fused32p.hip" -o "$here/libp32_probe${n:+_$n}.so"
  echo built "$here/libp32_probe${n:+_$n}.so"
done
