#!/bin/bash
# builds tools/probe/libesvit_ab_timeline.so: the product library with the fused attention branch compiled with per-phase cycle
# stamps (-DESVIT_AB_TIMELINE; tools/attn_branch_timeline.py).  Needs the product objects (python -m esvit_amd.build) first.
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
obj=$(mktemp -d)/attn_branch_tl.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -DESVIT_AB_TIMELINE $AB_EXTRA -I "$root/include" -I "$root/esvit_amd/csrc" \
    -x hip -c "$root/esvit_amd/csrc/attn_branch.hip" -o "$obj"
others=$(ls "$root"/esvit_amd/csrc/build/*.o | grep -v attn_branch.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/${AB_OUT:-libesvit_ab_timeline.so}" $others "$obj"
echo built "$here/${AB_OUT:-libesvit_ab_timeline.so}"
