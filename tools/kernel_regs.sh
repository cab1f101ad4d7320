#!/bin/bash
# VGPR / scratch / spill table of one translation unit's gfx950 kernels: bash tools/kernel_regs.sh gemm [name filter]
o=esvit_amd/csrc/build/$1.o; t=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $o $t/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/fat.bin --output=$t/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/k.co | python3 -c "
import re, sys
cur = {}
def flush():
    if 'name' in cur:
        print('%4s vgpr %4s agpr %5s scratch %4s spill  %s' % (cur.get('vgpr_count'), cur.get('agpr_count'), cur.get('private_segment_fixed_size'), cur.get('vgpr_spill_count'), cur['name'].replace('_ZN12_GLOBAL__N_1', '')[:120]))
for line in sys.stdin:
    m = re.match(r'\s+\.(name|vgpr_count|agpr_count|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)', line)
    if not m: continue
    if m.group(1) == 'name' and 'name' in cur and 'vgpr_count' in cur:
        flush(); cur.clear()
    cur[m.group(1)] = m.group(2)
flush()
" | grep "${2:-.}"
rm -rf $t
