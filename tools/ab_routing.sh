#!/bin/bash
# Same-box A/B of the round-4 GEMM changes over the default bench (and any other bench command line given after the round count):
#   A  the product library                                                 B  + P8 for weight gradients with a fused bias gradient and for the logits with row statistics
#   C  round-3 kernel choice (-DESVIT_NO_P8_ROUTING)                       D  product routing, non-temporal output stores
# Usage: bash tools/ab_routing.sh [build | rounds [bench args]]
set -e
root=$(cd "$(dirname "$0")/.." && pwd); cd $root
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-pass-failed -I include -I esvit_amd/csrc -x hip -c"
build_variant() {  # name, defines
  out=$root/tools/probe/libesvit_hip_$1.so
  if [ -f $out ] && [ ! esvit_amd/csrc/gemm.hip -nt $out ] && [ ! esvit_amd/csrc/gemm_kernels.h -nt $out ] && [ ! esvit_amd/csrc/gemm_p8.hip -nt $out ]; then return; fi
  objs=$(ls esvit_amd/csrc/build/*.o | grep -v "/gemm.o\|/gemm_p8.o")
  for f in gemm gemm_p8; do /opt/rocm/bin/hipcc $flags $2 esvit_amd/csrc/$f.hip -o /tmp/${f}_$1.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs /tmp/gemm_$1.o /tmp/gemm_p8_$1.o
  echo built $out
}
build_variant B "-DESVIT_P8_WGRAD_COLSUM -DESVIT_P8_ROWSTAT_AUTO"
build_variant C "-DESVIT_NO_P8_ROUTING"
build_variant D "-DESVIT_NT_STORES"
[ "$1" = "build" ] && exit 0
rounds=${1:-3}; shift || true
ms() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for r in $(seq 1 $rounds); do
  a=$(ms "$@")
  b=$(ESVIT_HIP_LIB=$root/tools/probe/libesvit_hip_B.so ms "$@")
  c=$(ESVIT_HIP_LIB=$root/tools/probe/libesvit_hip_C.so ESVIT_NO_P8_ROUTING=1 ms "$@")
  d=$(ESVIT_HIP_LIB=$root/tools/probe/libesvit_hip_D.so ms "$@")
  echo "round $r: A product $a   B + colsum wgrads, rowstat $b   C r3-choice $c   D product+nt $d   (ms per step)"
done
