#!/bin/bash
# builds tools/probe/libgemm_probe.so (experimental GEMM main loops for tools/bench_gemm.py; not shipped in the product library)
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -I "$root/include" -I "$root/esvit_amd/csrc" \
    -x hip "$here/gemm_probe.hip" -o "$here/libgemm_probe.so"
echo built "$here/libgemm_probe.so"
# ablations of the default main loop for tools/gemm_timeline.py --lib: without the MFMAs / without MFMAs and fragment reads
if [ "$1" = "ablate" ]; then
  for v in SKIP_MFMA "SKIP_MFMA -DESVIT_PROBE_SKIP_FRAG" A_RESIDENT; do
    n=$(echo $v | tr -d ' ' | sed 's/-DESVIT_PROBE_/_/' | tr 'A-Z' 'a-z')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -DESVIT_PROBE_ONLY7 -DESVIT_PROBE_$v -I "$root/include" -I "$root/esvit_amd/csrc" \
        -x hip "$here/gemm_probe.hip" -o "$here/libgemm_probe_$n.so"
    echo built "$here/libgemm_probe_$n.so"
  done
fi
# the eight-phase loop with per-item timestamps (tools/p8_timeline.py)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-pass-failed -DESVIT_P8_TIMELINE -I "$root/include" -I "$root/esvit_amd/csrc" \
    -x hip "$root/esvit_amd/csrc/gemm_p8.hip" -o "$here/libp8_probe.so"
echo built "$here/libp8_probe.so"
# the 256 x 128 two-set loop (tools/probe/gemm_p8n.hip: measured in round 4, routed to nothing, not part of the product library since
# round 5) with per-item timestamps, and the same without its epilogue stores (tools/p8_timeline.py --lib)
for v in "" "-DESVIT_P8N_NOSTORE"; do
  n=$( [ -z "$v" ] && echo libp8n_probe.so || echo libp8n_probe_nostore.so )
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-pass-failed -DESVIT_P8_TIMELINE $v -I "$root/include" -I "$root/esvit_amd/csrc" \
      -x hip "$here/gemm_p8n.hip" -o "$here/$n"
  echo built "$here/$n"
done
