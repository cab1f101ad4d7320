#!/bin/bash
# one-rank RCCL reducer path (ESVIT_FORCE_REDUCER=1 under torch.distributed.run) with env arms: bash tools/probe/ab_rccl1.sh ROUNDS arm1 arm2 ...  ("-" = default)
rounds=$1; shift
port=29600
for r in $(seq 1 $rounds); do
  line="round $r:"
  for a in "$@"; do
    port=$((port+1))
    e=""; [ "$a" != "-" ] && e="$a"
    v=$(env $e ESVIT_FORCE_REDUCER=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    line="$line  [$a] $v"
  done
  echo "$line"
done
