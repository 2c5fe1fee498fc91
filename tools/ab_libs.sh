#!/bin/bash
# A/B of two BUILDS of the kernel library on one box: bench.py (no CPU baseline, no roofline legs) alternating between
# ab_ref/libstonk_hip.so (tools/build_ref_lib.sh) and the in-tree library, ROUNDS times each; prints ms per step.
rounds=${1:-3}
steps=${2:-40}
for i in $(seq $rounds); do
  for which in ref new; do
    if [ $which = ref ]; then export STONK_HIP_LIB=ab_ref/libstonk_hip.so; else unset STONK_HIP_LIB; fi
    out=$(timeout -k 10 300 python bench.py --steps $steps --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "$which $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
