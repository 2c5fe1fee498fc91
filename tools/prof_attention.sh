#!/bin/bash
# per-kernel attention timings alone (rocprofv3 kernel stats over tools/bench_attention.py): bash tools/prof_attention.sh <out-subdir> [P=<dropout>]
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -- python3 $GRAFT_REPO_ROOT/tools/bench_attention.py > /dev/null 2>&1; grep "attn_" $GRAFT_REPO_ROOT/gpurun_out/$1/*/*kernel_stats.csv | cut -d, -f1-4
