# rocprofv3 per-kernel summary of the bench (kernel-trace + stats): TAG names the output files under gpurun_out/
set -x
TAG=${TAG:-r4}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kstats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_kstats.err
echo rc=$?
