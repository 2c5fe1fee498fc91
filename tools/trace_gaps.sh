# kernel trace (start / end of every kernel) of a few bench steps; tools/trace_gaps.py reads the CSV
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_gaps -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/trace_gaps.json 2> $R/gpurun_out/trace_gaps.err
echo rc=$?
cd $R
f=$(find gpurun_out/trace_gaps -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f > gpurun_out/trace_gaps.txt
rm -rf gpurun_out/trace_gaps
