# PMC passes over the bench (separate passes: FETCH_SIZE and WRITE_SIZE cannot share one on gfx950; SQ counters third)
set -x
TAG=${TAG:-r4}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc_fetch --output-format csv -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch.err; echo rc=$?
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc_write --output-format csv -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write.err; echo rc=$?
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq --output-format csv -- $B > /dev/null 2> $R/gpurun_out/${TAG}_pmc_sq.err; echo rc=$?
cd $R
python tools/summarize_pmc.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write 3 > gpurun_out/${TAG}_pmc_traffic.csv
python tools/summarize_pmc_sq.py gpurun_out/${TAG}_pmc_sq 3 > gpurun_out/${TAG}_pmc_mfma.csv
rm -rf gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_sq
