#!/bin/bash
# Run GPU steps one after another on a gpurun box; a step that was killed by its time limit (124 / 137) ends the call -
# no further GPU step is started after a kill. Usage: tools/gpu_steps.sh LOGPREFIX "cmd1" "cmd2" ...
mkdir -p gpurun_out
prefix=$1; shift
i=0
for cmd in "$@"; do
  i=$((i+1))
  log="gpurun_out/${prefix}_s${i}.log"
  echo "[gpu_steps] step $i: $cmd" | tee "$log"
  bash -c "$cmd" >> "$log" 2>&1
  rc=$?
  echo "[gpu_steps] step $i rc=$rc" | tee -a "$log"
  tail -n 6 "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[gpu_steps] step $i was killed: stopping"; exit $rc; fi
done
exit 0
