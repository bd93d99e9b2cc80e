#!/bin/bash
# Run ON THE GPU BOX: utilisation counters of the bench's kernels, one rocprofv3 pass per counter group
# (--pmc only with --kernel-trace), summarised per kernel by scripts/pmc_util_summary.py.
set -e
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_u*
W=${1:-1024}
i=0
for grp in "VALUBusy MfmaUtil" "LdsUtil LDSBankConflict" "MemUnitStalled OccupancyPercent" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pmc_u$i -o u --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-legs --windows $W > gpurun_out/pmc_u$i.log 2>&1
done
python3 scripts/pmc_util_summary.py gpurun_out/pmc_u*/u_counter_collection.csv > gpurun_out/r02_pmc_utilisation.csv
cat gpurun_out/r02_pmc_utilisation.csv
