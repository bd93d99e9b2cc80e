#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats of the bench command, then the two PMC passes (separate runs,
# --pmc only with --kernel-trace), everything under gpurun_out/; copy the summaries into profiles/ afterwards.
set -e
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench gpurun_out/pmc_fetch gpurun_out/pmc_write
W=${1:-1024}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-legs --windows $W > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --windows $W > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --windows $W > gpurun_out/pmc_write.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv $W gpurun_out r02 > gpurun_out/pmc_summary.log
head -30 gpurun_out/prof_bench/bench_kernel_stats.csv
