#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the rocprofv3 evidence of one round, everything under gpurun_out/prof_rNN/; copy the
# summaries into profiles/ afterwards (scripts/copy_profiles.sh).  --pmc passes are separate runs with --kernel-trace only.
# usage: bash scripts/refresh_profiles.sh [round prefix, default r05] [windows, default 1024] [part: A = steps 1-5, B = steps 6-11, default both]
# (one gpurun call is limited to 20 minutes: the two parts are two calls)
set -e
RN=${1:-r05}; W=${2:-1024}; PART=${3:-AB}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$RN
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-host-legs --windows $W"
if [[ $PART == *A* ]]; then
# 1. kernel-trace stats of the bench command (the launches of the benchmark only)
rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- $B --steps 5 --warmup 2 > $O/bench_under_rocprof.json 2> $O/bench.err
cp $O/bench/bench_kernel_stats.csv $O/${RN}_bench_${W}win_kernel_stats.csv
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python3 scripts/pmc_summary.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $W $O $RN > $O/pmc_summary.log
# 3. utilisation counters, one pass per group
i=0
for grp in "VALUBusy MfmaUtil" "LdsUtil LDSBankConflict" "MemUnitStalled OccupancyPercent"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_u$i -o u --output-format csv -- $B --steps 1 --warmup 1 > $O/pmc_u$i.log 2>&1
done
python3 scripts/pmc_util_summary.py $O/pmc_u*/u_counter_collection.csv > $O/${RN}_pmc_utilisation.csv
# 4. the reference's shape: 1024 windows of N = 18 / Vo = 8
REPS=5 rocprofv3 --kernel-trace --stats -d $O/n18 -o n18 --output-format csv -- python3 scripts/quick_cfg.py 1024 18 8 300 > $O/n18.log 2>&1
cp $O/n18/n18_kernel_stats.csv $O/${RN}_n18_1024win_kernel_stats.csv
# 5. BASELINE config 5: one window, 20 KF / 2000 landmarks / 30 000 factors
REPS=10 rocprofv3 --kernel-trace --stats -d $O/cfg5 -o cfg5 --output-format csv -- python3 scripts/quick_cfg.py 1 20 8 2000 30000 > $O/cfg5.log 2>&1
cp $O/cfg5/cfg5_kernel_stats.csv $O/${RN}_config5_kernel_stats.csv
REPS=3 rocprofv3 --kernel-trace --pmc VALUBusy MfmaUtil -d $O/cfg5_u -o u --output-format csv -- python3 scripts/quick_cfg.py 1 20 8 2000 30000 > $O/cfg5_u.log 2>&1
python3 scripts/pmc_util_summary.py $O/cfg5_u/u_counter_collection.csv > $O/${RN}_config5_pmc_utilisation.csv
fi
if [[ $PART == *B* ]]; then
# 6. the pose-graph kernel: one graph and 1024 graphs of 200 keyframes
PGO_CFGS=200:5:1,200:5:1024 PGO_NO_ORACLE=1 rocprofv3 --kernel-trace --stats -d $O/pgo -o pgo --output-format csv -- python3 scripts/pgo_bench.py > $O/pgo.log 2>&1
cp $O/pgo/pgo_kernel_stats.csv $O/${RN}_pgo_kernel_stats.csv
# 7. the resident replay's kernels (slide / append / build beside the solve): 256 sequences
python3 -c "
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import isvins_loader; isvins_loader.load()
import sequence_harness as sh
sh.write_stream('$O/stream_11.txt', 11, 5, 40, seed=1)"
rocprofv3 --kernel-trace --stats -d $O/replay -o replay --output-format csv -- ./tools/isv_replay $O/stream_11.txt --sequences 256 --groups 1 --write 0 > $O/replay.log 2>&1
cp $O/replay/replay_kernel_stats.csv $O/${RN}_resident_replay_256seq_kernel_stats.csv
# 8. the bench line itself and the two-rank rehearsal of the multi-GPU path on this one GPU (gloo stands in for RCCL);
#    only the JSON line goes into the artefact (gloo's connection messages used to land in the .json)
python3 bench.py > $O/bench_plain.out 2> $O/bench_plain.err
grep '^{' $O/bench_plain.out | tail -1 > $O/${RN}_bench_1gpu.json
ISV_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --windows 512 --no-host-legs --no-cpu-baseline --steps 50 > $O/rehearsal.out 2> $O/rehearsal.err
grep '^{' $O/rehearsal.out | tail -1 > $O/${RN}_2rank_rehearsal.json
# 9. kernel boundary against grid barrier (what a persistent single-window kernel would have to beat) and the launch chain of one window
hipcc -O3 --offload-arch=gfx950 -o /tmp/launch_vs_barrier scripts/launch_vs_barrier.hip && timeout -k 10 120 /tmp/launch_vs_barrier > $O/${RN}_launch_vs_barrier.json
REPS=6 rocprofv3 --kernel-trace --stats -d $O/one11 -o one --output-format csv -- python3 scripts/quick_cfg.py 1 11 5 300 > $O/one11.log 2>&1
cp $O/one11/one_kernel_stats.csv $O/${RN}_single_window_n11_kernel_stats.csv
REPS=6 rocprofv3 --kernel-trace --stats -d $O/one18 -o one --output-format csv -- python3 scripts/quick_cfg.py 1 18 8 300 > $O/one18.log 2>&1
cp $O/one18/one_kernel_stats.csv $O/${RN}_single_window_n18_kernel_stats.csv
# 10. every kernel ALONE (one stream): what the side-stream kernels cost without queueing behind k_lin_gram
ISV_ONE_STREAM=1 rocprofv3 --kernel-trace --stats -d $O/os11 -o os --output-format csv -- $B --steps 5 --warmup 2 > $O/os11.log 2>&1
cp $O/os11/os_kernel_stats.csv $O/${RN}_one_stream_n11_kernel_stats.csv
ISV_ONE_STREAM=1 rocprofv3 --kernel-trace --stats -d $O/os18 -o os --output-format csv -- $B --steps 5 --warmup 2 --frames 18 --vo 8 > $O/os18.log 2>&1
cp $O/os18/os_kernel_stats.csv $O/${RN}_one_stream_n18_kernel_stats.csv
# 11. (round 5) the single-stream launch chain of a small batch: timelines of one window (11 and 18 frames) and the kernels of the 128-window shard
N=11 V=5 rocprofv3 --kernel-trace --output-format csv -d $O/tl11 -o tl -- python3 scripts/b1_trace.py > $O/tl11.log 2>&1
python3 scripts/timeline.py $(find $O/tl11 -name "*kernel_trace.csv" | head -1) 40 > $O/${RN}_single_window_n11_timeline.txt
N=18 V=8 rocprofv3 --kernel-trace --output-format csv -d $O/tl18 -o tl -- python3 scripts/b1_trace.py > $O/tl18.log 2>&1
python3 scripts/timeline.py $(find $O/tl18 -name "*kernel_trace.csv" | head -1) 40 > $O/${RN}_single_window_n18_timeline.txt
REPS=6 rocprofv3 --kernel-trace --stats -d $O/b128 -o b128 --output-format csv -- python3 scripts/quick_cfg.py 128 11 5 300 > $O/b128.log 2>&1
cp $O/b128/b128_kernel_stats.csv $O/${RN}_shard_128win_kernel_stats.csv
fi
ls $O/*.csv $O/*.json
head -12 $O/${RN}_bench_${W}win_kernel_stats.csv
