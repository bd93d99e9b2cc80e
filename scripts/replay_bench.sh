#!/bin/bash
# Run ON THE GPU BOX: whole-sequence throughput of tools/isv_replay (native: stream -> window manager -> MI355X).
# usage: bash scripts/replay_bench.sh [N Nvo n_frames]   -> gpurun_out/replay_bench.log
set -e
N=${1:-11}; NVO=${2:-5}; NF=${3:-36}
mkdir -p gpurun_out/replay
python3 - <<PY
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import isvins_loader; isvins_loader.load()
import sequence_harness as sh
sh.write_stream("gpurun_out/replay/stream_${N}.txt", $N, $NVO, $NF, seed=1)
PY
for cfg in "1 1" "64 1" "256 1" "256 2" "256 4" "512 4" "1024 4"; do
  set -- $cfg
  ./tools/isv_replay gpurun_out/replay/stream_${N}.txt --sequences $1 --groups $2
done
