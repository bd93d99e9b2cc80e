#!/bin/bash
# Run ON THE GPU BOX: whole-sequence throughput of tools/isv_replay (native: stream -> window manager -> MI355X), with
# the windows resident on the device (default) and re-uploaded every frame (--no-resident).
# usage: bash scripts/replay_bench.sh [N Nvo n_frames]   -> stdout (JSON lines)
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
for cfg in "1 1" "256 1" "1024 1" "1024 2" "1024 4" "2048 2" "2048 4" "4096 4" "4096 8"; do
  set -- $cfg
  ./tools/isv_replay gpurun_out/replay/stream_${N}.txt --sequences $1 --groups $2
done
for cfg in "1 1" "1024 4" "2048 4"; do
  set -- $cfg
  ./tools/isv_replay gpurun_out/replay/stream_${N}.txt --sequences $1 --groups $2 --no-resident
done
