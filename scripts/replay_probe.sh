#!/bin/bash
# resident replay probe: the bench's device_resident_replay legs with the per-step breakdown (scratch output under gpurun_out/)
# usage: bash scripts/replay_probe.sh ["S K" ...]   (default: 2048/4, 2048/2, 2048/1, 4096/8, 4096/4)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/replay
python - <<PY
import sys, os
sys.path.insert(0, "tests")
import isvins_loader; isvins_loader.load()
import sequence_harness as sh
sh.write_stream("gpurun_out/replay/stream.txt", 11, 5, 48, seed=1)
PY
if [ $# -eq 0 ]; then set -- "2048 4" "2048 2" "2048 1" "4096 8" "4096 4"; fi
for cfg in "$@"; do
  set -- $cfg
  ./tools/isv_replay gpurun_out/replay/stream.txt --sequences $1 --groups $2 --out gpurun_out/replay --write 0 | tail -1
done
