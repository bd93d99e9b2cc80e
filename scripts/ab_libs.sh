#!/bin/bash
# A/B of prebuilt library variants under scratch_libs/ (run on the GPU box): scripts/ab_libs.sh out.log "64 1024" base maxilp ...
out=$1; sizes=$2; shift 2
for v in "$@"; do cp scratch_libs/lib_$v.so is-vins_amd/csrc/libisvins_hip.so; for W in $sizes; do echo "== $v W=$W" >> $out; python bench.py --windows $W --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['ms_per_optimize_single_window'] or 0,3), {k[:-4]:round(v,3) for k,v in d['kernel_ms'].items()})" >> $out; done; done
cat $out
