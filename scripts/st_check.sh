#!/bin/bash
# the in-tree library: streamed-solve tests, st_probe, a short bench; then the stamped variant's phase table (run on the GPU box)
out=gpurun_out/st_check.log; : > $out
python -m pytest tests/test_gpu_branches.py -k "streamed" -q 2>&1 | tail -4 >> $out
timeout -k 10 120 python scripts/st_probe.py 2>&1 | grep "^N=" >> $out
for i in 1 2; do python bench.py --windows 1024 --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k[:-4]:round(v,3) for k,v in d['kernel_ms'].items()})" >> $out; done
if [ -n "$1" ]; then
cp scratch_libs/lib_$1.so is-vins_amd/csrc/libisvins_hip.so
echo "== stamps ($1) B=1024" >> $out
timeout -k 10 200 python scripts/stamp_bs.py 1024 2>&1 | grep -A 12 "per k_build_solve_sb call" >> $out
fi
cat $out
