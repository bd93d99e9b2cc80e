#!/bin/bash
# k_build_solve_st variants (scratch_libs/lib_<v>.so, built with -DST_* switches): scripts/st_probe.py (the streamed kernel forced on
# small handles, every N <= 13 against k_build_solve_sb) + the 1024-window bench twice per variant; run on the GPU box
out=gpurun_out/st_ab.log; : > $out
for v in "$@"; do
  cp scratch_libs/lib_$v.so is-vins_amd/csrc/libisvins_hip.so
  echo "== $v" >> $out
  timeout -k 10 120 python scripts/st_probe.py 2>&1 | grep "^N=" | awk '{print $1, $3, $NF}' | tr '\n' ';' >> $out; echo >> $out
  for i in 1 2; do python bench.py --windows 1024 --steps 100 --warmup 5 --no-cpu-baseline --no-host-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k[:-4]:round(v,3) for k,v in d['kernel_ms'].items()})" >> $out; done
done
cat $out
