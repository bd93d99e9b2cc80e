#!/bin/bash
# Run ON THE GPU BOX: where the wavefronts of the benchmark's kernels spend their cycles -- instruction-cache hit rate and the SQ wait /
# active counters, one rocprofv3 --pmc pass per group (with --kernel-trace only), summed per kernel.  Output: gpurun_out/sq_probe/summary.txt (profiles/r05_sq_wait_active.csv)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/sq_probe; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-host-legs --windows 1024 --steps 1 --warmup 1"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o p --output-format csv -- $B > $O/p$i.log 2>&1
done
python3 - <<'PY' > gpurun_out/sq_probe/summary.txt
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/sq_probe/p*/**/p_counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name'].split('(')[0][:44]][row['Counter_Name']] += float(row['Counter_Value'])
names = sorted({c for v in acc.values() for c in v})
print('kernel,' + ','.join(names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:10]:
    print('"' + k + '",' + ','.join(f"{v.get(c, 0):.4g}" for c in names))
PY
cat gpurun_out/sq_probe/summary.txt
