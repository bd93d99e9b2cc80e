#!/bin/bash
# bit-level and timing A/B of the in-tree library against scratch_libs/lib_head.so (a build of another commit: git archive <rev> | tar -x -C /tmp/x;
# make -C /tmp/x/is-vins_amd/csrc libisvins_hip.so) -- GPU box.  Used for the two k_lin_gram bodies of profiles/r05_lin_gram_anatomy.txt.
out=gpurun_out/lgab; mkdir -p $out
timeout -k 10 300 python scripts/ab_bits.py > $out/bits_new.txt 2>&1
ISVINS_LIB=$PWD/scratch_libs/lib_head.so timeout -k 10 300 python scripts/ab_bits.py > $out/bits_head.txt 2>&1
diff $out/bits_new.txt $out/bits_head.txt > $out/bits_diff.txt && echo BITS_SAME | tee -a $out/bits_diff.txt
for mode in two one; do
  [ $mode = one ] && export ISV_ONE_STREAM=1
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 > $out/bench_${mode}_new.json 2> $out/bench_${mode}_new.err
  ISVINS_LIB=$PWD/scratch_libs/lib_head.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 > $out/bench_${mode}_head.json 2> $out/bench_${mode}_head.err
done
python - <<'PY' | tee $out/summary.txt
import json
for mode in ('two','one'):
    for n in ('new','head'):
        try:
            j=json.loads(open(f'gpurun_out/lgab/bench_{mode}_{n}.json').read().strip().split('\n')[-1])
            k=j['kernel_ms']
            print(mode, n, round(j['value']), round(j['ms_per_step'],3), 'single', j.get('ms_per_optimize_single_window'), 'shard', (j.get('strong_scaling_shard') or {}).get('ms_per_step'), {a[:-4]:round(b,3) for a,b in k.items() if a.endswith('_sum')})
        except Exception as e: print(mode, n, 'failed', e)
PY
cat $out/bits_diff.txt
