#!/bin/bash
# one-stream kernel traces of the benchmark for diagnostic builds of k_lin_gram (scratch_libs/lib_<v>.so) -- run on the GPU box.
# The builds behind profiles/r05_lin_gram_anatomy.txt were the library compiled with temporary -DLGX_* / -DLTX_* switches that took one
# part of the kernel out (MFMA, global stores, the evaluation, the Gram rounds, the group hand-over); the switches are not in the tree.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lgv; mkdir -p $out; export ISV_ONE_STREAM=1
run() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $out/$name -o t -- python bench.py --no-cpu-baseline --no-host-legs --steps 3 --warmup 1 > $out/$name.log 2>&1; }
[ -n "$SKIP_BASE" ] || run base
true
for v in "$@"; do ISVINS_LIB=$PWD/scratch_libs/lib_$v.so run $v; done
python scripts/lg_trace.py $out/base $out/lean $(for v in "$@"; do echo $out/$v; done) | tee $out/summary.txt
