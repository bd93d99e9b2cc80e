#!/bin/bash
# rocprofv3 kernel statistics of one quick_cfg.py configuration (GPU box): scripts/kstats.sh "B N Nvo L" out_prefix
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$2 -o ks -- python scripts/quick_cfg.py $1 > gpurun_out/ks_$2.log 2>&1
f=$(find gpurun_out/ks_$2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/ks_$2.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.3f} ms")
PY
rm -rf gpurun_out/ks_$2
