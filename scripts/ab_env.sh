#!/bin/bash
# A/B of an environment switch on the GPU box: scripts/ab_env.sh VAR "cfg1;cfg2;..."  (cfg = quick_cfg.py arguments)
# prints optimize ms per batch with VAR=0 and with VAR unset
var=$1; IFS=';' read -ra cfgs <<< "$2"
for c in "${cfgs[@]}"; do
  echo "== $c  ($var=0)"; env $var=0 REPS=20 python scripts/quick_cfg.py $c
  echo "== $c  (default)"; REPS=20 python scripts/quick_cfg.py $c
done
