#!/bin/bash
# build the stamped library, run the breakdown on the GPU box, rebuild the normal library
set -e
cd /root/repo/is-vins_amd/csrc
touch *.hip && make FLAGS="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -fgpu-rdc -Wall -Wno-unused-function -DISV_STAMP" 2>&1 | grep -E "error|warning" || true
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 600 -- 'python scripts/stamp_bs.py 512 > gpurun_out/stamp.log 2>&1; cat gpurun_out/stamp.log' 2>&1 | tail -12
cd /root/repo/is-vins_amd/csrc
touch *.hip && make 2>&1 | grep -E "error|warning" || true
