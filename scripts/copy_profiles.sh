#!/bin/bash
# after scripts/refresh_profiles.sh ran on the GPU box: the summaries that are judged go from scratch into profiles/
RN=${1:-r03}
cp gpurun_out/prof_$RN/${RN}_*.csv gpurun_out/prof_$RN/${RN}_*.json profiles/
ls profiles/${RN}_*
