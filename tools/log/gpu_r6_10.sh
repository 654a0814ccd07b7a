#!/bin/bash
# round 6: what runs when in the three-in-flight step (kernel trace with timestamps), relation-graph layer forms 0 and 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lean in 0 2; do
  rm -rf /tmp/pt_$lean
  ULTRA_DOL_LEAN=$lean PROBE_DEPTH=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_$lean -- python tools/step_probe.py 3 60 > gpurun_out/r6_10_probe_$lean.txt 2>&1
  f=$(ls /tmp/pt_$lean/*/*_kernel_trace.csv | head -1)
  python tools/pipeline_timeline.py $f 2.4 --list > gpurun_out/r6_10_pipeline_$lean.txt 2>&1
  grep -v amdgpu gpurun_out/r6_10_probe_$lean.txt | tail -2
  tail -16 gpurun_out/r6_10_pipeline_$lean.txt
done
