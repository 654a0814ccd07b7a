#!/bin/bash
# round 6: where the fused update backward's time goes -- timing-only switches (wrong results): 256 no dW matrix instructions,
# 512 none of d[x;agg], 1024 none of the recompute
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 0 256 512 1024 768 1792; do
  echo "PROBE_FLAGS=$f"; PROBE_FLAGS=$f timeout 300 python tools/conv_bwd_probe.py 116328 985456 2>&1 | grep -v amdgpu
done
