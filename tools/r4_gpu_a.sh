#!/bin/bash
# round 4, GPU call A: the update beside the walk -- parity first, then times (forms, store policies, shares)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a
mkdir -p $O
timeout 120 python -m pytest tests/test_fused_update_gpu.py -x -q -k "beside_the_walk_repeats" > $O/first.txt 2>&1
rc=$?
tail -3 $O/first.txt
if [ $rc -eq 124 ]; then echo "HANG in the first test: stopping"; exit 1; fi
timeout 300 python -m pytest tests/test_fused_update_gpu.py tests/test_order_gpu.py -x -q > $O/tests_new.txt 2>&1
tail -5 $O/tests_new.txt
timeout 200 python tools/beside_probe.py fb15k237 8 > $O/probe_fb.txt 2>&1; cat $O/probe_fb.txt
for v in postdef postsc1; do
  ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_$v.so timeout 200 python tools/beside_probe.py fb15k237 8 > $O/probe_fb_$v.txt 2>&1
  echo "== $v"; grep -E "back to back|hipGraph|100 launches" $O/probe_fb_$v.txt
done
for sh in "1.7,1.3,0.7" "1.4,1.2,0.9" "1.2,1.1,1.0" "1.6,1.0,0.6"; do
  echo "== shares $sh"
  ULTRA_STREAM_SHARES_12="$sh" timeout 200 python tools/beside_probe.py fb15k237 8 2>&1 | grep -E "hipGraph|form 2|end of work|max over" | tail -4
done > $O/shares.txt 2>&1
cat $O/shares.txt
timeout 200 python tools/beside_probe.py codex_l 8 max > $O/probe_codex_max.txt 2>&1; grep -E "==|back to back|hipGraph|100 launches" $O/probe_codex_max.txt
timeout 200 python tools/beside_probe.py codex_l 8 add > $O/probe_codex_add.txt 2>&1; grep -E "==|back to back|hipGraph|100 launches" $O/probe_codex_add.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -5 $O/tests_all.txt
