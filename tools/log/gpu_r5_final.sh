#!/bin/bash
# (one gpurun call, round 5) the round's tracked artefacts: GPU suite, tools/collect_profiles.sh A / B / C, form-3 probes of every
# shape, the launcher's rank with two and three batches in flight.  Everything lands in gpurun_out/r5/ (copied to profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/r5_gpu_tests.txt 2>&1
tail -3 $O/r5_gpu_tests.txt
bash tools/collect_profiles.sh A r5 > $O/collect_A.log 2>&1
bash tools/collect_profiles.sh B r5 > $O/collect_B.log 2>&1
bash tools/collect_profiles.sh C r5 > $O/collect_C.log 2>&1
{
echo "# tools/form3_probe.py [shape] [batch] [sum]: the one-launch entity layer (form 3) in a hipGraph of 20 + one traced launch"
timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
PROBE_GRID=192 timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py fb15k237 8 max 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py fb15k237 16 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py fb15k237 4 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py codex_l 8 max 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py wn18rr 8 2>&1 | grep -v amdgpu.ids | tail -1
echo "# tools/beside_probe.py fb15k237 8"
timeout 300 python tools/beside_probe.py fb15k237 8 2>&1 | grep -v amdgpu.ids
echo "# tools/step_probe.py 9 40"
timeout 300 python tools/step_probe.py 9 40 2>&1 | grep -v amdgpu.ids
} > $O/r5_form3_probe.txt 2>&1
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'], 'slot streams', d['config'].get('slot_streams'))"; }
{
for k in 2 3 2 3; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 20 --warmup 5 --in-flight $k --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "torchrun, 1 rank, --in-flight $k"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "plain bench.py"
} > $O/r5_launcher_path.txt 2>&1
cat $O/r5_launcher_path.txt
ls $O
