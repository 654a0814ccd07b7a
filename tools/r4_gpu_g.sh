#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/collect_profiles.sh A r4 > /dev/null 2>&1
bash tools/collect_profiles.sh B r4 > /dev/null 2>&1
bash tools/collect_profiles.sh C r4 > /dev/null 2>&1
ls -la gpurun_out/r4
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/r4_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "repeats")})
print([ (c["shape"], c["aggregate"], round(c["ms_per_forward"],3), c["parity"]["bit_equal"]) for c in d["secondary"]["forward"]])
print([ (c["shape"], c["aggregate"], round(c["ms_per_step"],2)) for c in d["secondary"]["fine_tune"]])
print(d["modes"]["one_batch_in_flight"]["ms_per_step"], d["modes"]["two_launch_layers"]["ms_per_step"])
PY
tail -3 gpurun_out/r4/bench.err
