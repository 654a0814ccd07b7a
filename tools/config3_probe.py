"""Profile target: BASELINE config 3 (ultra_50g weights, CoDEx-L shape, max aggregate, batch 8, all-tail), eager launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import json
import secondary_bench as sb
print(json.dumps(sb.forward_case("codex_l", sys.argv[1] if len(sys.argv) > 1 else "max", "ultra_50g")))
