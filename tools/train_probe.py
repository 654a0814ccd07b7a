"""Profile target: the fine-tuning step on the FB15k237 shape (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import secondary_bench as sb
import json
import sys as _sys
print(json.dumps(sb.train_case(_sys.argv[1] if len(_sys.argv) > 1 else "fb15k237")))
