"""Profile target: the fine-tuning step on the FB15k237 shape (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import secondary_bench as sb
sb.train_case("fb15k237")
