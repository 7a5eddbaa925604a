"""f64 MFMA issue-rate / clock probe (v_mfma_f64_16x16x4_f64, register-only) at 1..8 workgroups per CU."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi

ctx = hipabi.Context(0)
print(ctx.device_name())
for b in (1, 2, 4, 8):
    print(json.dumps(ctx.probe_f64_mfma_detail(b)))
