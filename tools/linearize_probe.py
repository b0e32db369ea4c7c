"""Development probe: the linearisation stage alone on the bench's C5-sized graph (run under rocprofv3 --kernel-trace
--stats for the per-kernel split)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.linearize_timing()))
