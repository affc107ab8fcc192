"""bench.py's real-terrain leg alone (for rocprofv3 --kernel-trace --stats):
   python3 tools/real_terrain_run.py [rep] [rough_n]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rough_n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
print(json.dumps(bench.real_terrain(rep, rough_n), indent=1))
