"""Run only the L/14@336 pattern leg of bench.py (for rocprofv3)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.l14_pattern_leg(torch.device("cuda:0"), None, steps=int(os.environ.get("STEPS", "2")))))
