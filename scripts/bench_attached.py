"""Only the attached_reference legs (bench.py's `attached_reference_leg`) of the five configurations, one JSON line each:
python scripts/bench_attached.py [config ...]"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

names = sys.argv[1:] or ["balance", "transport", "transport_2pkg", "navigation", "football"]
for name in names:
    cfg = bench.CONFIGS[name]
    try:
        out = bench.attached_reference_leg(name, bench.config_kwargs(name), int(os.environ.get("ENVS", cfg["envs"])), torch.device("cuda:0"),
                                           n=int(os.environ.get("N", "300")), brief=bool(os.environ.get("BRIEF")))
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        out = {"error": repr(e)[:400]}
    out["config"] = name
    print(json.dumps(out), flush=True)
