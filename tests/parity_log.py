"""Parity tests record their worst errors / alive / fork counts here so that the numbers quoted in DESIGN.md are reproducible
from a tracked file: the GPU run writes gpurun_out/parity_summary.json (the only directory a gpurun call brings back), which
is then committed as profiles/rNN_parity_summary.json."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "gpurun_out", "parity_summary.json")


def _plain(v):
    try:
        import numpy as np
        if isinstance(v, (np.generic,)):
            return v.item()
        if isinstance(v, np.ndarray):
            return v.tolist()
    except ImportError:
        pass
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def record(test_name: str, **values):
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        data = {}
        if os.path.exists(PATH):
            with open(PATH) as f:
                data = json.load(f)
        data[test_name] = _plain(values)
        with open(PATH, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass
