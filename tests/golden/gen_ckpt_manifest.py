#!/usr/bin/env python3
"""Key / shape / dtype MANIFEST of a checkpoint written by the reference's own MultiAlgorithmAgent.save_checkpoint
(agent/multi_algorithm_agent.py:1098-1141), after one SAC update so that the three Adam optimizers hold state.

Build container only (needs /root/reference).  The manifest is data: names, shapes, dtypes, container types and the
optimizer hyper-parameters -- no tensor values.  Output: tests/golden/ckpt_ref_manifest.json
"""
import collections
import json
import os
import sys
import tempfile

import numpy as np
import torch
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def describe(v):
    if isinstance(v, torch.Tensor):
        return {"tensor": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
    if isinstance(v, dict):
        return {"dict": {str(k): describe(x) for k, x in v.items()}}
    if isinstance(v, collections.deque):
        return {"deque": len(v), "maxlen": v.maxlen}
    if isinstance(v, (list, tuple)):
        return {type(v).__name__: [describe(x) for x in v][:8], "len": len(v)}
    return {type(v).__name__: v if isinstance(v, (int, float, str, bool, type(None))) else str(v)}


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from agent.multi_algorithm_agent import MultiAlgorithmAgent  # the reference, unmodified
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "config.yaml")))
    cfg["hardware"] = {"device": "cpu"}
    torch.manual_seed(0)
    agent = MultiAlgorithmAgent(10, 2, cfg)
    agent.device = torch.device("cpu")
    rng = np.random.default_rng(0)
    B = 8
    batch = {"states": torch.from_numpy(rng.standard_normal((B, 10)).astype(np.float32)),
             "actions": torch.from_numpy(rng.uniform(-1, 1, (B, 2)).astype(np.float32)),
             "rewards": torch.from_numpy(rng.standard_normal(B).astype(np.float32)),
             "next_states": torch.from_numpy(rng.standard_normal((B, 10)).astype(np.float32)),
             "dones": torch.zeros(B)}
    res = agent.update(batch, algorithm="sac")
    assert "error" not in res, res
    agent.update_performance("sac", 12.5)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ref.pth")
        agent.save_checkpoint(path)
        size = os.path.getsize(path)
        ck = torch.load(path, map_location="cpu", weights_only=False)  # a file this script just wrote itself
    sac = ck["algorithms"]["sac"]
    man = {"top_level_keys": list(ck.keys()), "algorithms": list(ck["algorithms"].keys()), "file_bytes": size,
           "performance_history": describe(ck["performance_history"]), "algorithm_weights": describe(ck["algorithm_weights"]),
           "sac_keys": list(sac.keys()), "sac_type": sac["type"], "nets": {}, "optimizers": {}}
    live = agent.algorithms["sac"]
    for net in ("policy", "q1", "q2", "target_q1", "target_q2"):
        sd = sac[f"{net}_state"]
        params = [n for n, _ in live[net].named_parameters()]
        man["nets"][net] = {"state_dict": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
                            "parameters": params}
    for opt, net in (("optimizer_policy", "policy"), ("optimizer_q1", "q1"), ("optimizer_q2", "q2")):
        od = sac[f"{opt}_state"]
        groups = []
        for g in od["param_groups"]:
            groups.append({k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in g.items()})
        state = {}
        for idx, st in od["state"].items():
            state[str(idx)] = {k: ([list(v.shape), str(v.dtype).replace("torch.", "")] if isinstance(v, torch.Tensor)
                                   else [type(v).__name__]) for k, v in st.items()}
        names = man["nets"][net]["parameters"]
        man["optimizers"][opt] = {"param_groups": groups, "state": state,
                                  "params_without_state": [names[i] for i in range(len(names)) if i not in od["state"]],
                                  "step_value_after_one_update": float(next(iter(od["state"].values()))["step"])}
    json.dump(man, open(os.path.join(OUT, "ckpt_ref_manifest.json"), "w"), indent=1)
    print("wrote manifest:", {k: len(v["state"]) for k, v in man["optimizers"].items()},
          "params without state:", man["optimizers"]["optimizer_policy"]["params_without_state"][:6], "...",
          "file", size, "bytes", "algorithms", man["algorithms"])


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    main()
