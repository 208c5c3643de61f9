#!/usr/bin/env python3
"""Golden trace of the reference CurriculumManager (scripts/curriculum_manager.py) on config/config.yaml's
curriculum section and a synthetic evaluation sequence.  Build container only.  Output: curriculum_ref.json"""
import importlib.util
import json
import os
import sys

import numpy as np
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def eval_sequence():
    rng = np.random.default_rng(5)
    seq = []
    step = 0
    for i in range(400):
        step += int(rng.integers(2000, 12000))
        seq.append((step, {"eval_success_rate": float(np.clip(0.4 + 0.002 * i + rng.normal(0, 0.1), 0, 1)),
                           "eval_reward_mean": float(60 + 0.3 * i + rng.normal(0, 20))} if i % 3 else None))
    return seq


def main():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_curriculum_manager", os.path.join(REF, "scripts", "curriculum_manager.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "config.yaml")))["curriculum"]
    mgr = m.CurriculumManager(cfg)
    trace = []
    for step, metrics in eval_sequence():
        out = mgr.update(step, metrics)
        info = out.get("_curriculum_info", {})
        trace.append([step, mgr.current_stage_idx, info.get("stage_name"), out.get("wind_force"), out.get("mass_variation"),
                      out.get("max_initial_tilt"), info.get("stage_progress")])
    json.dump({"curriculum": cfg, "trace": trace, "stages": [[s.name, s.duration_steps] for s in mgr.stages]},
              open(os.path.join(HERE, "curriculum_ref.json"), "w"))
    print("stages", [(s.name, s.duration_steps) for s in mgr.stages], "final idx", mgr.current_stage_idx,
          "transitions", mgr.stage_transition_steps)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    main()
