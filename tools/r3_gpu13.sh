#!/bin/bash
mkdir -p gpurun_out
echo "== r2 tree"; (cd .r2ref && python bench.py --steps 300 --warmup 50 --loop-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r2 65536:', round(d['ms_per_step'],4))")
echo "== r3 tree"; python bench.py --steps 300 --warmup 50 --loop-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 65536:', round(d['ms_per_step'],4), d['sac']['acting_rows_in_sharing_form'])"
echo "== r2 tree again"; (cd .r2ref && python bench.py --steps 300 --warmup 50 --loop-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r2 65536:', round(d['ms_per_step'],4))")
echo "== r3 tree again, fixed split"; python bench.py --steps 300 --warmup 50 --loop-only --share-rows 32768 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 65536:', round(d['ms_per_step'],4), d['sac']['acting_rows_in_sharing_form'])"
python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench
from types import SimpleNamespace
from tvc_ai_amd import trainer
args = SimpleNamespace(family=0, envs_per_gpu=4096, dr_stage=5, exact_reward=False, shipped_acting=False, updates_per_step=1, no_overlap=False, share_cus="auto", share_rows=-1, reward_window=0, acting_dropout=False)
dev = torch.device("cuda:0")
for order in (("eager", "segment_graphs"), ("segment_graphs", "eager", "eager")):
    for mode in order:
        res = trainer.bench_train(args, 1, 0, dev, n_envs=4096)
        t = res["trainer"]; fn = res["step_fn"]
        if mode == "segment_graphs":
            seg = t.capture_segments(); fn = lambda k: seg()
        dt, _, _ = bench.timed_steps(fn, 200, 30, 1, dev, False)
        print(order, mode, round(dt / 200 * 1e3, 4), "ms/step", flush=True)
        t.close()
PY
