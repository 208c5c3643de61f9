#!/bin/bash
# round 3, first GPU call: new schedule tests, the whole GPU suite, baseline of the per-GPU shard sizes, DR integrator variants
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_schedule_gpu.py -x -q > gpurun_out/r3_t1_schedule.log 2>&1; echo "schedule tests rc=$?"
tail -15 gpurun_out/r3_t1_schedule.log
python -m pytest tests -m gpu -q --deselect tests/test_schedule_gpu.py > gpurun_out/r3_t1_all.log 2>&1; echo "gpu suite rc=$?"
tail -8 gpurun_out/r3_t1_all.log
python tools/env_dr_bench.py > gpurun_out/r3_env_dr.log 2>&1; echo "env dr rc=$?"; cat gpurun_out/r3_env_dr.log
for n in 4096 8192; do
  python bench.py --envs-per-gpu $n --steps 300 --warmup 40 --loop-only > gpurun_out/r3_base_$n.json 2> gpurun_out/r3_base_$n.err; echo "bench $n rc=$?"
  python bench.py --envs-per-gpu $n --steps 300 --warmup 40 --loop-only --segments on > gpurun_out/r3_base_seg_$n.json 2> gpurun_out/r3_base_seg_$n.err; echo "bench seg $n rc=$?"
  python bench.py --envs-per-gpu $n --steps 300 --warmup 40 --loop-only --graph > gpurun_out/r3_base_graph_$n.json 2> gpurun_out/r3_base_graph_$n.err; echo "bench graph $n rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_base*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), "ms/step", round(d["sac_updates_per_s"],1), "upd/s", d["config"]["workload"][-60:])
    except Exception as e: print(f, "ERR", e)
PY
