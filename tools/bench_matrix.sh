#!/bin/bash
# Runs bench.py over the BASELINE.json configurations that fit one GPU and prints one summary line each.
# usage (on the GPU box): bash tools/bench_matrix.sh > gpurun_out/bench_matrix.txt
cd "$(dirname "$0")/.."
run() {
  python bench.py --loop-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('| %s | %.3g | %.4f | %s | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['ms_per_step'], ('%.0f' % d['sac_updates_per_s']) if 'sac_updates_per_s' in d else '-', sys.argv[1]))" "$*"
}
echo "| workload | env-steps/s | ms/step | SAC updates/s | flags |"
echo "|---|---|---|---|---|"
run --workload physics --envs-per-gpu 4096 --steps 2000 --warmup 100
run --workload physics --envs-per-gpu 8192 --steps 2000 --warmup 100
run --workload physics --envs-per-gpu 65536 --steps 2000 --warmup 100
run --workload physics --envs-per-gpu 4194304 --steps 100 --warmup 10
run --workload physics --envs-per-gpu 4096 --steps 1920 --warmup 192 --steps-per-launch 64
run --workload physics --envs-per-gpu 65536 --steps 1920 --warmup 192 --steps-per-launch 64
run --workload train --envs-per-gpu 4096 --steps 300 --warmup 30
run --workload train --envs-per-gpu 8192 --steps 300 --warmup 30
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --dr-stage 0
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --reward-window 10
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --updates-per-step 2
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --updates-per-step 4
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --share-cus off
run --workload train --envs-per-gpu 16384 --steps 300 --warmup 30
run --workload train --envs-per-gpu 262144 --steps 100 --warmup 10
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --no-overlap
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --shipped-acting
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --graph
run --workload train --envs-per-gpu 4096 --steps 300 --warmup 30 --family 1
run --workload train --envs-per-gpu 8192 --steps 300 --warmup 30 --family 1
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --family 1
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --family 1 --graph
