#!/bin/bash
# updates-per-step x acting-kernel placement at 65 536 envs (loop only).  usage (GPU box): bash tools/share_matrix.sh
cd "$(dirname "$0")/.."
run() {
  python bench.py --loop-only --steps 100 --warmup 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('| %s | %.3g | %.4f | %.0f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['ms_per_step'], d['sac_updates_per_s'], sys.argv[1]))" "$*"
}
echo "| workload | env-steps/s | ms/step | SAC updates/s | flags |"
echo "|---|---|---|---|---|"
for k in 1 2 4; do
  run --updates-per-step $k --share-cus off
  run --updates-per-step $k --share-cus on
done
run --updates-per-step 1 --share-cus off --no-overlap
