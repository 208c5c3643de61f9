#!/bin/bash
for w in 50 250 1200; do
python bench.py --steps 300 --warmup $w --loop-only --share-rows 32768 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 65536 warmup $w:', round(d['ms_per_step'],4))"
done
(cd .r2ref && python bench.py --steps 300 --warmup 1200 --loop-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r2 65536 warmup 1200:', round(d['ms_per_step'],4))")
python tools/step_events.py 65536 2>/dev/null | tail -14
