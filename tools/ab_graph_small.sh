# eager launches vs one hipGraph of K steps at small env counts (is the host the limit there?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$* -> "; python bench.py --loop-only --steps 300 --warmup 30 $@ 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d.get('sac_updates_per_s',0)))"; }
run --envs-per-gpu 4096
run --envs-per-gpu 4096 --graph
run --envs-per-gpu 8192
run --envs-per-gpu 8192 --graph
run --envs-per-gpu 16384
run --envs-per-gpu 16384 --graph
