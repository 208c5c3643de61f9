#!/bin/bash
# round-3 profile set (one gpurun call): kernel stats of the train loop (rocprofv3 --kernel-trace --stats), the update's kernel
# sequence, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs), the bench matrix and the default bench line.
# Everything lands in gpurun_out/r03/; the summaries are then copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
PART=${1:-AB}
if [[ $PART == *A* ]]; then
python bench.py > $O/default_bench.json 2> $O/default_bench.err; echo "default bench rc=$?"
for n in 65536 8192 4096; do
  extra="--share-rows 32768"; [ $n -lt 65536 ] && extra="--segments on"   # (a fixed split: the warm-up tuning would run other splits under the profiler)
  rm -rf /tmp/ks_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -- python3 bench.py --loop-only --envs-per-gpu $n --steps 100 --warmup 20 $extra > $O/loop_$n.log 2>&1; echo "kernel stats $n rc=$?"
  f=$(find /tmp/ks_$n -name "*kernel_stats.csv" | head -1); cp $f $O/train_loop_kernel_stats_$n.csv
  python3 tools/loop_stats_to_json.py $f $n $O/train_loop_kernel_stats.json > /dev/null
done
# the same loop with the split-operand acting kernel (bench.py --acting-x3)
rm -rf /tmp/ks_x3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_x3 -- python3 bench.py --loop-only --envs-per-gpu 65536 --steps 100 --warmup 20 --acting-x3 --share-rows 16384 > $O/loop_65536_x3.log 2>&1; echo "kernel stats x3 rc=$?"
f=$(find /tmp/ks_x3 -name "*kernel_stats.csv" | head -1); cp $f $O/train_loop_kernel_stats_65536_x3.csv
python3 tools/loop_stats_to_json.py $f 65536_x3 $O/train_loop_kernel_stats.json > /dev/null
rm -rf /tmp/upd; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/upd -- python3 tools/learner_only.py 30 0 0.1 > $O/upd.log 2>&1
(cd tools && python3 update_timeline.py $(find /tmp/upd -name "*kernel_trace.csv" | head -1) ../$O/update_timeline.md > /dev/null 2>&1); echo "update timeline rc=$?"
python tools/update_bench.py > $O/update_bench.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c/a -- python3 tools/pmc_run.py envdr1000 65536 > $O/pmc_$c.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c/b -- python3 tools/pmc_run.py envdr 65536 4194304 >> $O/pmc_$c.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c/c -- python3 tools/pmc_run.py act 4096 8192 32768 65536 >> $O/pmc_$c.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c/d -- python3 tools/pmc_run.py env 65536 4194304 >> $O/pmc_$c.log 2>&1
  TVC_ACT_X3=1 timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c/e -- python3 tools/pmc_run.py act 65536 >> $O/pmc_$c.log 2>&1
  echo "pmc $c done"
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python3 tools/pmc_to_json.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/pmc_traffic.json > /dev/null; echo "pmc json rc=$?"
fi
if [[ $PART == *B* ]]; then
{
echo "| workload | env-steps/s | ms/step | SAC updates/s | flags |"; echo "|---|---|---|---|---|"
run() { python bench.py --loop-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('| %s | %.3g | %.4f | %s | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['ms_per_step'], ('%.0f' % d['sac_updates_per_s']) if 'sac_updates_per_s' in d else '-', sys.argv[1]))" "$*"; }
run --workload train --envs-per-gpu 4096 --steps 300 --warmup 30
run --workload train --envs-per-gpu 4096 --steps 300 --warmup 30 --segments on
run --workload train --envs-per-gpu 4096 --steps 300 --warmup 30 --graph
run --workload train --envs-per-gpu 8192 --steps 300 --warmup 30
run --workload train --envs-per-gpu 8192 --steps 300 --warmup 30 --segments on
run --workload train --envs-per-gpu 8192 --steps 300 --warmup 30 --graph
run --workload train --envs-per-gpu 16384 --steps 300 --warmup 30
run --workload train --envs-per-gpu 32768 --steps 300 --warmup 30
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --segments on
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --acting-x3
run --workload train --envs-per-gpu 32768 --steps 300 --warmup 30 --acting-x3
run --workload train --envs-per-gpu 16384 --steps 300 --warmup 30 --acting-x3
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --acting-x3 --updates-per-step 2
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --acting-x3 --updates-per-step 4
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --acting-dropout
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --acting-dropout --acting-x3
run --workload train --envs-per-gpu 4096 --steps 300 --warmup 30 --acting-dropout --segments on
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --updates-per-step 2
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --updates-per-step 4
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --dr-stage 0
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --shipped-acting
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --shipped-acting --acting-x3
run --workload train --envs-per-gpu 65536 --steps 300 --warmup 30 --family 1
run --workload train --envs-per-gpu 4096 --steps 300 --warmup 30 --family 1 --segments on
run --workload physics --envs-per-gpu 65536 --steps 2000 --warmup 100
run --workload physics --envs-per-gpu 4194304 --steps 100 --warmup 10
} > $O/bench_matrix.md
python tools/env_dr_bench.py > $O/env_dr_bench.jsonl 2>/dev/null
python tools/env_ring_bench.py 4096 65536 1048576 2>/dev/null | grep envs > $O/env_ring_bench.txt
{ echo "## tools/chain_bench.py: us per launch in a hipGraph chain of 64 dependent launches"; python tools/chain_bench.py 2>/dev/null | tail -1
  for mnk in "512 256 256" "512 256 512" "512 512 512"; do echo; echo "## tools/gemm_stamps.py $mnk (library built with -DTVC_GEMM_STAMPS)"; python tools/gemm_stamps.py $mnk 2>/dev/null | grep -v "^ *$"; done; } > $O/gemm_stamps.md
echo "--- act bench" > $O/act_bench.txt; python tools/act_bench.py 1024 4096 8192 12288 16384 32768 65536 2>/dev/null | grep -v "^ *$" >> $O/act_bench.txt
echo "--- act bench, split-operand kernel (tvc_sac_act flags bit 4)" >> $O/act_bench.txt; TVC_ACT_X3=1 python tools/act_bench.py 16384 32768 65536 262144 2>/dev/null | grep "^rows" >> $O/act_bench.txt
bash tools/pmc_x3.sh > /dev/null 2>&1; cp gpurun_out/pmc_x3_summary.txt $O/pmc_x3_sq.txt
cp gpurun_out/parity_summary.json $O/parity_summary.json 2>/dev/null
ls $O
fi
