# A/B: wave priority of the learner's kernels (prebuilt libraries via TVC_HIP_LIB) x HIP priority of the learner's stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in libtvc_hip_prio0.so libtvc_hip.so; do
  for sp in 0 -1; do
    for utd in 1 2 4; do
      echo "== $lib side_priority=$sp utd=$utd"
      TVC_SIDE_PRIORITY=$sp TVC_HIP_LIB=$PWD/tvc_ai_amd/csrc/$lib timeout -k 10 200 python tools/step_events.py 65536 $utd 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['median_us']; print({k: round(v) for k,v in d.items() if k in ('step_us','act0_us','act1_us','upd_us','upd_end')})"
    done
  done
done
