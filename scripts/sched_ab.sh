#!/bin/bash
# A/B of the step schedule on one box: prefetch of the next batch's backbone (behind the levels' forward) x lanes x hardware queues.
cd $GRAFT_REPO_ROOT
run() { # label, env..., -- bench args
  label=$1; shift
  out=$(env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config4 --no-config5 --no-alt-dtype --no-kernel-timing --no-forward-only $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))")
  echo "$label: $out"
}
for rep in 1 2; do
EXTRA="" run "base  lanes3 q4" CMPC_STREAMS=3
EXTRA="--prefetch"            run "pref  lanes3 q4" CMPC_STREAMS=3
EXTRA="--prefetch"            run "pref  lanes2 q4" CMPC_STREAMS=2
EXTRA="" run "base  lanes2 q4" CMPC_STREAMS=2
EXTRA="--prefetch"            run "pref  lanes3 q8" CMPC_STREAMS=3 GPU_MAX_HW_QUEUES=8
EXTRA="--prefetch"            run "pref  lanes2 q8" CMPC_STREAMS=2 GPU_MAX_HW_QUEUES=8
EXTRA="" run "base  lanes2 q8" CMPC_STREAMS=2 GPU_MAX_HW_QUEUES=8
EXTRA="--prefetch"            run "pref  lanes2 q6" CMPC_STREAMS=2 GPU_MAX_HW_QUEUES=6
done
