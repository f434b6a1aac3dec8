#!/bin/bash
# same-box A/B of library builds / switches, alternating: each argument is "tag[:lib.so][:ENV=VAL...]" (default: prev = build/libcmpc_prev.so
# against the in-tree build).  Prints ms_per_step of bench.py per variant and round.
VARS=("$@"); [ ${#VARS[@]} -eq 0 ] && VARS=("prev:build/libcmpc_prev.so" "cur")
for i in $(seq 1 ${ROUNDS:-3}); do
  for v in "${VARS[@]}"; do
    IFS=: read -r tag lib envs <<< "$v"
    r=$( ( [ -n "$lib" ] && export CMPC_LIB_PATH=$PWD/$lib; for e in ${envs//,/ }; do export $e; done
          timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-alt-dtype --no-kernel-timing --no-forward-only 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" ) )
    echo "$tag $r"
  done
done
