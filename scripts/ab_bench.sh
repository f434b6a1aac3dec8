#!/bin/bash
# same-box A/B of two builds of the library: build/libcmpc_prev.so (CMPC_LIB_PATH) against the in-tree one, alternating
for i in 1 2 3; do
  for v in prev cur; do
    if [ $v = prev ]; then export CMPC_LIB_PATH=$PWD/build/libcmpc_prev.so; else unset CMPC_LIB_PATH; fi
    r=$(timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-alt-dtype --no-kernel-timing --no-forward-only 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v $r"
  done
done
