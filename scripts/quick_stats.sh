#!/bin/bash
# one rocprofv3 kernel-stats pass of 20 single-stream train steps -> gpurun_out/<tag>_stats.csv (per-kernel summary)
set -o pipefail
TAG=${1:-quick}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qs_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/scripts/prof_step.py 20 1 > $O/stats.log 2>&1 || exit 1
cd $R
python scripts/summarize_profile.py $O/stats $R/gpurun_out/${TAG}_stats.csv 20
rm -rf $O
