#!/bin/bash
# Collect the round's committed profiles on the GPU box (outputs under gpurun_out/prof_r03; summaries copied to profiles/ afterwards).
#   kernel stats: every launch of 30 train steps on ONE stream (rocprofv3 --kernel-trace --stats)
#   PMC passes (separate runs, counters only): FETCH_SIZE, WRITE_SIZE, MFMA busy
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/scripts/prof_step.py 30 1 > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/prof_step.py 10 1 > $O/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/scripts/prof_step.py 10 1 > $O/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/mfma -- python3 $R/scripts/prof_step.py 10 1 > $O/mfma.log 2>&1 || exit 1
cd $R
python scripts/summarize_profile.py $O/stats $O/r03_bench_b8_f16_kernel_stats.csv 30
python scripts/pmc_traffic.py $O/fetch $O/write gemm_nt_v $O/r03_gemm_nt_traffic.json
python scripts/pmc_mfma.py $O/mfma $O/r03_mfma_busy.txt
timeout -k 10 300 python scripts/phase_timeline.py > $O/r03_phase_timeline.txt 2>/dev/null
# BASELINE configs 4 and 5: kernel stats of single-stream train steps
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -- python3 $R/scripts/prof_step_v5.py 14 1 v5 > $O/stats4.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5 -- python3 $R/scripts/prof_step_v5.py 24 1 video > $O/stats5.log 2>&1 || exit 1
cd $R
python scripts/summarize_profile.py $O/stats4 $O/r03_config4_b8_f16_kernel_stats.csv 14
python scripts/summarize_profile.py $O/stats5 $O/r03_config5_video_f16_kernel_stats.csv 24
rm -rf $O/stats4 $O/stats5
# phase trace of the 256 x 256 gemm_nt kernel (needs build/libcmpc_trace.so: scripts/build_trace_lib.sh, run before gpurun)
if [ -f build/libcmpc_trace.so ]; then (export CMPC_LIB_PATH=$PWD/build/libcmpc_trace.so; for s in "12800 1024 1024" "12800 5120 1088" "12800 1024 5120"; do timeout -k 10 100 python scripts/v5_trace.py $s; done) > $O/r03_gemm_nt_v5_phase_trace.txt 2>/dev/null; fi
rm -rf $O/stats $O/fetch $O/write $O/mfma
ls -la $O
