#!/bin/bash
# Takes the evidence set of one iteration on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1150 -- 'bash tools/take_profiles.sh r03_a'
# Everything lands in gpurun_out/<tag>/ (scratch); the summaries are then copied to profiles/<tag>_*.  Every rocprofv3
# command has python3 directly behind `--`; counters (--pmc) run in their own passes with --kernel-trace only.
set -o pipefail
T=$1; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
say() { echo "[$(date +%H:%M:%S)] $*"; }
B="--no-cpu-baseline --no-similarity --no-hbm-family"
say bench c3;   timeout -k 10 400 python3 bench.py --conv-report $O/conv_per_shape_c3.jsonl < /dev/null > $O/bench_c3.json 2> $O/bench_c3.err || exit 1
say kernel stats c3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o kt -- python3 bench.py $B --steps 4 --warmup 3 < /dev/null > $O/kt_c3.log 2>&1 || exit 1
say pmc fetch
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 bench.py $B --steps 1 --warmup 1 < /dev/null > $O/pmc_f.log 2>&1 || exit 1
say pmc write
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 bench.py $B --steps 1 --warmup 1 < /dev/null > $O/pmc_w.log 2>&1 || exit 1
F=$(find $O/pmc_f -name '*counter_collection.csv' | sort | tail -1); W=$(find $O/pmc_w -name '*counter_collection.csv' | sort | tail -1)
KF=$(find $O/pmc_f -name '*kernel_trace.csv' | sort | tail -1)
python3 tools/pmc_summary.py "$F" "$W" $O/pmc_traffic_c3.json "$KF" > /dev/null || exit 1
KT=$(find $O/kt_c3 -name '*kernel_trace.csv' | sort | tail -1); KS=$(find $O/kt_c3 -name '*kernel_stats.csv' | sort | tail -1)
cp "$KS" $O/kernel_stats_c3.csv; python3 tools/gap_report.py "$KT" > $O/step_report_c3.txt 2>&1
say bench c2;   timeout -k 10 200 python3 bench.py --batch 8 --height 512 --width 1024 --criterion pixelcontrast_focal $B --steps 20 --warmup 5 < /dev/null > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
say bench c5;   timeout -k 10 300 python3 bench.py --model deeplabv3plus_resnet101 --batch 4 $B --steps 5 --warmup 2 < /dev/null > $O/bench_c5.json 2> $O/bench_c5.err || exit 1
say op report;  timeout -k 10 300 python3 tools/op_report.py --steps 3 --out $O/op_roofline_c3.json < /dev/null > $O/op_report.log 2>&1 || exit 1
say microbench; timeout -k 10 200 python3 tools/conv_bench.py all 10 < /dev/null > $O/conv_microbench.txt 2>&1 || exit 1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"
say pmc sq conv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_conv -o s -- python3 tools/conv_bench.py all 3 l < /dev/null > $O/sq_conv.log 2>&1 || exit 1
python3 tools/pmc_sq_summary.py "$(find $O/sq_conv -name '*counter_collection.csv' | sort | tail -1)" "$(find $O/sq_conv -name '*kernel_trace.csv' | sort | tail -1)" $O/pmc_sq_conv.csv conv > /dev/null
say pmc sq contrast
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_con -o s -- python3 tools/contrast_bench.py --reps 5 < /dev/null > $O/sq_con.log 2>&1 || exit 1
python3 tools/pmc_sq_summary.py "$(find $O/sq_con -name '*counter_collection.csv' | sort | tail -1)" "$(find $O/sq_con -name '*kernel_trace.csv' | sort | tail -1)" $O/pmc_sq_contrast.csv contrast_ > /dev/null
rm -rf $O/kt_c3 $O/pmc_f $O/pmc_w $O/sq_conv $O/sq_con       # raw traces: too large to pull back
say done; ls -la $O
