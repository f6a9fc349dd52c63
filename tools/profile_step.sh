#!/bin/bash
# Profiles of the headline training step for profiles/<round>: kernel durations, HBM traffic (two --pmc passes) and SQ counters
# (two --pmc passes).  Run on the GPU box from the repository root:   bash tools/profile_step.sh <out dir under gpurun_out>
set -e
OUT=$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# BENCH_ARGS: extra arguments of bench.py, e.g. "--workload cfg4"
B="python3 bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline $BENCH_ARGS"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o step -- $B > $OUT/bench_profiled.json 2> $OUT/stats.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > /dev/null 2> $OUT/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > /dev/null 2> $OUT/write.err
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq1 -- $B > /dev/null 2> $OUT/sq1.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- $B > /dev/null 2> $OUT/sq2.err
python3 tools/hbm_traffic.py $OUT/fetch $OUT/write $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt
python3 tools/pmc_summary.py $OUT/sq1 > $OUT/pmc_sq1.txt
python3 tools/pmc_summary.py $OUT/sq2 > $OUT/pmc_sq2.txt
python3 tools/pipe_bound.py $OUT/sq1 $OUT/sq2 $OUT/pipe.json > $OUT/pipe.txt
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep only the small summaries
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
