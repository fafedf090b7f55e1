#!/bin/bash
# Measurement set of one build, run on the GPU box from the repository root: tools/refresh_profiles.sh <tag>
# Leaves under gpurun_out/<tag>_*: the default bench line, rocprofv3 kernel statistics of decode + training, the two
# counter passes of the vocabulary kernel, and the per-kernel roofline table.
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o p -- python3 $R/bench.py --steps 5 --warmup 2 --train-steps 5 --no-cpu-baseline --no-variants > $O/${tag}_stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --train-steps 0 --no-cpu-baseline --no-variants > $O/${tag}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --train-steps 0 --no-cpu-baseline --no-variants > $O/${tag}_write.log 2>&1
timeout -k 10 300 python3 $R/tools/kernel_roofline.py > $O/${tag}_roofline.md 2> $O/${tag}_roofline.err

# MFMA utilisation: counters in a run of their own, durations from a trace-only run of the same workload
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_mfma_pmc -o p -- python3 $R/tools/mfma_workload.py > $O/${tag}_mfma_pmc.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_mfma_trace -o p -- python3 $R/tools/mfma_workload.py > $O/${tag}_mfma_trace.log 2>&1
python3 $R/tools/mfma_util.py $O/${tag}_mfma_pmc/p_counter_collection.csv $O/${tag}_mfma_trace/p_kernel_trace.csv > $O/${tag}_mfma_util.md
echo mfma done
# kernels that only the secondary workloads launch: kernel statistics of the whole bench line, variants included
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_vstats -o p -- python3 $R/bench.py --steps 3 --warmup 1 --train-steps 3 --no-cpu-baseline > $O/${tag}_vstats.log 2>&1
python3 $R/tools/variant_roofline.py $O/${tag}_vstats/p_kernel_stats.csv > $O/${tag}_variant_roofline.md
echo variants done
