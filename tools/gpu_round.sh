#!/bin/bash
# One GPU-box session: parity tests, bench, kernel-trace profile, PMC passes of the matcher.
# usage (through gpurun): bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $O/trace.log 2>&1; echo "trace rc=$?"
if [ "$2" == "pmc" ]; then
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/tools/prof_knn.py > $O/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/tools/prof_knn.py > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/tools/prof_knn.py > $O/pmc3.log 2>&1; echo "pmc3 rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc4 -- python3 $R/tools/prof_knn.py > $O/pmc4.log 2>&1; echo "pmc4 rc=$?"
rocprofv3 -L > $O/counters.txt 2>&1
fi
ls $O
