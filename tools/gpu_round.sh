#!/bin/bash
# One GPU-box session: parity tests, bench, kernel-trace profile, PMC passes of the matcher.
# usage (through gpurun): bash tools/gpu_round.sh <tag> [pmc]
# A step that times out (or is killed) ends the session: no further GPU step is started after it.
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
step() {   # step <name> <seconds> <cmd...>
    local name=$1 secs=$2; shift 2
    timeout -k 10 $secs "$@"
    local rc=$?
    echo "$name rc=$rc"
    if [ $rc -ge 124 ]; then echo "$name timed out / was killed: stopping"; ls $O; exit $rc; fi
    return 0
}
step pytest 600 bash -c "python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1"; tail -3 $O/pytest_gpu.log
step bench 300 bash -c "python bench.py > $O/bench.json 2> $O/bench.err"; cat $O/bench.json
step bench_c4 300 bash -c "python bench.py --workload c4 > $O/bench_c4.json 2> $O/bench_c4.err"; cat $O/bench_c4.json
cd /tmp && export TMPDIR=/tmp
step trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $O/trace.log 2>&1
if [ "$2" == "pmc" ]; then
step pmc1 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/tools/prof_knn.py > $O/pmc1.log 2>&1
step pmc2 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/tools/prof_knn.py > $O/pmc2.log 2>&1
step pmc3 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/tools/prof_knn.py > $O/pmc3.log 2>&1
step pmc4 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc4 -- python3 $R/tools/prof_knn.py > $O/pmc4.log 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
fi
ls $O
