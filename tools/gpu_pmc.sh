#!/bin/bash
# PMC passes over one profiling driver (each counter group in its own rocprofv3 run, as the
# MI355X guide prescribes; never combined with sys/hip/hsa traces).
# usage (through gpurun): bash tools/gpu_pmc.sh <tag> <driver.py> [driver args...]
set -o pipefail
TAG=$1; DRV=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {   # pass <name> <counters...>
    local name=$1; shift
    timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/$DRV $ARGS > $O/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc"; tail -1 $O/$name.log
    if [ $rc -ge 124 ]; then echo "timed out: stopping"; exit $rc; fi
}
ARGS="$*"
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
ls $O
