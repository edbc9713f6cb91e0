#!/bin/bash
# after a kernel-source change late in the round: the PMC passes (traffic stamp), then the C4 line and its trace again
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r3final2
bash tools/gpu_round3.sh r3final2 pmc-only
timeout -k 10 300 python bench.py --workload c4 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench_c4 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-verify --sustain-seconds 0 --no-large > $O/trace_c4.log 2>&1; echo "trace_c4 rc=$?"
