#!/bin/bash
# rocprofv3 kernel trace of the step at the reference's own size (512 x 512, 2000 hypotheses)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3al; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_small -- python3 $R/bench.py --workload c2 --nq 512 --nt 512 --hyps 2000 --steps 50 --warmup 10 --no-cpu-baseline --no-verify --sustain-seconds 0 --headline-only --no-large > $O/trace_small.log 2>&1
echo "trace rc=$?"
f=$(find $O/trace_small -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats_small_512.csv && cut -c1-150 $O/kernel_stats_small_512.csv | head -12
