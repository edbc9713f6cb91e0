#!/bin/bash
# after the small-shape default of the fused filter + the capture guard: whole GPU suite, PMC passes (traffic stamp),
# the C3 line and the small-shape lines again
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
T=r3final6
O=$R/gpurun_out/$T
mkdir -p $O
bash tools/gpu_full.sh || exit 1
grep -q "pytest rc=0" gpurun_out/full/pytest.log || exit 1
bash tools/gpu_round3.sh $T pmc-only || exit 1
cd $R
timeout -k 10 300 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench_c3 rc=$?"
for n in 512 1024; do
  timeout -k 10 200 python bench.py --workload c2 --nq $n --nt $n --hyps 2000 --no-large --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_c2_small_$n.json 2> $O/bench_c2_small_$n.err; echo "bench_c2_small_$n rc=$?"
done
timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench_c2 rc=$?"
ls $O
