set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/c5/pytest_full.log 2>&1 || { tail -30 gpurun_out/c5/pytest_full.log; exit 1; }
tail -2 gpurun_out/c5/pytest_full.log
timeout -k 10 200 python tools/h2d_ceiling.py > gpurun_out/c5/h2d.log 2>&1 && cat gpurun_out/c5/h2d.log | grep -v amdgpu.ids
for l in 2 3 4 6 8; do
  timeout -k 10 300 python bench.py --workload c5 --lanes $l --steps 3 --warmup 5 --no-cpu-baseline --no-verify > gpurun_out/c5/lanes_$l.json 2> gpurun_out/c5/lanes_$l.err || exit 1
  python -c "
import json,sys
d=json.loads(open('gpurun_out/c5/lanes_$l.json').read().strip().splitlines()[-1]); print($l, d['ms_per_step'], d['pcie'])"
done
