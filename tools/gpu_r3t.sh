#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_pair_batch_gpu.py tests/test_mgpu_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
for d in u8 f32; do for lanes in 6 8 12; do for th in 1 2; do
  echo "== c5 $d lanes $lanes host threads $th" | tee -a $O/c5.log
  timeout -k 10 200 python bench.py --workload c5 --c5-desc $d --lanes $lanes --c5-host-threads $th --steps 5 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_batch %.2f  image_pairs_per_s %.0f  h2d %.1f GB/s parity %s' % (d['ms_per_step'], d['image_pairs_per_s'], d['pcie']['h2d_GBps'], d['parity']))" | tee -a $O/c5.log
done; done; done
