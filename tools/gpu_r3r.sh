#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ransac_fused_gpu.py tests/test_pair_batch_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
for d in u8 f32; do for lanes in 3 6; do for ids in 0 64 32; do
  echo "== c5 $d lanes $lanes wg_ids $ids" | tee -a $O/c5.log
  timeout -k 10 200 python bench.py --workload c5 --c5-desc $d --lanes $lanes --ransac-wg-ids $ids --steps 5 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_batch %.2f  image_pairs_per_s %.0f  parity %s' % (d['ms_per_step'], d['image_pairs_per_s'], d['parity']))" | tee -a $O/c5.log
done; done; done
