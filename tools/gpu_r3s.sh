#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3s; mkdir -p $O
for lanes in 4 6 8 12; do for ids in 16 24 32 48; do
  echo "== c5 u8 lanes $lanes wg_ids $ids" | tee -a $O/c5.log
  timeout -k 10 200 python bench.py --workload c5 --c5-desc u8 --lanes $lanes --ransac-wg-ids $ids --steps 5 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_batch %.2f  image_pairs_per_s %.0f  parity %s' % (d['ms_per_step'], d['image_pairs_per_s'], d['parity']))" | tee -a $O/c5.log
done; done
