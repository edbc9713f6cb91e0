#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_knn_hamming_gpu.py tests/test_pair_batch_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
timeout -k 10 300 python bench.py --workload c4 --no-cpu-baseline --no-large --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 ms_per_step %.4f match_ms %.4f kernels %s parity %s' % (d['ms_per_step'], d['stage_ms']['match'], d['kernels_us'], d['parity']))" | tee -a $O/c4.log
