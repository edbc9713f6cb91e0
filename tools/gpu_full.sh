#!/bin/bash
# the whole -m gpu suite in one process, then smoke()
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/full; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -8 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
