#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_42
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
timeout 400 python scripts/bench_data_path.py > $O/bench_data_path.json 2> /dev/null; grep -o '"cma_update_ms": [0-9.]*' $O/bench_data_path.json
(cd gpurun_tmp/r4 && timeout 400 python scripts/bench_data_path.py 2>/dev/null | grep -o '"cma_update_ms": [0-9.]*' | sed 's/^/round-4 final: /')
