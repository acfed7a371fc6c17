#!/bin/bash
# kernel tests touched since the last full run (conv_m3 ring, linear dx_from / planes), smoke
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "m3 or linear or bn_sums_every or conv2d_fwd" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
