#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-f32-compare 2>/dev/null | awk '{print NR": "substr($0,1,70)}'
