#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 3000 python -m pytest tests/ -q -m gpu --timeout 900 > $O/step5_pytest_all.log 2>&1
echo "pytest rc $?" >> $O/step5_pytest_all.log
grep -E "^FAILED|^ERROR|passed|failed" $O/step5_pytest_all.log | tail -30
