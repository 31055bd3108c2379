#!/bin/bash
D=gpurun_out/r3final; mkdir -p $D
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > $D/pytest_gpu.txt 2>&1
echo "pytest rc $?"; tail -5 $D/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
