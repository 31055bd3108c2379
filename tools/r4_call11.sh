#!/bin/bash
D=gpurun_out/r4p; mkdir -p $D
( timeout 1500 python -m pytest tests -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -6 $D/pytest.txt
