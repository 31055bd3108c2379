#!/bin/bash
D=gpurun_out/r4o; mkdir -p $D
( timeout 900 python -m pytest tests/test_hip_image.py -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -15 $D/pytest.txt
