#!/bin/bash
D=gpurun_out/r4r; mkdir -p $D
( timeout 1200 python -m pytest tests/test_hip_train.py tests/test_hip_soak.py tests/test_hip_module.py -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -15 $D/pytest.txt
