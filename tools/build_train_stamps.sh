#!/bin/bash
# Diagnostic build: the training kernel with in-kernel s_memtime phase stamps.  Never shipped, never timed end to end.
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DGBNF_TRAIN_STAMPS $GBNF_STAMP_FLAGS -c gbnf_train.hip -o /tmp/gbnf_train_stamps.o
objs=$(ls obj/*.o | grep -v gbnf_train.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_train_stamps.so $objs /tmp/gbnf_train_stamps.o
echo "built tools/libgbnf_train_stamps.so"
