#!/bin/bash
# Diagnostic builds of the training kernels with one ingredient removed (results are WRONG; read the time only):
#   tools/libgbnf_train_noloads.so   no weight-fragment loads      tools/libgbnf_train_nomfma.so   no MFMAs
#   tools/libgbnf_train_noact.so     no activation function
# Use: GBNF_LIB_PATH=tools/libgbnf_train_<x>.so python tools/bench_train.py --batch 65536
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
objs=$(ls obj/*.o | grep -v gbnf_train.o)
for v in LOADS:noloads MFMA:nomfma ACT:noact; do
  flag=${v%%:*}; name=${v##*:}
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DGBNF_TR_ABLATE_$flag -c gbnf_train.hip -o /tmp/gbnf_train_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_train_$name.so $objs /tmp/gbnf_train_$name.o
  echo "built tools/libgbnf_train_$name.so"
done
