#!/bin/bash
# Diagnostic build: the image kernels with in-kernel s_memtime phase stamps (-DGBNF_IMG_STAMPS=<W>: the fused coupling-net kernel of
# the W-wide level is stamped, tools/image_stamps2.py).  Never shipped.
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DGBNF_IMG_STAMPS=${1:-16} ${GBNF_STAMP_EXTRA:-}"
hipcc $F -c gbnf_image.hip -o /tmp/gbnf_image_stamps.o &
hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -c gbnf_image_net.hip -o /tmp/gbnf_image_net_stamps.o &
wait
objs=$(ls obj/*.o | grep -v "gbnf_image.o\|gbnf_image_net.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_image_stamps.so $objs /tmp/gbnf_image_stamps.o /tmp/gbnf_image_net_stamps.o
echo "built tools/libgbnf_image_stamps.so"
