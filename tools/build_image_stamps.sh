#!/bin/bash
# Diagnostic build: the split-f16 image "mid" kernel with in-kernel s_memtime phase stamps.  Never shipped.
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DGBNF_IMG_STAMPS=${1:-16} -c gbnf_image.hip -o /tmp/gbnf_image_stamps.o
objs=$(ls obj/*.o | grep -v gbnf_image.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_image_stamps.so $objs /tmp/gbnf_image_stamps.o
echo "built tools/libgbnf_image_stamps.so"
