#!/bin/bash
D=gpurun_out/r4g; mkdir -p $D
cp tools/libgbnf_image_stamps16.so tools/libgbnf_image_stamps.so; python tools/image_stamps2.py 256 16 > $D/stamps16.txt 2>&1; cat $D/stamps16.txt | tail -9
cp tools/libgbnf_image_stamps8.so tools/libgbnf_image_stamps.so; python tools/image_stamps2.py 256 8 > $D/stamps8.txt 2>&1; cat $D/stamps8.txt | tail -9
