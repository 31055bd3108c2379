#!/bin/bash
D=gpurun_out/r4b; mkdir -p $D
./tools/ubench/balance > $D/balance.txt 2>&1
cat $D/balance.txt
