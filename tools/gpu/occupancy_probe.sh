#!/bin/bash
# How long does the fill take for launches of 1/4, 1/2, 1, 2, 4, ... "rounds" of the chip's 4 096 wavefront slots?
# (8 reads = 4 wavefronts: 2 work items x 2 graph directions)  -> gpurun_out/occupancy_probe.jsonl
mkdir -p gpurun_out; : > gpurun_out/occupancy_probe.jsonl
for n in 2048 4096 8192 12288 16384 24576 32768 65536 131072 200000; do
  timeout 200 python tools/fill_probe.py $n 2>/dev/null | tail -1 | tee -a gpurun_out/occupancy_probe.jsonl
done
