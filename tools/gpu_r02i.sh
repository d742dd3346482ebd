#!/bin/bash
mkdir -p gpurun_out/r02i
bash tools/sweep.sh r02 2>&1 | tail -14
timeout 120 python tools/host_profile_captured.py 100000 512 > gpurun_out/r02i/host_captured.txt 2>&1; head -40 gpurun_out/r02i/host_captured.txt | cut -c1-150
