#!/bin/bash
# Developer tool: the home-list pass's and the finish kernel's tuning keys at batch 4096 / 16384 on the bench index (GPU box).
#   usage: bash tools/home_sweep.sh   (round 5: the defaults are the optimum, gpurun_out/home_sweep.txt)
for t in "HOME_CHUNK=128" "HOME_CHUNK=256" "HOME_CHUNK=512" "HOME_CHUNK=1024" "HOME_DEPTH=8" "HOME_DEPTH=16" "HOME_DEPTH=24" "HOME_CHUNK=512,HOME_DEPTH=16" "FINISH_SPAN=16" "FINISH_SPAN=32"; do echo "== $t"; HNSWGPU_TUNE="$t" timeout -k 10 200 python tools/ivf_batch_time.py 4096 16384 2>&1 | grep batch; done
