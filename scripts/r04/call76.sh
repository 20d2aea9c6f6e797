#!/bin/bash
# round 4, call 76: 150 Adam steps in every precision mode from the same initialisation on the same batch: do the loss curves stay together?
set -o pipefail
o=gpurun_out/r04/c76
mkdir -p $o
timeout -k 10 600 python scripts/r04/train_compare.py 150 2>&1 | grep -v amdgpu.ids | tee $o/train_compare.txt
