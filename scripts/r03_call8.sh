#!/bin/bash
# Round 3, GPU call 8: trained weights for config 2's own model (CDNA 64x64) and for the DNA variant.
set -e -o pipefail
mkdir -p gpurun_out/r03
python3 tests/golden/train_weights.py --model CDNA --size 64 --steps 6000 --out gpurun_out/r03/trained_cdna64_q8.npz > gpurun_out/r03/train_cdna64.log 2>&1
tail -6 gpurun_out/r03/train_cdna64.log
python3 tests/golden/train_weights.py --model DNA --size 64 --steps 3000 --out gpurun_out/r03/trained_dna64_q8.npz > gpurun_out/r03/train_dna64.log 2>&1
tail -6 gpurun_out/r03/train_dna64.log
