#!/bin/bash
# round 4, call 45: HBM traffic and MFMA-busy counters of the fp16x3 rollout's kernels (separate --pmc passes)
set -o pipefail
out=gpurun_out/r04/c45
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R="--precision fp16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py $R > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py $R > $out/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o m -- python3 bench.py $R > $out/mfma.log 2>&1
python3 - <<'EOF2'
import sys
sys.path.insert(0, 'scripts')
import pmc_summary as P
o = 'gpurun_out/r04/c45'
f = P.per_kernel(o + '/fetch', 'FETCH_SIZE'); w = P.per_kernel(o + '/write', 'WRITE_SIZE')
P.write_summary(o + '/pmc_fetch_size_summary_fp16x3.csv', f, 'avg_FETCH_SIZE_KB_raw')
P.write_summary(o + '/pmc_write_size_summary_fp16x3.csv', w, 'avg_WRITE_SIZE_KB')
P.mfma_busy(o + '/mfma', o + '/pmc_mfma_busy_summary_fp16x3.csv')
for k in f:
    if 'x6g' in k or 'convlstm_bf16' in k:
        print(k[:60], 'launches', f[k][0], 'fetch KB raw %.0f (x2 = %.1f MB)' % (f[k][1], f[k][1] * 2 / 1024), 'write KB %.0f' % w.get(k, (0, 0))[1])
EOF2
find $out -name '*counter_collection.csv' -delete
head -8 $out/pmc_mfma_busy_summary_fp16x3.csv | cut -c1-150
