#!/bin/bash
# round 4, call 53: shader clock and socket power (rocm-smi, every 0.5 s) while the gate kernels run back to back on random / zero operands:
# is the operand-dependent MFMA rate of the 16-bit kernels the power cap?
set -o pipefail
o=gpurun_out/r04/c53
mkdir -p $o
sample() {
  for i in $(seq 1 16); do
    kill -0 $2 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $o/$1.txt; echo >> $o/$1.txt
    sleep 0.5
  done
}
rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -E "sclk|Power" > $o/idle.txt
for mode in h3 6 1 0; do
  for data in random zero; do
    it=8000; [ $mode = 0 ] && it=3000
    PIVP_BENCH_DATA=$data PIVP_BENCH_BF16=$mode python3 scripts/bench_lstm_layers.py 32 $it lstm1,lstm2,lstm7 > $o/run_${mode}_$data.txt 2>&1 &
    p=$!; sleep 5; sample smi_${mode}_$data $p; wait $p || exit 1
    echo "== mode $mode data $data: $(grep sum $o/run_${mode}_$data.txt)"; tail -3 $o/smi_${mode}_$data.txt | cut -c1-260
  done
done
cat $o/idle.txt
