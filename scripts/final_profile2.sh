#!/bin/bash
# Second profile call of a round (after scripts/profile_round.sh): BASELINE.json's configs 3-5, the train step on ONE stream (standalone kernel times:
# scripts/train_standalone_table.py), the CPU thread scan, a two-rank shared-GPU rehearsal of the data-parallel path.   bash scripts/final_profile2.sh <tag>
set -o pipefail
tag=${1:-cur}
out=gpurun_out/prof2_$tag
mkdir -p $out
export TMPDIR=/tmp
NB="--no-cpu-baseline"
for c in 3 4 5; do python3 bench.py --config $c > $out/config${c}_bench.json 2> $out/config${c}_bench.err || echo "config $c failed"; tail -c 400 $out/config${c}_bench.json; echo; done
for prec in fp32 bf16; do
    python3 bench.py --mode train --precision $prec --steps 10 --warmup 3 $NB --no-roofline > $out/train_${prec}_two_streams.json 2> $out/train_${prec}_two_streams.err
    PIVP_SIDE_STREAM=0 python3 bench.py --mode train --precision $prec --steps 10 --warmup 3 $NB --no-roofline > $out/train_${prec}_one_stream.json 2> $out/train_${prec}_one_stream.err
    d=$out/ss_$prec
    PIVP_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ss -- python3 bench.py --mode train --precision $prec --steps 10 --warmup 3 $NB --no-roofline > $out/ss_$prec.log 2>&1
    f=$(find $d -name "ss_kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp "$f" $out/train_${prec}_one_stream_kernel_stats.csv
    rm -rf $d
    echo "== $prec: two streams / one stream"; python3 -c "
import json,sys
def last(p):
    for l in reversed(open(p).read().strip().split('\n')):
        if l.startswith('{'): return json.loads(l)
a=last('$out/train_${prec}_two_streams.json')['ms_per_step']; b=last('$out/train_${prec}_one_stream.json')['ms_per_step']; print(a,b)
import subprocess
print(subprocess.run([sys.executable,'scripts/train_standalone_table.py','$out/train_${prec}_one_stream_kernel_stats.csv','13',str(a),str(b)],capture_output=True,text=True).stdout)" | tee $out/train_standalone_table_$prec.md
done
d=$out/c3
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o c3 -- python3 bench.py --config 3 --no-roofline > $out/config3_prof.log 2>&1
f=$(find $d -name "c3_kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp "$f" $out/config3_kernel_stats.csv
tr=$(find $d -name "c3_kernel_trace.csv" | head -n 1)
[ -n "$tr" ] && { python3 scripts/queue_breakdown.py "$tr" > $out/config3_queues.txt 2>&1; python3 scripts/launch_count.py "$tr" > $out/config3_launches.txt 2>&1; python3 scripts/overlap_report.py "$tr" > $out/config3_overlap.txt 2>&1; }
rm -rf $d
python3 scripts/cpu_thread_scan.py > $out/cpu_thread_scan.txt 2>&1; cat $out/cpu_thread_scan.txt
python3 bench.py --gpus 2 --share-gpu --backend gloo --steps 3 --warmup 1 $NB --no-roofline > $out/share_gpu_2ranks.json 2> $out/share_gpu_2ranks.err; echo "share-gpu rc=$?"; tail -c 600 $out/share_gpu_2ranks.json
