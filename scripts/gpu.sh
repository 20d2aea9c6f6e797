#!/bin/bash
# The ONE runner for GPU-box calls (replaces the per-call scripts of rounds 3 and 4; their invocations are tabled in scripts/README.md).
#   gpurun --timeout S -- 'bash scripts/gpu.sh <tag> <job> [<job> ...]'      -> gpurun_out/<tag>/...
# Jobs run in the order given and the call stops at the first failure (a GPU step that failed or timed out is never followed by another).
#   tests[:<pytest -k expression>]   pytest -m gpu (one process)
#   bench[:<bench.py flags>]         the driver's command (flags appended), JSON line -> bench[_<n>].json
#   kstats:<name>:<bench.py flags>   rocprofv3 --kernel-trace --stats of bench.py <flags> -> <name>_kernel_stats.csv (+ <name>_queues.txt)
#   trace:<name>:<bench.py flags>    the same, keeping the per-dispatch trace's two-queue reports (overlap, queues), not the raw trace
#   pmc:<name>:<counters>:<flags>    one --pmc pass (counters comma-separated) -> <name>_counters.csv summary via scripts/pmc_summary.py is up to the caller
#   py:<script and args>             python3 <script> <args> > <script stem>.txt
#   sh:<command>                     any shell command (quoted by the caller)
# Every step is wrapped in `timeout -k 10 ${STEP_TIMEOUT:-900}`.
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
T="timeout -k 10 ${STEP_TIMEOUT:-900}"
n=0
for job in "$@"; do
    n=$((n + 1))
    kind=${job%%:*}
    rest=${job#*:}; [ "$rest" = "$job" ] && rest=""
    echo "== [$n] $job"
    case $kind in
    tests)
        if [ -n "$rest" ]; then $T python3 -m pytest tests -m gpu -x -q -k "$rest" > "$out/tests_$n.txt" 2>&1
        else $T python3 -m pytest tests -m gpu -x -q > "$out/tests_$n.txt" 2>&1; fi
        rc=$?; tail -n 6 "$out/tests_$n.txt" ;;
    bench)
        $T python3 bench.py $rest > "$out/bench_$n.json" 2> "$out/bench_$n.err"
        rc=$?; cat "$out/bench_$n.json"; tail -n 5 "$out/bench_$n.err" | grep -v amdgpu.ids ;;
    kstats|trace)
        name=${rest%%:*}; flags=${rest#*:}
        d=$out/prof_$name
        $T rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o "$name" -- python3 bench.py $flags > "$out/${name}_prof.log" 2>&1
        rc=$?
        f=$(find "$d" -name "${name}_kernel_stats.csv" | head -n 1)
        [ -n "$f" ] && cp "$f" "$out/${name}_kernel_stats.csv"
        tr=$(find "$d" -name "${name}_kernel_trace.csv" | head -n 1)
        if [ -n "$tr" ]; then
            python3 scripts/queue_breakdown.py "$tr" > "$out/${name}_queues.txt" 2>&1 || true
            [ "$kind" = trace ] && { python3 scripts/overlap_report.py "$tr" > "$out/${name}_overlap.txt" 2>&1 || true; }
            python3 scripts/launch_count.py "$tr" > "$out/${name}_launches.txt" 2>&1 || true
        fi
        rm -rf "$d"
        grep -h '"metric"' "$out/${name}_prof.log" | tail -n 1 > "$out/${name}_bench.json" || true
        head -n 25 "$out/${name}_kernel_stats.csv" ;;
    pmc)
        name=${rest%%:*}; r2=${rest#*:}; ctr=${r2%%:*}; flags=${r2#*:}
        d=$out/pmc_$name
        $T rocprofv3 --pmc ${ctr//,/ } --output-format csv -d "$d" -o "$name" -- python3 bench.py $flags > "$out/${name}_pmc.log" 2>&1
        rc=$? ;;
    py)
        stem=$(basename "${rest%% *}" .py)
        $T python3 $rest > "$out/${stem}_$n.txt" 2>&1
        rc=$?; tail -n 40 "$out/${stem}_$n.txt" | grep -v amdgpu.ids ;;
    sh)
        $T bash -c "$rest" > "$out/sh_$n.txt" 2>&1
        rc=$?; tail -n 40 "$out/sh_$n.txt" ;;
    *)
        echo "unknown job kind: $kind"; exit 2 ;;
    esac
    if [ $rc -ne 0 ]; then echo "== job $n ($job) failed with $rc: stopping"; exit $rc; fi
done
