#!/bin/bash
# The profile recipe of a round (run on the GPU box: gpurun -- 'bash tools/profile_round.sh round6'): the default bench line, the
# 2-rank functional lines, rocprofv3 kernel traces + statistics of the bench commands the judged numbers come from, and the HBM
# traffic counters (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, never together with a trace — MI355X_MICROARCH.md, rocprofv3
# section). Everything lands under gpurun_out/<tag>/; tools/prof_summary.py condenses each pass into <name>.summary.txt, which is
# what gets copied into profiles/ (tools/collect_profiles.py <tag>). Every rocprofv3 invocation runs under `timeout`: a counter
# pass once hung for 25 GPU-minutes. The profiled commands carry --no-oracle-parity: the parity scan over the first resident
# record would be one more launch of the SAME kernel and skew its average duration and per-launch traffic.
set -u
TAG=${1:-round6}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/$TAG; [ -z "${ONLY_PROF:-}" ] && rm -rf $O; mkdir -p $O
if [ -z "${ONLY_PROF:-}" ]; then  # (ONLY_PROF=1: the rocprofv3 passes alone, into the same directory; with ONLY_PASS=<name>: that pass alone)
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 2 --force-local --steps 10 --warmup 2 > $O/cfg4_force_local_line.json 2> $O/cfg4_force_local.err; echo "force-local rc=$?"
fi
prof() { # name "command" [fetch] [write]
  local name=$1; shift
  local cmd="$1"; shift
  [ -n "${ONLY_PASS:-}" ] && [ "$ONLY_PASS" != "$name" ] && return 0  # (ONLY_PASS=<name>: that pass alone)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name/kt -o kt -- $cmd > $O/$name.kt.log 2>&1
  for p in "$@"; do
    case $p in
      fetch) timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$name/fetch -o fetch -- $cmd > $O/$name.fetch.log 2>&1 ;;
      write) timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$name/write -o write -- $cmd > $O/$name.write.log 2>&1 ;;
    esac
  done
  python tools/prof_summary.py $O/$name > $O/$name.summary.txt 2>&1
}
prof cfg1B "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-oracle-parity" ${PMC_1B:-fetch write}
prof cfg3 "python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-oracle-parity" fetch write
prof cfg5 "python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity" fetch write
prof cfg5_sorted "python bench.py --config 5 --cfg5-sorted --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity" fetch write
prof cfg5_sorted_sets "python bench.py --config 5 --cfg5-sorted --push-order 2,0,3,1 --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity"
prof cfg5_sorted_wide "python bench.py --config 5 --cfg5-sorted --cfg5-wide-dicts --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity" fetch write
prof cfg5_sorted_widerec "env FDB_RUNS_WIDE=1 python bench.py --config 5 --cfg5-sorted --steps 5 --warmup 1 --no-cpu-baseline --no-oracle-parity" fetch write
prof select "python bench.py --rows 100000000 --steps 5 --warmup 1 --no-cpu-baseline --only-other select --no-oracle-parity" fetch write
prof cfg2_sorted "python bench.py --config 2 --cfg2-sorted --steps 10 --warmup 2 --no-cpu-baseline --no-oracle-parity" fetch write
prof cfg5_merge "python bench.py --rows 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-oracle-parity --only-other cfg5_merge"
prof cfg5_1B "python bench.py --rows 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-oracle-parity --only-other cfg5_1B" fetch write
prof parquet "python bench.py --rows 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-oracle-parity --only-other parquet"
prof cfg5_exchange "python bench.py --gpus 2 --force-local --config 5 --steps 3 --warmup 1 --no-cpu-baseline"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O
[ -n "${ONLY_PROF:-}" ] && exit 0
timeout 150 python tools/step_probe.py 2>&1 | grep -v amdgpu > $O/step_probe_125M.txt
timeout 150 python tools/step_probe.py 100000000 2>&1 | grep -v amdgpu > $O/step_probe_100M.txt
