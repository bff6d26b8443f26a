#!/bin/bash
# One parameterised script for every call made on the GPU box through gpurun (replaces the per-call scripts of round 3).
# usage: tools/gpu_call.sh OUT VERB [args...]      (OUT = sub-directory of gpurun_out/, created if missing, kept between verbs)
#   tests  <pytest args>                 pytest with the given arguments, log in OUT/pytest_<n>.log, last lines echoed
#   bench  <cfg> [bench.py args]         python bench.py --config <cfg> ..., JSON in OUT/bench_<cfg>[_<QRL_TAG>].json, one summary line echoed
#   prof   <cfg> [bench.py args]         rocprofv3 --kernel-trace --stats of a short bench run, per-kernel table appended to OUT/kernel_trace_summary.md
#   ab     <cfg> <kernel regex> <variant>...   same-box A/B: every variant library (build/libqrl_<v>.so; "base" = the in-tree one) twice,
#                                        alternating, under the profiler; matching kernels + ms_per_step appended to OUT/ab.log
#   abx    <cfg> <lib[,ENV=VAL...]>...   same-box A/B without the profiler: library variants and / or environment settings, twice, alternating -> OUT/abx.log
#   pmc    <cfg> <COUNTER>               one rocprofv3 --pmc pass (kernel-trace only), summary appended to OUT/pmc_summary.txt
#   alone  <name> <kprof.py args>        rocprofv3 --kernel-trace --stats of tools/kprof.py (one call at a time, a sync after every call: every kernel alone on
#                                        the chip), per-kernel table appended to OUT/kernel_alone_summary.md
#   timeline <name> <rows> <kprof.py args>   the same calls queued back to back (QRL_KPROF_PIPELINED=1); start / duration / queue of the last <rows> kernels
#   smoke                                __graft_entry__.smoke()
# Several verbs in one gpurun call: gpurun -- 'tools/gpu_call.sh r04a tests tests/test_gpu_chan.py -x -q; tools/gpu_call.sh r04a bench c4 --no-extra'
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p $O
verb=$1; shift
summ() { python - "$1" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(sys.argv[1], "ms/step", d["ms_per_step"], "value", d["value"], r.get("kernel"), r.get("kernel_ms"), "frac", r.get("frac"),
          "parity", d.get("parity_check", {}).get("status"), d.get("step_spread_ms"))
    for k in ("c2", "c3", "c4", "c5"):
        if k in d:
            e = d[k]; r = e.get("roofline", {})
            print("  ", k, "ms/step", e["ms_per_step"], "value", e["value"], r.get("kernel_ms"), "frac", r.get("frac"), "parity", e.get("parity_check", {}).get("status"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
clean() { find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -size +4M -delete; }
case $verb in
tests)
  n=$(ls $O/pytest_*.log 2>/dev/null | wc -l)
  timeout ${QRL_TEST_TIMEOUT:-1500} python -m pytest "$@" > $O/pytest_$n.log 2>&1; echo "pytest rc $?" >> $O/pytest_$n.log
  tail -n ${QRL_TAIL:-8} $O/pytest_$n.log ;;
bench)
  cfg=$1; shift
  f=$O/bench_$cfg${QRL_TAG:+_$QRL_TAG}.json
  timeout ${QRL_BENCH_TIMEOUT:-600} python bench.py --config $cfg "$@" > $f 2> ${f%.json}.err; summ $f; tail -n 3 ${f%.json}.err ;;
prof)
  cfg=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o $cfg${QRL_TAG:+_$QRL_TAG} -- python bench.py --config $cfg --steps 5 --warmup 1 --no-extra "$@" > $O/prof_$cfg.log 2>&1
  f=$(find $O/prof -name "$cfg${QRL_TAG:+_$QRL_TAG}_results.db" | head -1)
  python tools/prof_summary.py $f "$cfg${QRL_TAG:+ $QRL_TAG}: rocprofv3 --kernel-trace --stats -- python bench.py --config $cfg --steps 5 --warmup 1 --no-extra $*" | tee -a $O/kernel_trace_summary.md | head -${QRL_TAIL:-14}
  clean ;;
ab)
  cfg=$1; re=$2; shift 2
  for rep in 1 2; do for v in "$@"; do
    L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
    echo "== $cfg $v (pass $rep)" >> $O/ab.log
    QRL_LIB_PATH=$L timeout 200 rocprofv3 --kernel-trace --stats -d $O/p_$v -o $cfg -- python bench.py --config $cfg --steps ${QRL_AB_STEPS:-6} --warmup 2 --no-extra ${QRL_AB_ARGS:-} > $O/run_$v.log 2>&1
    f=$(find $O/p_$v -name '*_results.db' | head -1)
    python tools/prof_summary.py $f $v 2>/dev/null | grep -E "$re" >> $O/ab.log
    grep -o '"ms_per_step": [0-9.]*' $O/run_$v.log | head -1 >> $O/ab.log
    rm -rf $O/p_$v
  done; done
  cat $O/ab.log ;;
abx)
  # same-box A/B WITHOUT the profiler: variant spec = lib[,ENV=VAL,...] (lib = base or build/libqrl_<lib>.so); ms_per_step + median step of every run
  cfg=$1; shift
  for rep in 1 2 ${QRL_AB_REPS:-}; do for spec in "$@"; do
    v=${spec%%,*}; envs=$(echo "$spec" | tr ',' '\n' | tail -n +2 | tr '\n' ' ')
    L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
    env QRL_LIB_PATH=$L $envs timeout 200 python bench.py --config $cfg --steps ${QRL_AB_STEPS:-20} --warmup 3 --no-extra ${QRL_AB_ARGS:-} > $O/runx.log 2>$O/runx.err
    python - "$spec" $rep $O/runx.log >> $O/abx.log <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    sp = d.get("step_spread_ms") or {}
    r = d.get("roofline", {})
    print("%-40s pass %s  ms/step %.3f  median %s  kernel %s %.3f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"], sp.get("median"), r.get("kernel"), r.get("kernel_ms") or 0))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
  done; done
  cat $O/abx.log ;;
pmc)
  cfg=$1; cnt=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $cnt -d $O/pmc_${cfg}_$cnt -o $cnt --output-format csv -- python bench.py --config $cfg --steps 3 --warmup 1 --no-extra "$@" > $O/pmc_${cfg}_$cnt.log 2>&1
  f=$(find $O/pmc_${cfg}_$cnt -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## $cfg $cnt"; python tools/pmc_summary.py "$f"; } | tee -a $O/pmc_summary.txt | cut -c1-200
  clean ;;
alone)
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kprof -o $name -- python tools/kprof.py "$@" > $O/kprof_$name.log 2>&1
  f=$(find $O/kprof -name "${name}_results.db" | head -1)
  python tools/prof_summary.py $f "$name: rocprofv3 --kernel-trace --stats -- python tools/kprof.py $* (one call at a time, sync after every call)" | tee -a $O/kernel_alone_summary.md | head -${QRL_TAIL:-12}
  clean ;;
timeline)
  name=$1; rows=$2; shift 2
  QRL_KPROF_PIPELINED=1 timeout 300 rocprofv3 --kernel-trace -d $O/tl -o $name -- python tools/kprof.py "$@" > $O/tl_$name.log 2>&1
  f=$(find $O/tl -name "${name}_results.db" | head -1)
  { echo "## $name: rocprofv3 --kernel-trace -- QRL_KPROF_PIPELINED=1 python tools/kprof.py $*"; python tools/prof_timeline.py $f $rows; } | tee $O/timeline_$name.log | tail -${QRL_TAIL:-40}
  clean ;;
smoke)
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -n 3 $O/smoke.log ;;
*) echo "unknown verb $verb"; exit 2 ;;
esac
