#!/bin/bash
# tools/gpu.sh STAGE TASK [TASK ...] — the ONE runner for everything measured on the MI355X box (replaces the per-stage
# one-off scripts of rounds 1-2).  Every task writes under gpurun_out/STAGE/; what is to be judged is then copied to
# profiles/rNN/STAGE_*.  Arguments inside a task are comma separated (commas become spaces).
#
#   gpurun --timeout 900 -- 'bash tools/gpu.sh a tests smoke bench:train_l bench:train_l_b4:--batch,4,--no-cpu-baseline prof'
#
# tasks
#   tests[:K_EXPR[:ARGS]]        pytest tests -m gpu -q [-k "K_EXPR" with + for spaces] [args] -> pytest_gpu.log, parity_table.json
#   tenv:K=V,K=V:K_EXPR          the same subset under environment switches                 -> pytest_gpu_env.log
#   smoke                        __graft_entry__.smoke()                                -> smoke.log
#   bench:NAME[:ARGS]            python bench.py ARGS                                   -> bench_NAME.json
#   benv:NAME:K=V,K=V[:ARGS]     same with environment variables (A/B switches)         -> bench_NAME.json
#   prof[:ARGS]                  rocprofv3 --kernel-trace --stats of bench.py (train l, 5 steps) -> kernel_stats.csv, rocprof_last_step.txt
#   traffic[:WORKLOAD[:MODEL]]   PMC HBM-traffic passes (tools/pmc_traffic.py)          -> traffic_WORKLOAD_MODEL.{json,txt}
#   mfma                         MFMA-busy counter pass (tools/pmc_mfma_util.py)         -> mfma_util_train_l.txt
#   layers                       per-layer kernel times (tools/profile_train.py)         -> train_l_layer_profile.txt
#   host[:MODEL]                 host-side launch profile (tools/host_profile.py)        -> host_profile_train_MODEL.txt
#   py:NAME:SCRIPT[:ARGS]        python SCRIPT ARGS                                      -> NAME.txt
#   tl                           launch timeline of the l step from the probe build (tools/step_timeline.py) -> step_timeline_train_l.{json,txt}
#   install                      this run's traffic / timeline / rocprof-step JSON -> profiles/r06/ on the box (bench.py reports them when their kernel-source key matches)
#   tunecache                    copy the tuner cache the runs above wrote (lib/tune_cache.json) -> tune_cache.json
STAGE=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/$STAGE
mkdir -p "$O"
git rev-parse HEAD > "$O/commit.txt" 2>/dev/null || cp tools/.head_commit "$O/commit.txt" 2>/dev/null
noise='RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids'
for task in "$@"; do
    IFS=: read -r kind a b c <<< "$task"
    t0=$(date +%s)
    case $kind in
        tests) if [ -n "$a" ]; then kexpr=(-k "${a//+/ }"); else kexpr=(); fi
               rm -f $O/parity_table.json        # the parity tests write the figures they assert on (tests/conftest.py record_parity)
               (STREAMYOLO_PARITY_OUT=$PWD/$O/parity_table.json timeout 1200 python -m pytest tests -m gpu -q --durations=12 "${kexpr[@]}" ${b//,/ } 2>&1 | grep -vE "$noise") > $O/pytest_gpu.log 2>&1
               grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 ;;
        tenv)  (env ${a//,/ } timeout 900 python -m pytest tests -m gpu -q -k "${b//+/ }" 2>&1 | grep -vE "$noise") > $O/pytest_gpu_env.log 2>&1
               grep -E "passed|failed|error" $O/pytest_gpu_env.log | tail -3 ;;
        smoke) (timeout 600 python __graft_entry__.py smoke 2>&1 | grep -vE "$noise" | tail -2) > $O/smoke.log 2>&1; cat $O/smoke.log ;;
        bench) (timeout 900 python bench.py ${b//,/ } 2>$O/bench_$a.err | tail -1) > $O/bench_$a.json
               python tools/bench_line.py $O/bench_$a.json ;;
        benv)  (env ${b//,/ } timeout 900 python bench.py ${c//,/ } 2>$O/bench_$a.err | tail -1) > $O/bench_$a.json
               python tools/bench_line.py $O/bench_$a.json ;;
        prof)  rm -rf /tmp/prof_$STAGE
               (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$STAGE -- python $OLDPWD/bench.py --workload train --model l --steps 5 --warmup 4 --no-cpu-baseline --extras 0 ${a//,/ } 2>&1 | tail -1) > $O/rocprof_bench_line.json 2>&1
               cp /tmp/prof_$STAGE/*/*kernel_stats.csv $O/train_l_b8_bf16_kernel_stats.csv 2>/dev/null
               python tools/trace_analyze.py $(ls /tmp/prof_$STAGE/*/*kernel_trace.csv | head -1) --json $O/rocprof_step_train_l.json > $O/rocprof_last_step.txt 2>&1
               head -40 $O/rocprof_last_step.txt ;;
        traffic) w=${a:-train}; m=${b:-l}
               (timeout 1200 python tools/pmc_traffic.py --out $O/traffic_${w}_$m.json -- --workload $w --model $m 2>&1 | tail -20) > $O/traffic_${w}_$m.txt 2>&1
               tail -20 $O/traffic_${w}_$m.txt ;;
        mfma)  rm -rf /tmp/pmc_m_$STAGE
               (cd /tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m_$STAGE -- python $OLDPWD/bench.py --workload train --model l --steps 2 --warmup 4 --no-cpu-baseline --extras 0 > /dev/null 2>&1)
               python tools/pmc_mfma_util.py /tmp/pmc_m_$STAGE > $O/mfma_util_train_l.txt 2>&1; tail -30 $O/mfma_util_train_l.txt ;;
        layers) (timeout 900 python tools/profile_train.py ${a//,/ } 2>&1 | grep -vE "$noise") > $O/train_l_layer_profile.txt 2>&1; tail -12 $O/train_l_layer_profile.txt ;;
        host)  m=${a:-l}; (timeout 600 python tools/host_profile.py $m 2>&1 | grep -v "^$" | tail -40) > $O/host_profile_train_$m.txt 2>&1; tail -12 $O/host_profile_train_$m.txt ;;
        tl)    (STREAMYOLO_HIP_LIB=$PWD/tools/probes/_build/libstreamyolo_probe.so timeout 600 python tools/step_timeline.py --bins 0.5 --json $O/step_timeline_train_l.json 2>&1 | grep -vE "$noise") > $O/step_timeline_train_l.txt 2>&1
               head -6 $O/step_timeline_train_l.txt ;;
        install) mkdir -p profiles/r06; for f in traffic_train_l.json step_timeline_train_l.json rocprof_step_train_l.json parity_table.json; do [ -f $O/$f ] && cp $O/$f profiles/r06/; done; echo "counter files of this run installed for the bench lines behind this task" ;;
        tunecache) cp streamyolo_amd/lib/tune_cache.json $O/tune_cache.json 2>/dev/null; ls -la $O/tune_cache.json ;;
        py)    (timeout 1200 python $b ${c//,/ } 2>&1 | grep -vE "$noise") > $O/$a.txt 2>&1; tail -40 $O/$a.txt ;;
        *) echo "unknown task $task" ;;
    esac
    echo "== $task: $(( $(date +%s) - t0 )) s"
done
