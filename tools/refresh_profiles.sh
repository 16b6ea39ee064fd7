#!/bin/bash
# Evidence of one round, written under gpurun_out/<tag>/ on the GPU box (copy what is to be judged into profiles/):
#   tools/refresh_profiles.sh r03
#   bench.json                     the default bench command's JSON line (counters measured in-run by its rocprofv3 --pmc children)
#   bench_under_rocprofv3.json     the same command under rocprofv3 --kernel-trace --stats ...
#   kernel_stats.csv               ... and that run's per-kernel summary
#   pmc_step.json                  tools/pmc_step.sh: per-kernel MFMA-busy / VALU per MFMA / HBM traffic of the step
#   pp64_phase.txt, h16_phase.txt  phase timing of the layer1 ping-pong conv and of the dominant ring conv
#   s2_phase.txt                   phase timing of the plane-gather stride-2 conv (pair / single, train / eval)
#   rsp_kstats.txt                 tools/kstats.sh over the RSP workload (config 3): launches per step, busy time, top kernels
R=$GRAFT_REPO_ROOT; T=${1:-r03}; O=$R/gpurun_out/$T
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --aux-stream 0 --wgrad-stream 0 > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err
cp $O/prof/p_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
cd $R
tools/pmc_step.sh $T/pmc > $O/pmc.log 2>&1
cp $O/pmc/pmc_step.json $O/pmc_step.json; rm -rf $O/pmc/pass*
(for op in 0 1 2 3; do tools/microbench/pp64_phase_bench 640 $op; done; tools/microbench/pp64_phase_bench 128 0) > $O/pp64_phase.txt 2>&1
(tools/microbench/h16_phase_bench 640 32 128 0; tools/microbench/h16_phase_bench 640 32 128 1; tools/microbench/h16_phase_bench 640 32 128 3; tools/microbench/h16_phase_bench 640 16 256 0; tools/microbench/h16_phase_bench 64 32 128 0) > $O/h16_phase.txt 2>&1
(tools/microbench/s2_phase_bench 640 64 64 128 1 0; tools/microbench/s2_phase_bench 640 32 128 256 1 0; tools/microbench/s2_phase_bench 640 32 128 256 0 0; tools/microbench/s2_phase_bench 448 64 64 128 1 1) > $O/s2_phase.txt 2>&1
tools/kstats.sh 10 --workload rsp > $O/rsp_kstats.txt 2>&1
ls -la $O
