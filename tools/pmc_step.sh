#!/bin/bash
# PMC counters of the DEFAULT bench step, per kernel (run on the GPU box):  tools/pmc_step.sh <tag>
# Three counter-only rocprofv3 passes (--pmc with --kernel-trace, nothing else) over the same command bench.py times:
#   pass 0  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
#   pass 1  FETCH_SIZE        pass 2  WRITE_SIZE      (TCC slots: one per pass, MI355X_MICROARCH.md)
# Reduced to gpurun_out/<tag>/pmc_step.json (copy to profiles/r02_pmc_step.json; bench.py reads mfma_busy / traffic from it).
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-pmc_step}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline --no-also"
i=0
for s in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 240 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $out/pass$i -o pmc -- $CMD > $out/pass$i.log 2>&1
  i=$((i+1))
done
cd $R
python tools/pmc_step_reduce.py $out
