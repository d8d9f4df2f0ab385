#!/bin/bash
# Round-2 profile collection on the GPU box (from the repo root): kernel stats of the bench command, PMC passes (separate
# runs, counters only) of the step's kernels, the conv legs and the wide fp32 layer.  Summaries land in gpurun_out/.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof.sh r02_bench -- python bench.py
python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_line.err
for w in step:run_step.py conv_lenet:"run_conv.py lenet" conv_cifar:"run_conv.py cifar" wide:run_wide.py; do
  tag=${w%%:*}; cmd=${w#*:}
  bash tools/pmc.sh r02_${tag}_fetch "FETCH_SIZE" -- python tools/$cmd
  bash tools/pmc.sh r02_${tag}_write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" -- python tools/$cmd
  bash tools/pmc.sh r02_${tag}_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" -- python tools/$cmd
  for g in fetch write sq; do python tools/pmc_summary.py gpurun_out/r02_${tag}_${g}_counters.csv bnn > gpurun_out/r02_pmc_${tag}_${g}.txt; done
  bash tools/prof.sh r02_${tag} -- python tools/$cmd
done
