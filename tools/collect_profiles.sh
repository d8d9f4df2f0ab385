#!/bin/bash
# Round-3 profile collection on the GPU box (from the repo root): kernel stats of the bench command the driver runs, kernel
# stats + PMC passes (separate runs, counters only: FETCH_SIZE alone; WRITE_SIZE + L2 hit / miss; two SQ groups, the second
# with the MFMA-busy pair) of the step's kernels, the conv legs, the wide fp32 layer and the whole-network legs, the step and
# pipeline timelines.  Summaries land in gpurun_out/ (copy the ones to keep into profiles/).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
bash tools/prof.sh r03_bench -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
for w in step:run_step.py dense_l2:run_dense_l2.py conv_lenet:"run_conv.py lenet" conv_cifar:"run_conv.py cifar" wide:run_wide.py; do
  tag=${w%%:*}; cmd=${w#*:}
  bash tools/pmc.sh r03_${tag}_fetch "FETCH_SIZE" -- python3 tools/$cmd
  bash tools/pmc.sh r03_${tag}_write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" -- python3 tools/$cmd
  bash tools/pmc.sh r03_${tag}_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" -- python3 tools/$cmd
  bash tools/pmc.sh r03_${tag}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" -- python3 tools/$cmd
  for g in fetch write sq mfma; do python3 tools/pmc_summary.py gpurun_out/r03_${tag}_${g}_counters.csv bnn > gpurun_out/r03_pmc_${tag}_${g}.txt; done
  bash tools/prof.sh r03_${tag} -- python3 tools/$cmd
done
bash tools/prof.sh r03_legs -- python3 tools/run_legs.py both
bash tools/prof.sh r03_train -- python3 tools/bench_train.py --steps 30 --no-graph
python3 tools/bench_train.py --steps 50 > gpurun_out/r03_train_step.json 2>/dev/null
export BNN_DENSE_XCD=0
bash tools/pmc.sh r03_wide_plain_order_fetch "FETCH_SIZE" -- python3 tools/run_wide.py
unset BNN_DENSE_XCD
python3 tools/pmc_summary.py gpurun_out/r03_wide_plain_order_fetch_counters.csv bnn > gpurun_out/r03_pmc_wide_fetch_plain_order.txt
python3 tools/wide_xcd_ab.py > gpurun_out/r03_wide_tile_order.txt 2>&1
BNN_DENSE_XCD=0 python3 tools/wide_xcd_ab.py >> gpurun_out/r03_wide_tile_order.txt 2>&1
python3 tools/coissue_probe.py > gpurun_out/r03_coissue_probe.txt 2>&1
BNN_DENSE_TILE=3 python3 tools/coissue_probe.py >> gpurun_out/r03_coissue_probe.txt 2>&1
python3 tools/graph_gap_probe.py > gpurun_out/r03_graph_gap_probe.txt 2>&1
python3 tools/pipe_parts.py > gpurun_out/r03_pipe_parts.txt 2>&1
bash tools/prof_trace.sh r03_step 12 -- python3 tools/run_step_graph.py 60
bash tools/prof_trace.sh r03_pipeline 40 -- python3 tools/run_pipe.py 240 4
python3 tools/overlap_probe.py > gpurun_out/r03_overlap_probe.txt 2>&1
python3 tools/dense_stamps.py 1200 > gpurun_out/r03_dense_stamps.txt 2>&1
STAMP_DIAG=8 python3 tools/dense_stamps.py 1200 >> gpurun_out/r03_dense_stamps.txt 2>&1
STAMP_DIAG=9 python3 tools/dense_stamps.py 1200 >> gpurun_out/r03_dense_stamps.txt 2>&1
STAMP_DIAG=7 python3 tools/dense_stamps.py 1200 >> gpurun_out/r03_dense_stamps.txt 2>&1
python3 tools/draw_exp2.py > gpurun_out/r03_draw_breakdown.txt 2>&1
tools/ubench_draw > gpurun_out/r03_ubench_draw.txt 2>&1 || true
python3 tools/conv_diag.py > gpurun_out/r03_conv_phases.txt 2>&1
for d in 1 2 4 7; do BNN_CONV_DIAG=$d python3 tools/conv_diag.py >> gpurun_out/r03_conv_phases.txt 2>&1; done
echo done
