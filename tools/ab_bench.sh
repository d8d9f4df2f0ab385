#!/bin/bash
# usage: tools/ab_bench.sh : bench (no legs) with the current library, with tools/ab_old/libbnn_hip.so, and the current one again
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --gpus 1 --steps 100 --warmup 20 --windows 11 --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read())
p=l['config']['pipeline']
print('$1', 'pipelined', l['ms_per_step'], 'single', p['single_stream_ms_per_step'], 'f32', l['f32']['ms_per_step'], 'train', l['train']['ms_per_step'], 'L2', l['roofline']['avg_launch_us'], 'draw', l['roofline_draw']['avg_launch_us'], l['checked']['ok'], l.get('failed'))"; }
run new
cp bayesianneuralnetworks_amd/libbnn_hip.so /tmp/new.so
cp tools/ab_old/libbnn_hip.so bayesianneuralnetworks_amd/libbnn_hip.so
run old
cp /tmp/new.so bayesianneuralnetworks_amd/libbnn_hip.so
run new
