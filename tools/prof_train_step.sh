#!/bin/bash
# kernel table of one steady-state training step (BASELINE.json configs[2]): bash tools/prof_train_step.sh <tag>
O=gpurun_out/${1:-train}
export TMPDIR=/tmp
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o t -- python3 tools/bench_train.py --batch 16 --steps 3 --warmup 2 > $O/train.log 2>&1
python3 tools/prof_db.py /tmp/prof_train/t_results.db 80 --step k_fused_adam | sed -n '/one step/,$p' > $O/train_step_kernels.txt
head -45 $O/train_step_kernels.txt | cut -c1-150
