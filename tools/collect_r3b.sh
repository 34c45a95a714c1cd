#!/bin/bash
# Round 3, second session: the GPU suite on the final revision, the folded kernel trace of one 24-cloud launch set and of one cloud
# alone (tools/trace_one_batch.py), the pass / per-queue timeline of the driver's --steps 20 run (tools/trace_passes.py,
# tools/trace_timeline.py), and the skeleton stage of each of the bench's four seeds alone (tools/diag_phases.py).
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -q -m gpu -x 2>&1 | grep -vE "amdgpu.ids|RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -8 > gpurun_out/r03_gpu_tests.txt
bash tools/run_trace_batch.sh
bash tools/run_trace_steps20.sh
O=gpurun_out/r03_select_by_seed.txt; : > $O
for S in 0 1 2 3; do
for P in "" "14=1,1=1048576,2=1048576,3=1"; do
echo "== seed $S params [$P]" >> $O
python tools/diag_phases.py 1000000 0.02 0 $S "$P" 2>&1 | grep -E "^params|phases" | cut -c1-420 >> $O
done
done
# per-layer table of every sparse-conv kernel at 16 clouds per launch set (f32 vector / f32 matrix / split-bf16 matrix / f16)
python tools/bench_conv.py 1000000 0.02 0 16 1,18 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_conv_layers_batch16.txt
