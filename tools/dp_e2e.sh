# two data-parallel ranks sharing one MI355X (gloo carries the collectives) against the single-process run, 2 periods
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m ader_amd.main --dataset DIGINETICA --max_periods 2 --num_epochs 3 --results_root gpurun_out/dp1 > gpurun_out/dp_single.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 -m ader_amd.main --dataset DIGINETICA --max_periods 2 --num_epochs 3 --dist_backend gloo --results_root gpurun_out/dp2 > gpurun_out/dp_two.log 2>&1
grep -E "test|Average|saved|Total time" gpurun_out/dp_single.log
echo ---
grep -E "test|Average|saved|Total time" gpurun_out/dp_two.log
