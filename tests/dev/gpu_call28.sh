#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPTQHIP_BENCH_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/call28_rep.txt 2>&1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --model llama3-70b --steps 10 --warmup 2 > gpurun_out/call28_tp.txt 2>&1
