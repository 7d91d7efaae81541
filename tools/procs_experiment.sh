#!/bin/bash
# Where does the throughput of one GPU saturate: one process with 8 registrations in flight vs several processes
# sharing the GPU (PLADE_BENCH_ONE_GPU=1: every rank on cuda:0, results over gloo), vs more hardware queues.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
B="--steps 256 --host-steps 0 --no-cpu-baseline --profiled-steps 1"
run1() { python bench.py --gpus 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('  value %.1f reg/s  ms/step %.3f  busy host threads %.2f' % (d['value'], d['ms_per_step'], d['host_rank0']['busy_host_threads_avg']))"; }
runN() { n=$1; shift; PLADE_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) bench.py --gpus $n $B "$@" 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('  value %.1f reg/s (all processes)  ms/step %.3f  busy host threads rank0 %.2f' % (d['value'], d['ms_per_step'], d['host_rank0']['busy_host_threads_avg']))"; }
echo "1 process x 8 in flight"; run1 --inflight 8
echo "1 process x 4 in flight"; run1 --inflight 4
echo "1 process x 16 in flight"; run1 --inflight 16
echo "2 processes x 4 in flight"; runN 2 --inflight 4
echo "2 processes x 8 in flight"; runN 2 --inflight 8
echo "4 processes x 4 in flight"; runN 4 --inflight 4
echo "1 process x 8 in flight, GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 run1 --inflight 8
