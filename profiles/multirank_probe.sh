#!/bin/bash
# Exercises bench.py's N>1 code paths on a ONE-GPU box: two ranks share device 0
# and meet through gloo (the driver's real runs use one GPU per rank and RCCL).
cd /root/repo
A="--steps 3 --warmup 1 --no-cpu-baseline --batch 64"
python bench.py $A --shard sites | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('N=1 sites      ', d['value'], d['lnl_check'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 2 $A --shard sites --dist-backend gloo --device 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('N=2 sites gloo ', d['value'], d['lnl_check'], d['scaling'], d['n_gpus'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29582 bench.py --gpus 2 $A --dist-backend gloo --device 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('N=2 candidates ', d['value'], d['scaling'], d['n_gpus'], d['config']['sharding'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29583 bench.py --gpus 4 $A --shard grid --site-groups 2 --dist-backend gloo --device 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('N=4 grid 2x2    ', d['value'], d['lnl_check'], d['config']['sharding'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29584 bench.py --gpus 2 $A --dist-backend gloo --device 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('N=2 candidates ', d['value'], d['lnl_check'], '(rank 0 holds the same candidates as group 0 of the grid)')"
