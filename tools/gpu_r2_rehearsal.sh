#!/bin/bash
# bench.py --gpus 3 on a ONE-GPU box: three ranks on device 0 over gloo (VS_BENCH_REHEARSAL=2), the
# whole N > 1 flow including the pipelined gather leg; checks the line and the gather's equality flag
export VS_BENCH_REHEARSAL=2
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29618 bench.py --gpus 3 --lanes 16500 --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
ok = False
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('n_gpus', d['n_gpus'], 'value', d['value'], 'gather', json.dumps(d.get('gather')))
        ok = d['n_gpus'] == 3 and d['gather'].get('equals_unoverlapped_gather') is True and d['gather'].get('overlapped') is True
sys.exit(0 if ok else 1)"
