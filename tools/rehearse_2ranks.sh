set -e
mkdir -p gpurun_out/reh
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --one-gpu > gpurun_out/reh/iso.json 2> gpurun_out/reh/iso.err || { tail -20 gpurun_out/reh/iso.err; exit 1; }
python -c "
import json; d=[json.loads(l) for l in open('gpurun_out/reh/iso.json') if l.startswith('{')][-1]; print('isolates', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['parallelism'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --one-gpu --mode sharded > gpurun_out/reh/sh.json 2> gpurun_out/reh/sh.err || { tail -20 gpurun_out/reh/sh.err; exit 1; }
python -c "
import json; d=[json.loads(l) for l in open('gpurun_out/reh/sh.json') if l.startswith('{')][-1]; print('sharded', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['parallelism'], d['config']['ncontigs'])"
