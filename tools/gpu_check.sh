set -e
mkdir -p gpurun_out
for n in 2 4; do
MGX_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 5 --warmup 1 > gpurun_out/rehearse_$n.json 2> gpurun_out/rehearse_$n.err
python3 -c "import json; d=json.loads(open('gpurun_out/rehearse_$n.json').read().strip().splitlines()[-1]); print($n, round(d['ms_per_step'],3), d['config']['decomposition'], d['resnorm_first_last'], d['roofline']['kernel'][-40:], round(d['roofline']['points_per_launch']))"
done
timeout -k 10 100 python bench.py --steps 5 --warmup 1 --cpu-baseline off | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(1, d['resnorm_first_last'])"
