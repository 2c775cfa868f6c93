DS2_BENCH_BACKEND=gloo DS2_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 8 --warmup 2 > gpurun_out/r05_bench_sharded_gloo_dryrun.json 2> gpurun_out/r05_bench_sharded_gloo_dryrun.err
tail -c 400 gpurun_out/r05_bench_sharded_gloo_dryrun.json
bash tools/bench_configs.sh
