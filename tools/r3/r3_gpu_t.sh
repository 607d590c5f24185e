cd /root/repo; mkdir -p gpurun_out/r3r
timeout 300 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_transforms.py -x -q -m gpu -k "every_instantiation or few_scales or launch_group or full_size or lean" 2>&1 | tail -2 | cut -c1-200
REPS=2 NWS="12" bash tools/ab_run.sh base | tee gpurun_out/r3r/ab.txt
timeout 200 python bench.py --no-cpu --steps 4 --warmup 1 --batch 64 > gpurun_out/r3r/bench_b64.json 2> gpurun_out/r3r/bench_b64.err; python -c "import json; d=json.load(open('gpurun_out/r3r/bench_b64.json')); print('B=64', round(d['value']), d['ms_per_step'], d['per_gpu'])"
