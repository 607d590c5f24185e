cd /root/repo; mkdir -p gpurun_out/r3q
timeout 300 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_transforms.py -x -q -m gpu -k "every_instantiation or few_scales or launch_group or full_size or lean" 2>&1 | tail -2 | cut -c1-200
for eg in 2 1 4 16; do echo -n "exact_group=$eg "; SSQ_EXACT_GROUP=$eg timeout 60 python bench.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; done
SSQ_TILE_TRACE=gpurun_out/r3q/trace12.bin timeout 100 python bench.py --no-cpu --steps 3 > gpurun_out/r3q/b1.json 2>gpurun_out/r3q/b1.err
