set -e
python -m pytest tests/test_graph_gpu.py tests/test_ddp_gpu.py tests/test_training_curve_gpu.py tests/test_lineareval_gpu.py -m gpu -x -q 2>&1 | tail -3
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d["value"], d["ms_per_step"], d["config"]["hip_graph"])'
for i in 1 2; do python bench.py --graph on --lean --no-cpu-baseline 2>/dev/null | python3 -c "$P"; done
python bench.py --arch dino_vits16 --img 32 --batch 128 --lean --no-cpu-baseline 2>/dev/null | python3 -c "$P"
python bench.py --arch cait_S24_224 --batch 256 --lean --no-cpu-baseline 2>/dev/null | python3 -c "$P"
tools/prof.sh s3i --steps 5 --warmup 2 --lean
grep -i "cast_kernel\|total kernel" gpurun_out/s3i.txt
