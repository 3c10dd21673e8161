set -e
python -m pytest tests/test_ops_gpu.py tests/test_vit_gpu.py tests/test_swin_gpu.py tests/test_cait_gpu.py tests/test_ddp_gpu.py tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -3
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d["value"], d["ms_per_step"])'
for v in 0 1 0 1; do echo -n "vitb defer $v: "; VITMI_DEFER_FOLDS=$v python bench.py --lean --no-cpu-baseline 2>/dev/null | python3 -c "$P"; done
for a in swin_tiny_patch4_window7_224 cait_S24_224; do for v in 0 1 0 1; do echo -n "$a defer $v: "; VITMI_DEFER_FOLDS=$v python bench.py --arch $a --batch 256 --lean --no-cpu-baseline 2>/dev/null | python3 -c "$P"; done; done
