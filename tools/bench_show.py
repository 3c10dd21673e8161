"""Print the interesting parts of a bench.py JSON line.  usage: python tools/bench_show.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ['value', 'ms_per_step', 'step_mfma_frac', 'config', 'cls_only_last_block', 'parity_mode_rate', 'residual_alt', 'cpu_baseline']:
    print(k, json.dumps(d.get(k))[:700])
for k, v in (d.get('configs') or {}).items():
    print('  ', k, json.dumps(v))
print('dp_proxy', json.dumps({k: v for k, v in (d.get('dp_proxy') or {}).items() if k != 'note'})[:900])
r = d['roofline']
print({k: r.get(k) for k in ['achieved', 'frac', 'step_frac', 'gemm_ms_per_step', 'clock_mhz_under_load', 'traffic', 'timing_source']})
print('by_shape', [(x['kernel'], x['MNK'][1:], x['avg_ms'], x['tflops']) for x in r['by_shape']])
p = d.get('parity') or {}
print({k: {kk: vv for kk, vv in v.items() if kk in ('logits_rel', 'loss_diff', 'gradnorm_rel', 'gradsample_rel', 'gradsample_cos_min', 'pass', 'gemm_flop_share_on_256x256_tiles')} for k, v in p.items() if k in ('fp32', 'bf16', 'bf16_graph')})
