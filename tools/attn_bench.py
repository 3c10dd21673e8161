"""Micro-benchmark of the fused attention kernels on the ViT-B/16 bs256 shape (or B:N:H:hd args)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops  # noqa: E402

specs = sys.argv[1:] or ["256:197:12:64"]
for spec in specs:
    B, N, H, hd = map(int, spec.split(":"))
    D = H * hd
    qkv = (torch.randn(B, N, 3 * D, device="cuda") * 0.5).to(torch.bfloat16)
    do = torch.randn(B, N, D, device="cuda").to(torch.bfloat16)
    O = torch.empty_like(do)
    lse = torch.empty(B * H * N, device="cuda")
    dqkv = torch.empty_like(qkv)
    part = torch.empty((ops.attn_bwd_dbias_rows(B, N), 3 * D), device="cuda")
    scale = hd ** -0.5

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    import ctypes
    from vit_torch_amd import _lib
    raw = ctypes.CDLL(str(_lib.LIB_PATH))
    fl = 4.0 * B * H * N * N * hd
    for cap in (0, 4, 0, 4):
        raw.vitmi_debug_attn_fwd_waves(cap)
        tf = timed(lambda: ops.attn_fwd(qkv, O, lse, B, N, H, hd, scale))
        print(f"{spec}: fwd[waves cap {cap}] {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF)")
    raw.vitmi_debug_attn_fwd_waves(0)
    for rnd in range(2):                     # interleaved rounds, one process (guide rule 24)
        for mode, name in ((0, "split"), (1, "fused")):
            raw.vitmi_debug_attn_bwd(mode)
            tb = timed(lambda: ops.attn_bwd(qkv, O, do, lse, dqkv, B, N, H, hd, scale, dbias_part=part))
            print(f"   bwd[{name}] {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.1f} TF)")
    raw.vitmi_debug_attn_bwd(1)
    for rnd in range(2):
        tb = timed(lambda: ops.attn_bwd(qkv, O, do, lse, dqkv, B, N, H, hd, scale, dbias_part=None))
        print(f"   bwd[fused, no bias sums] {tb:7.1f} us")
        tb = timed(lambda: ops.attn_bwd(qkv, O, do, lse, dqkv, B, N, H, hd, scale, dbias_part=part))
        print(f"   bwd[fused, bias sums]    {tb:7.1f} us")
    buf = torch.zeros(64 * 8, dtype=torch.int64, device="cuda")
    raw.vitmi_debug_attn_stamps.argtypes = [ctypes.c_void_p]
    for rnd in range(2):
        for st in (0,):
            tb = timed(lambda: ops.attn_bwd(qkv, O, do, lse, dqkv, B, N, H, hd, scale, dbias_part=part))
            raw.vitmi_debug_attn_stamps(buf.data_ptr())
            ops.attn_bwd(qkv, O, do, lse, dqkv, B, N, H, hd, scale, dbias_part=part)
            torch.cuda.synchronize()
            raw.vitmi_debug_attn_stamps(None)
            t8 = buf.cpu().view(64, 8)
            t = t8[:, :5]
            d = (t[:, 1:] - t[:, :-1]).float()
            print("   fused %7.1f us; median cycles over 64 mid-launch workgroups: stage %d, K frags + zero dQ %d, "
                  "loop %d, stores + bias sums %d" % ((tb,) + tuple(d.median(0).values.tolist())))
            fine = torch.stack([t8[:, 5] - t8[:, 3], t8[:, 6] - t8[:, 5], t8[:, 7] - t8[:, 6], t8[:, 4] - t8[:, 7]], 1).float()
            print("      store phase: rows to LDS + barrier %d, next loads issued %d, store loop %d, bias sums + drain %d"
                  % tuple(fine.median(0).values.tolist()))
    raw.vitmi_debug_attn_bwd(-1)
