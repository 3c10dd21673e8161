import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from vit_torch_amd import ops
from test_cait_fused_gpu import make, reference
bt = torch.bfloat16
B, N, H, hd = 2, 196, 8, 48
qkv, dO, Wl, bl, Ww, bw = make(B, N, seed=1)
scale = hd ** -0.5
want = reference(qkv, Wl, bl, Ww, bw, dO, scale)[0].view(B, N, H, hd)
q = qkv.cuda().contiguous()
W = [t.cuda().contiguous() for t in (Wl, bl, Ww, bw)]
O = torch.full((B, N, H, hd), float("nan"), device="cuda", dtype=bt)
ops.th_attn_fwd(q, *W, O, B, H, N, hd, scale)
torch.cuda.synchronize()
got = O.float().cpu()
err = (got - want).abs()
print("max err", err.max().item(), "max want", want.abs().max().item())
print("err by head:", [round(err[:, :, h].max().item(), 3) for h in range(H)])
print("err by row block of 16:", [round(err[:, i:i + 16].max().item(), 3) for i in range(0, N, 16)])
print("err by d block:", [round(err[..., d:d + 16].max().item(), 3) for d in range(0, 48, 16)])
# the packs
ws = list(ops._WS.values())[0]
ptr0 = ws.data_ptr(); off = (-ptr0) % 256
raw = ws[off:].view(torch.int16)
rf = raw[: B * H * 28 * 512].view(B, H, 14, 2, 64, 8).view(torch.bfloat16) if False else None
rfK = ws[off: off + B * H * 28 * 1024].view(bt).view(B, H, 14, 2, 64, 8).float().cpu()
K = qkv[:, :, 1].permute(0, 2, 1, 3).float()          # [B,H,N,hd]
bad = 0
for kb in range(14):
    for ks in range(2):
        for lane in range(64):
            n, kq = lane & 15, lane >> 4
            tok, d0 = 16 * kb + n, 32 * ks + 8 * kq
            exp = torch.zeros(B, H, 8)
            if tok < N and d0 < 48:
                exp = K[:, :, tok, d0:d0 + 8]
            if not torch.equal(rfK[:, :, kb, ks, lane], exp):
                bad += 1
print("RF(K) mismatching fragments-lanes:", bad)
tf = ws[off + B * H * 28 * 1024: off + B * H * 28 * 1024 + B * H * 21 * 1024].view(bt).view(B, H, 3, 7, 64, 8).float().cpu()
V = qkv[:, :, 2].permute(0, 2, 1, 3).float()
bad = 0
for db in range(3):
    for ks in range(7):
        for lane in range(64):
            n, kq = lane & 15, lane >> 4
            exp = torch.zeros(B, H, 8)
            for e in range(8):
                tok = 32 * ks + 8 * kq + e
                if tok < N:
                    exp[:, :, e] = V[:, :, tok, 16 * db + n]
            if not torch.equal(tf[:, :, db, ks, lane], exp):
                bad += 1
print("TF(V) mismatching fragments-lanes:", bad)
