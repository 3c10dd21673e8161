"""Diagnostic: timeline of EVERY workgroup of one GEMM launch (entry, epilogue start, end in
shader cycles + the 100 MHz wall clock at the end), to see how tile time evolves over the
rounds of a launch and what the shader clock is under load.
usage: python tools/gemm_timeline.py layout:M:N:K[:epi[:cdt]] ..."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops, _lib  # noqa: E402

lib = _lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
raw.vitmi_debug_gemm_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
raw.vitmi_debug_gemm_tail(0)       # plain launch: no split tail
EPI = {"store": _lib.EPI_STORE, "gelu": _lib.EPI_BIAS_GELU, "res": _lib.EPI_RESIDUAL, "dgelu": _lib.EPI_DGELU}
for spec in sys.argv[1:]:
    if spec.startswith("stagger="):                 # start stagger of the short-list workgroups, permille of a tile time
        raw.vitmi_debug_gemm_stagger(int(spec[8:]))
        print(f"--- stagger {spec[8:]}")
        continue
    if spec.startswith("phases="):
        raw.vitmi_debug_gemm_stagger_phases(int(spec[7:]))
        continue
    if spec.startswith("rfold="):                   # residual fold on (1) / off (0) for the specs that follow
        raw.vitmi_debug_gemm_rfold(int(spec[6:]))
        print(f"--- rfold {spec[6:]}")
        continue
    parts = spec.split(":")
    layout, M, N, K = parts[0], *map(int, parts[1:4])
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    A = torch.randn((M, K) if akm else (K, M), device="cuda").to(torch.bfloat16)
    B = torch.randn((N, K) if bkm else (K, N), device="cuda").to(torch.bfloat16)
    epi = parts[4] if len(parts) > 4 else "store"
    cdt = torch.float32 if (len(parts) > 5 and parts[5] == "f32") else torch.bfloat16
    C = torch.empty((M, N), device="cuda", dtype=cdt)
    kw = dict(a_kmajor=akm, b_kmajor=bkm, epilogue=EPI[epi])
    if epi == "gelu":
        kw.update(bias=torch.randn(N, device="cuda"), C2=torch.empty_like(C))
    elif epi == "res":
        kw.update(bias=torch.randn(N, device="cuda"), R=torch.randn((M, N), device="cuda").to(cdt))
    elif epi == "dgelu":
        kw.update(aux=torch.randn((M, N), device="cuda").to(torch.bfloat16))
    nb = (M // 256) * (N // 256)
    buf = torch.zeros(64 + 8 * nb, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.gemm(A, B, C, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ops.gemm(A, B, C, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    raw.vitmi_debug_gemm_timeline(buf.data_ptr(), nb)
    for _ in range(3):                              # the last launch's stamps survive: warm clocks
        ops.gemm(A, B, C, **kw)
    torch.cuda.synchronize()
    raw.vitmi_debug_gemm_timeline(None, 64)
    t = buf.cpu().numpy()[64:].reshape(nb, 8).astype(np.float64)
    ok = t[:, 0] > 0
    print(f"{spec}: {nb} tiles, {us:.1f} us per launch unstamped; stamped blocks {int(ok.sum())}")
    # shader clock under this kernel: each workgroup's (exit - entry) in shader cycles (s_memtime) over the same
    # span on the 100 MHz wall clock (s_memrealtime), both stamped by the same wave (guide: DVFS give-back, item 6)
    life_c, life_w = t[ok, 5] - t[ok, 0], (t[ok, 6] - t[ok, 4]) * 10.0      # cycles, ns
    good = life_w > 0
    ghz = life_c[good] / life_w[good]
    print(f"  shader clock (per-workgroup entry->exit, {int(good.sum())} workgroups): median {np.median(ghz):.3f} GHz "
          f"(p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); workgroup life {np.median(life_w) / 1e3:.1f} us")
    main, ep = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
    order = np.argsort(t[:, 3], kind="stable")
    q = max(nb // 8, 1)
    for i in range(0, nb, q):
        s = order[i:i + q]
        print(f"  blocks ending {i:5d}..{min(i + q, nb) - 1:5d}: main {np.median(main[s]):8.0f} (p90 {np.percentile(main[s], 90):8.0f})"
              f"  epilogue {np.median(ep[s]):8.0f} (p90 {np.percentile(ep[s], 90):8.0f}) cycles")
