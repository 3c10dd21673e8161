"""Helpers shared by the parity tests."""
import torch


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    """max |got - want| / max |want|  (the metric BASELINE.md §4 / SURVEY.md §8d name)."""
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    denom = want.abs().max().item()
    if denom == 0:
        denom = 1.0
    return (got - want).abs().max().item() / denom


def assert_close(name, got, want, tol):
    assert tuple(got.shape) == tuple(want.shape), f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    g = got.detach().float().cpu()
    assert torch.isfinite(g).all(), f"{name}: non-finite values in result"
    e = rel_err(got, want)
    assert e <= tol, (f"{name}: rel-to-max error {e:.3e} > {tol:.1e} "
                      f"(max|want|={want.abs().max().item():.4g}, max|got|={g.abs().max().item():.4g})")
    return e


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


def grad_sample_index(name: str, numel: int, k: int = 256) -> torch.Tensor:
    """The fixed sample of a parameter's gradient entries that the full-size fixtures keep
    (tests/golden/gen_golden_full.py `gradsample/<name>`): k indices drawn with replacement from a CPU
    generator seeded by crc32(name).  Directional evidence where storing whole gradients (340 MB for
    ViT-B/16) is not an option: a gradient of the right norm and the wrong direction fails it."""
    import zlib
    g = torch.Generator("cpu").manual_seed(zlib.crc32(name.encode()))
    return torch.randint(0, numel, (min(k, numel),), generator=g)


def cosine(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    na, nb = a.norm().item(), b.norm().item()
    if na == 0.0 or nb == 0.0:
        return 1.0 if na == nb else 0.0
    return float((a @ b).item() / (na * nb))


def grad_agreement(ref, m, zero_rel: float = 1e-6):
    """Per-parameter comparison of a HIP module's gradients with the oracle's on the same weights and batch:
    (worst |norm - norm_ref| / norm_ref, its name, worst cosine, its name).  Parameters whose reference
    gradient is analytically zero (softmax shift invariance: k biases, CaiT's first talking-heads bias —
    rounding noise on both sides, below zero_rel of the largest gradient norm) are skipped."""
    pairs = [(n, pr.grad, pm.grad) for (n, pr), (n2, pm) in zip(ref.named_parameters(), m.named_parameters())
             if pr.grad is not None]
    gmax = max(g.double().norm().item() for _, g, _ in pairs)
    worst, wname, cmin, cname = 0.0, "", 1.0, ""
    for n, gr, gm in pairs:
        nr = gr.double().norm().item()
        if nr < zero_rel * gmax:
            continue
        rel = abs(gm.double().norm().item() - nr) / nr
        if rel > worst:
            worst, wname = rel, n
        c = cosine(gm, gr)
        if c < cmin:
            cmin, cname = c, n
    return worst, wname, cmin, cname
