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
