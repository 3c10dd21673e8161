"""The data-parallel path on a real GPU: RCCL ("nccl") in a world of ONE rank, so the bucketed
async all-reduce (identity here), its ordering against the backward's two HIP streams and the
join before the optimizer all run exactly as in the 8-GPU job.  World-size-2 semantics are
covered on CPU with gloo (test_ddp_cpu.py)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_world_of_one():
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_bucketed_allreduce_overlapped_with_backward_matches_plain_step(nccl_world_of_one, lib):
    from vit_torch_amd import CrossEntropyLoss, FusedSGD, VisionTransformer
    from vit_torch_amd.ddp import GradReducer

    def run(with_reducer):
        torch.manual_seed(3)
        m = VisionTransformer(img_size=64, patch_size=16, embed_dim=256, depth=4, num_heads=4, num_classes=10,
                              compute_dtype="bf16").cuda()
        m.head = torch.nn.Linear(256, 10, bias=False).cuda()
        eng = m.engine()
        red = None
        if with_reducer:
            red = GradReducer(eng.pack, min_bucket_elems=1 << 18, force=True)
            red.broadcast_parameters(0)
            eng.reducer = red
        opt = FusedSGD(m.parameters(), lr=1e-2, momentum=0.9)
        g = torch.Generator("cpu").manual_seed(0)
        x = torch.randn(256, 3, 64, 64, generator=g).cuda()     # M = 256 * 17 rows: fast GEMM path
        y = torch.randint(0, 10, (256,), generator=g).cuda()
        losses = []
        for _ in range(3):
            opt.zero_grad()
            loss = CrossEntropyLoss()(m(x), y)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        return losses, eng.pack.flat.clone(), red

    l0, p0, _ = run(False)
    l1, p1, red = run(True)
    assert len(red.launched) >= 3, "several buckets must have been exchanged during backward"
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1), "an all-reduce over one rank must not change the step"
