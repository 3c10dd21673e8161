"""world_size-2 gloo test of the data-parallel gradient exchange (runs on CPU).

The HIP engine cannot run here, so the per-rank gradients come from the oracle;
what is under test is the product's GradReducer: contiguous buckets over the flat
gradient buffer, async all-reduce, merge of adjacent sections, SUM + grad_scale =
mean, and that the result equals single-process gradients on the global batch."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vit_ref
    from vit_torch_amd.ddp import GradReducer
    from vit_torch_amd.packing import ParamPack
    cfg = dict(img_size=16, patch_size=8, embed_dim=32, depth=3, num_heads=2)
    model = vit_ref.VisionTransformer(**cfg, apply_head=True)
    model.head = vit_ref.get_classifier_head(32, 10)
    vit_ref.seeded_init_(model, 1)
    g = torch.Generator("cpu").manual_seed(0)
    X = torch.randn(8, 3, 16, 16, generator=g)
    Y = torch.randint(0, 10, (8,), generator=g)
    # single-process reference on the global batch
    F.cross_entropy(model(X), Y).backward()
    want = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    # this rank's shard
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    F.cross_entropy(model(xs), ys).backward()
    pack = ParamPack(list(model.named_parameters()), "cpu", shadow=False)
    for p in pack.params:
        pack.g(p).copy_(p.grad)
    red = GradReducer(pack, min_bucket_elems=20000)
    # sections in the order the engine's backward finishes them
    red.section_ready(list(model.norm.parameters()) + list(model.head.parameters()))
    for blk in reversed(model.blocks):
        red.section_ready(list(blk.parameters()))
    red.section_ready([model.cls_token, model.pos_embed] + list(model.patch_embed.parameters()))
    red.finish()
    # buckets go out in REVERSE layer order (head + norm first, embeddings last), each one
    # ending where the previous one began in the flat gradient buffer (named_parameters order)
    order_ok = all(red.launched[i][0] == red.launched[i + 1][1] for i in range(len(red.launched) - 1))
    spans = {"first": red.launched[0], "last": red.launched[-1], "head_hi": pack.span(list(model.head.parameters()))[1],
             "emb_lo": pack.span([model.cls_token])[0]}
    order_ok = order_ok and spans["first"][1] == spans["head_hi"] == pack.total and spans["last"][0] == spans["emb_lo"] == 0
    covered = sorted(red.launched)
    ok_cover = covered[0][0] == 0 and covered[-1][1] == pack.total and all(
        covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    worst = 0.0
    for n, p in model.named_parameters():
        got = pack.g(p) / world          # FusedSGD applies grad_scale = 1/world
        worst = max(worst, (got - want[n]).abs().max().item() / (want[n].abs().max().item() + 1e-12))
    q.put((rank, ok_cover and order_ok, len(covered), worst))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_cover, nb, worst in res:
        assert ok_cover, f"rank {rank}: buckets do not tile the flat gradient buffer in reverse layer order"
        assert nb >= 2, "expected more than one bucket"
        assert worst < 1e-5, f"rank {rank}: averaged gradients differ from the global-batch gradients ({worst})"


def test_world1_reducer_is_a_noop():
    sys.path.insert(0, ROOT)
    from vit_torch_amd.ddp import GradReducer
    from vit_torch_amd.packing import ParamPack
    lin = torch.nn.Linear(8, 8)
    pack = ParamPack(list(lin.named_parameters()), "cpu", shadow=False)
    red = GradReducer(pack)
    red.section_ready(list(lin.parameters()))
    red.finish()
    assert red.launched == []


# ------------------------------------------------------------------ Network(ddp=...) : SURVEY §8(e) in the library ---
def _fit_data():
    g = torch.Generator("cpu").manual_seed(3)
    train = [(torch.randn(8, 3, 16, 16, generator=g), torch.randint(0, 10, (8,), generator=g)) for _ in range(3)]
    val = [(torch.randn(8, 3, 16, 16, generator=g), torch.randint(0, 10, (8,), generator=g)) for _ in range(2)]
    return train, val


def _fit_model(seed):
    from oracle import vit_ref
    cfg = dict(img_size=16, patch_size=8, embed_dim=32, depth=2, num_heads=2)
    model = vit_ref.VisionTransformer(**cfg, apply_head=True)
    model.head = vit_ref.get_classifier_head(32, 10)
    vit_ref.seeded_init_(model, seed)
    return model


def _fit_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vit_torch_amd.network import Network
    train, val = _fit_data()
    # every rank starts from DIFFERENT weights: the broadcast from rank 0 must make them equal
    net = Network(_fit_model(1 + rank), opt="torch_sgd", loss_fn=torch.nn.CrossEntropyLoss(), lr=0.05, lr_type="step",
                  lr_step=1, lr_gamma=0.5, device="cpu", ddp={"size": world, "rank": rank})
    hist = net.fit(net.shard(train), net.shard(val), epochs=2)
    out = {"rank": rank, "buckets": len(net.reducer.launched),
           "train": [(h["train"]["loss_avg"], h["train"]["acc"], h["train"]["samples_global"]) for h in hist],
           "val": [(h["val"]["loss_avg"], h["val"]["acc"]) for h in hist],
           "local_losses": [h["train"]["loss"] for h in hist],
           "params": {n: p.detach().numpy().copy() for n, p in net.model.named_parameters()}}      # numpy: pickled by value
    q.put(out)
    dist.destroy_process_group()


def test_network_fit_world2_gloo_equals_one_process_on_the_global_batches():
    """Two ranks, each on its half of every global batch (ShardedLoader), rank 1 starting from other weights: after two
    `Network.fit` epochs the parameters, the epoch losses and accuracies equal a single-process run on the whole batches
    (mean cross-entropy over the global batch = mean of the two half-batch means; gradients averaged by the reducer);
    loss and accuracy reach the ranks through ONE 2-float all-reduce per epoch."""
    sys.path.insert(0, ROOT)
    from vit_torch_amd.network import Network
    train, val = _fit_data()
    ref = Network(_fit_model(1), opt="torch_sgd", loss_fn=torch.nn.CrossEntropyLoss(), lr=0.05, lr_type="step", lr_step=1,
                  lr_gamma=0.5, device="cpu")
    want = ref.fit(train, val, epochs=2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r["buckets"] >= 1
        for e in range(2):
            loss, acc, n = r["train"][e]
            assert n == 24
            assert abs(loss - want[e]["train"]["loss_avg"]) < 2e-5, (r["rank"], e, loss, want[e]["train"]["loss_avg"])
            assert abs(acc - want[e]["train"]["acc"]) < 1e-9
            assert abs(r["val"][e][0] - want[e]["val"]["loss_avg"]) < 2e-5
            assert abs(r["val"][e][1] - want[e]["val"]["acc"]) < 1e-9
        for n, p in ref.model.named_parameters():
            d = (torch.from_numpy(r["params"][n]) - p.detach()).abs().max().item() / (p.detach().abs().max().item() + 1e-12)
            assert d < 2e-5, f"rank {r['rank']}: {n} differs from the one-process run by {d}"
    # the two ranks' local losses differ (they see different halves) while the all-reduced epoch figures agree
    assert res[0]["local_losses"] != res[1]["local_losses"]
    assert res[0]["train"] == res[1]["train"] and res[0]["val"] == res[1]["val"]
