"""CPU oracle: plain-PyTorch fp32 restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (vit_torch_amd) never
does.
"""
