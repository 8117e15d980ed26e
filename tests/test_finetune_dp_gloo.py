"""Data-parallel fine-tune exchange on CPU: two gloo ranks, each with half the batch; bucketed,
hook-launched all-reduce must reproduce the single-process full-batch gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mst_amd  # noqa: F401
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # spawned workers import tests/torch_reference.py too


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    import mst_amd.synthetic as syn
    from mst_amd.model.mdm_forstyledataset import StyleDiffusion
    m = StyleDiffusion("", 24, 1, 1, True, "rot6d", True, True, latent_dim=512, ff_size=1024, num_layers=2, num_heads=4,
                       dropout=0.0, activation="gelu", data_rep="hml_vec", cond_mode="text", cond_mask_prob=0.0,
                       arch="trans_enc", dataset="stylexia_posrot")
    sd = {k: torch.from_numpy(syn.tensor_for(3, k, tuple(v.shape)).copy()) for k, v in m.state_dict().items()
          if not k.endswith(".pe") and "clip_model" not in k}
    m.load_state_dict(sd, strict=False)
    from torch_reference import use_torch_ops
    use_torch_ops(m)               # CPU ranks: the subject here is the reducer, not the kernels
    return m.train()


def _batch(lo, hi):
    import mst_amd.synthetic as syn
    x = torch.from_numpy(syn.normal(3, "x", (4, 24, 1, 10)))[lo:hi]
    t = torch.tensor([5, 100, 600, 900])[lo:hi]
    emb = torch.from_numpy(syn.normal(3, "emb", (4, 512)))[lo:hi]
    tgt = torch.from_numpy(syn.normal(3, "tgt", (4, 24, 1, 10)))[lo:hi]
    return x, t, {"text_embed": emb}, tgt


def _loss(m, lo, hi):
    x, t, y, tgt = _batch(lo, hi)
    return ((m(x, t, y=y) - tgt) ** 2).mean()


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from mst_amd.finetune_dp import LayerBucketReducer
        m = _model()
        red = LayerBucketReducer(m)
        red.zero_grad()
        # overlap: the all-reduce of layer 1 must be LAUNCHED while backward is still running, i.e. before layer 0 has been
        # differentiated (its bucket is still all zeros at that moment)
        layer0 = [p for n, p in m.named_parameters() if p.requires_grad and ".layers.0." in n]
        seen, launch = [], red._launch

        def spy(b, where):
            seen.append((b["layer"], where, float(sum(p.grad.abs().sum() for p in layer0))))
            launch(b, where)

        red._launch = spy
        _loss(m, 2 * rank, 2 * rank + 2).backward()
        done = len(seen)                                     # launches that happened before backward() returned
        red.finish()
        g = {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad}
        q.put((rank, {k: v.numpy() for k, v in g.items()}, red.launch_order, red.bucket_bytes(), seen, done, red.launched_in))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_equals_full_batch_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.set_num_threads(2)
    m = _model()
    _loss(m, 0, 4).backward()                       # single process, whole batch (mean over 4 clips)
    ref = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    assert len(ref) == 24                            # 2 layers x 12 tensors; frozen prior has none
    for rank, grads, order, nbytes, seen, done, where in res:
        assert order == [1, 0]                       # buckets fire in backward order: last layer first
        assert done == 2 and where == ["backward", "backward"]      # both launched from inside the backward pass ...
        assert seen[0][0] == 1 and seen[0][2] == 0.0                 # ... layer 1's before layer 0 was differentiated
        assert seen[1][0] == 0 and seen[1][2] > 0.0
        assert nbytes == [2102784 * 4, 2102784 * 4]  # one 8.4 MB bucket per layer
        for n, g in ref.items():
            assert torch.allclose(torch.from_numpy(grads[n]), g, rtol=1e-4, atol=1e-7), n


def _worker_native_protocol(rank, ws, port, q):
    """The native training node's hand-off, without the GPU: gradients are computed by autograd on CPU, then delivered the
    way GradSink.flush delivers them (added into the existing p.grad bucket views in one go, followed by the
    `_native_grads_ready` notification) to a reducer in native mode (no per-parameter hooks)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from mst_amd.finetune_dp import LayerBucketReducer
        m = _model()
        params = [p for p in m.parameters() if p.requires_grad]
        local = torch.autograd.grad(_loss(m, 2 * rank, 2 * rank + 2), params)
        red = LayerBucketReducer(m)
        red.native = True                                   # as with an engine-backed model on the GPU
        m.__dict__["_native_grads_ready"] = red._native_grads_ready
        red.zero_grad()
        torch._foreach_add_([p.grad for p in params], list(local))          # GradSink.flush, existing-gradient branch
        m._native_grads_ready()
        red.finish()
        q.put((rank, {n: p.grad.clone().numpy() for n, p in m.named_parameters() if p.requires_grad}, red.launch_order))
    finally:
        dist.destroy_process_group()


def test_native_gradient_handoff_allreduces_buckets():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_native_protocol, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.set_num_threads(2)
    m = _model()
    _loss(m, 0, 4).backward()
    ref = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    for rank, grads, order in res:
        assert order == [1, 0]                       # the end-of-pass fallback still launches last layer first
        for n, g in ref.items():
            assert torch.allclose(torch.from_numpy(grads[n]), g, rtol=1e-4, atol=1e-7), n
