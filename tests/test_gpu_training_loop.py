"""SURVEY section 8f-1 on the GPU: our `TrainInpaintingLoop` (native training node + fused AdamW / norms) driven like
the reference's own loop, against what the REFERENCE produced on the CPU for the same 12 seeded steps
(tests/golden/train_loop.npz; dropout off, every draw recorded).

Two runs: tests/torch_reference.py installed on the model (fp32 torch ops on the GPU + the fused optimizer kernel: isolates the optimizer and the
loop) and the default native backend (f16 MFMA operands).  Bars: fp32 path 2e-4 on the loss curve; native path 2e-3
(operand rounding, and AdamW's sign-like first steps amplify tiny gradient differences into lr-sized parameter
differences)."""
import os

import numpy as np
import pytest
import torch

from conftest import SEED, rel_l2
import loop_fixture as lf

pytestmark = pytest.mark.gpu


def _check(g, loop, rec, tmp_path, tol_loss, tol_norm, tol_delta):
    assert loop.step == int(g["final_step"]) == 12
    assert np.array_equal(rec["t"], g["t"])
    assert np.allclose(rec["lr"], g["lr"], rtol=1e-12)
    assert np.allclose(rec["loss"], g["loss"], rtol=tol_loss), (rec["loss"], g["loss"])
    assert np.allclose(rec["text_cosine"], g["text_cosine"], rtol=10 * tol_loss, atol=2e-5)
    assert rel_l2(rec["rot_mse"], g["rot_mse"]) < tol_loss
    assert np.allclose(rec["grad_norm"], g["grad_norm"], rtol=tol_norm), (rec["grad_norm"], g["grad_norm"])
    assert np.allclose(rec["param_norm"], g["param_norm"], rtol=1e-5)
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".pt"))
    assert "\n".join(files) == str(g["files"])
    ck = torch.load(os.path.join(tmp_path, "model000000012.pt"), map_location="cpu")
    assert "\n".join(ck.keys()) == str(g["ckpt_keys"])
    for k in [k[len("param|"):] for k in g.files if k.startswith("param|")]:
        init = torch.from_numpy(np.ascontiguousarray(lf.syn.tensor_for(SEED, k, tuple(ck[k].shape))))
        delta = (ck[k].detach() - init).reshape(-1)[:64].numpy()
        assert rel_l2(delta, g["delta|" + k]) < tol_delta, (k, rel_l2(delta, g["delta|" + k]))
    opt = torch.load(os.path.join(tmp_path, "opt000000012.pt"), map_location="cpu")
    assert len(opt["state"]) == 96 and float(opt["state"][sorted(opt["state"])[0]]["step"]) == 12.0
    # the optimizer file loads into a plain torch.optim.AdamW over the same parameters (what the reference's resume does)
    ref_opt = torch.optim.AdamW(list(loop.model.parameters()), lr=1e-4, weight_decay=0.01)
    ref_opt.load_state_dict(opt)


def test_loop_fp32_ops_and_fused_optimizer_vs_reference(golden, tmp_path):
    loop, rec = lf.run_loop(torch.device("cuda:0"), str(tmp_path), "torch")
    from mst_amd.optim import FusedAdamW
    assert isinstance(loop.opt, FusedAdamW)
    _check(golden["train_loop"], loop, rec, tmp_path, tol_loss=2e-4, tol_norm=2e-3, tol_delta=5e-2)


def test_loop_native_vs_reference(golden, tmp_path):
    loop, rec = lf.run_loop(torch.device("cuda:0"), str(tmp_path), "native")
    _check(golden["train_loop"], loop, rec, tmp_path, tol_loss=2e-3, tol_norm=1e-2, tol_delta=0.25)


def test_fused_adamw_step_equals_torch_adamw():
    """Three steps on random tensors of awkward sizes: parameters, both moments and the two norms."""
    from mst_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    shapes = [(1536, 512), (512,), (70001,), (3, 5, 7), (1,)]
    a = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    fused = FusedAdamW(a, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05)
    ref = torch.optim.AdamW(b, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05)
    for it in range(3):
        grads = [torch.randn(s, device=dev) * (10.0 ** (it - 1)) for s in shapes]
        for p, q, g in zip(a, b, grads):
            p.grad, q.grad = g.clone(), g.clone()
        g2 = sum(float((g.double() ** 2).sum()) for g in grads)
        p2 = sum(float((p.detach().double() ** 2).sum()) for p in a)
        fused.step()
        ref.step()
        got = fused.last_sq_norms.tolist()
        assert abs(got[0] - g2) < 1e-5 * g2 and abs(got[1] - p2) < 1e-5 * p2
        for p, q in zip(a, b):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)
            assert torch.allclose(fused.state[p]["exp_avg"], ref.state[q]["exp_avg"], rtol=1e-5, atol=1e-7)
            assert torch.allclose(fused.state[p]["exp_avg_sq"], ref.state[q]["exp_avg_sq"], rtol=1e-5, atol=1e-9)
    sd = fused.state_dict()
    ref.load_state_dict(sd)                                   # same layout: interchangeable files
    with pytest.raises(RuntimeError, match="GPU parameters"):
        c = [torch.randn(4, requires_grad=True)]
        c[0].grad = torch.randn(4)
        FusedAdamW(c).step()


def test_fused_adamw_rebuilt_on_the_same_model_equals_torch_adamw():
    """build, step, delete and rebuild an optimizer on the same tensors (the caching allocator hands the second
    one the same workspace address, and small tensors reuse that block in between): the device-side tables are re-uploaded
    because the OWNER of the workspace tracks them, not a process-global cache keyed by address."""
    import copy
    import gc
    from mst_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    shapes = [(512, 512), (1024,), (4099,)]
    a = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    ref = torch.optim.AdamW(b, lr=1e-3, weight_decay=0.01)
    for round_ in range(3):
        fused = FusedAdamW(a, lr=1e-3, weight_decay=0.01)
        fused.load_state_dict(copy.deepcopy(ref.state_dict()))   # carry the moments over, as a resumed run does (torch's
        #                                                          load_state_dict aliases tensors that need no cast)
        for it in range(2):
            for p, q in zip(a, b):
                g = torch.randn_like(p)
                p.grad, q.grad = g.clone(), g.clone()           # fresh gradient tensors: pointers move between steps
            fused.step()
            ref.step()
            for p, q in zip(a, b):
                assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (round_, it)
        ws_ptr = fused._ws.data_ptr()
        del fused
        gc.collect()
        junk = [torch.full((ws_n,), 0xAB, dtype=torch.uint8, device=dev) for ws_n in (512, 4096, 9000)]   # scribble over freed blocks
        del junk
    assert ws_ptr


def test_loop_with_bucket_reducer_single_rank(golden, tmp_path):
    """reducer=LayerBucketReducer(model): world size 1 must reproduce the plain loop (gradients live in the buckets,
    the reducer is notified by the native gradient sink, finish() runs before the optimizer step)."""
    import types
    from mst_amd.diffusion import logger
    from mst_amd.finetune_dp import LayerBucketReducer
    from mst_amd.train.training_loop import TrainInpaintingLoop
    g = golden["train_loop"]
    model, diffusion = lf.build_model(torch.device("cuda:0"))
    logger.configure(dir=str(tmp_path))
    args = types.SimpleNamespace(save_dir=str(tmp_path), **lf.ARGS)
    data, style_data = lf.batches()
    platform = types.SimpleNamespace(report_scalar=lambda **k: None, close=lambda: None)
    red = LayerBucketReducer(model)
    assert red.native
    loop = TrainInpaintingLoop(args, platform, model, data, diffusion=diffusion, style_data=style_data, reducer=red)
    losses = []
    orig = diffusion.few_shot_style_finetune_losses
    diffusion.few_shot_style_finetune_losses = lambda *a, **k: (lambda t: (losses.append(float(t["loss"].detach())), t)[1])(orig(*a, **k))
    np.random.seed(SEED % (2 ** 31))
    with lf.recorded_noise("loop"):
        loop.run_loop()
    assert red.launch_order == list(range(7, -1, -1))                 # fired on the last step too
    # ... and from INSIDE the backward pass: the last native node (the text-to-motion call) hands every layer to the reducer
    # behind that layer's gradient event, before the pass has ended
    assert red.launched_in == ["backward"] * 8, red.launched_in
    assert np.allclose(losses, g["loss"], rtol=2e-3), (losses, g["loss"])
