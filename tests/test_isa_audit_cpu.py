"""The MFMA -> inline-asm hazard audit (tools/audit_asm_hazards.py) on two cross-compiled builds of the layer tail: the shipped
header must be clean, and the build WITHOUT tail_acc_settle() (round 4's 48-token bug) must be reported -- the audit fires.
hipcc cross-compiles gfx950 without a GPU; the full-library audit is tools/audit_lib.sh (profiles/r05_isa_audit.txt)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "diffusion-based-motion-style-transfer_amd", "csrc")
AUDIT = os.path.join(ROOT, "tools", "audit_asm_hazards.py")


def _compile(tmp_path, name, defines):
    out = str(tmp_path / (name + ".s"))
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", *defines, "-I" + CSRC, "-S", "--cuda-device-only",
           "-o", out, os.path.join(ROOT, "tools", "audit_selftest.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_hazard_audit_reports_the_tail_without_its_settle_and_passes_the_shipped_one(tmp_path):
    good = _compile(tmp_path, "good", [])
    r = subprocess.run([sys.executable, AUDIT, good], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 hazards" in r.stdout.splitlines()[-1]
    broken = _compile(tmp_path, "broken", ["-DMST_AUDIT_SELFTEST_NO_SETTLE"])
    r = subprocess.run([sys.executable, AUDIT, broken], capture_output=True, text=True)
    assert r.returncode == 1, "the build without tail_acc_settle() passed the audit"
    # the instantiation round 4 measured 7 % wrong: the lo-residual adds (asm v_fma_mix_f32) read accumulators of pass 1's last MFMAs
    bad = [l for l in r.stdout.splitlines() if "k_layer_tailILi3E" in l]
    assert bad and " 0 hazards" not in bad[0], r.stdout[-2000:]
    assert "v_fma_mix_f32" in r.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_wait_state_table_is_hipccs_own():
    r = subprocess.run([sys.executable, AUDIT, "--calibrate"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("table says") == 4
