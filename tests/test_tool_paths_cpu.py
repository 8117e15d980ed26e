"""Scripts named in the sources must exist where the text says (files moved to tools/experiments/ left dangling names in
a kernel comment and in a profile script).  Checked: every `tools/....sh|py` mentioned under csrc/, tools/ (scripts) and in tools/README.md."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r"tools/[A-Za-z0-9_/.-]+\.(?:sh|py)\b")


def test_every_tool_named_in_sources_exists():
    files = glob.glob(os.path.join(ROOT, "diffusion-based-motion-style-transfer_amd", "csrc", "**", "*.h*"), recursive=True)
    files += glob.glob(os.path.join(ROOT, "tools", "**", "*.sh"), recursive=True) + glob.glob(os.path.join(ROOT, "tools", "**", "*.py"), recursive=True)
    files += [os.path.join(ROOT, "tools", "README.md"), os.path.join(ROOT, "bench.py")]
    missing = []
    for f in files:
        with open(f, errors="replace") as fh:
            for ln, line in enumerate(fh, 1):
                for m in PAT.findall(line):
                    if "*" in m or "<" in m:
                        continue
                    if not os.path.exists(os.path.join(ROOT, m)):
                        missing.append(f"{os.path.relpath(f, ROOT)}:{ln}: {m}")
    assert not missing, "\n".join(missing)
