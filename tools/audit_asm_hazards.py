#!/usr/bin/env python3
"""Audit of MFMA -> inline-asm register hazards in a gfx950 .s file (hipcc -save-temps).

hipcc's hazard recognizer pads `v_mfma_*` -> VALU/VMEM/LDS accesses of the result registers with `s_nop`s, but it does not look
INSIDE an inline-asm statement (cdna guide 5.7 item 2): an asm instruction that reads or writes a VGPR an MFMA wrote fewer than
the required wait states earlier gets the OLD register contents (round 4: `add_half` = asm `v_fma_mix_f32` next to pass 1's last
MFMAs made the 48-token layer tail 7 % wrong).  The matrix pipe is not interlocked for these accesses.

This script walks every kernel of the file (or the ones named) instruction by instruction, BOTH outcomes of every conditional
branch (a worklist over (program counter, pending-register state), memoised), and keeps for every VGPR the number of wait
states still owed since an MFMA wrote it.  Required wait states are hipcc's own (GCNHazardRecognizer, gfx950; checked by
--calibrate, which compiles a probe kernel per opcode and counts the `s_nop`s hipcc inserts):

    XDL result (vDst) -> VALU read or write, VMEM / LDS / export READ (address, store data) of the same VGPR:   passes + 3 (+1 on
    gfx950 unless 2 passes); the destination of a load is not checked (it is written when the data returns)
        v_mfma_f32_16x16x32_{f16,bf16}   4 passes ->  8 wait states
        v_mfma_f32_32x32x16_{f16,bf16}   8 passes -> 12
        any other v_mfma opcode                    -> 20 (the 16-pass figure: conservative)
    inline-asm VALU result -> MFMA operand (A / B / C) read:  2 wait states (hipcc pads only ONE state behind ;;#ASMEND)

Every instruction is one wait state, `s_nop N` is N + 1.  Only instructions between `;;#ASMSTART` and `;;#ASMEND` are checked as
consumers (the compiler pads its own); MFMAs inside inline asm would be producers hipcc does not see either -- reported as such.

    python tools/audit_asm_hazards.py kernel.s                 # every kernel in the file
    python tools/audit_asm_hazards.py kernel.s k_layer_tail    # kernels whose mangled name contains the string
    python tools/audit_asm_hazards.py --calibrate              # re-derive the wait-state table from hipcc (needs hipcc)
Exit code 1 when any hazard is found.
"""
import os
import re
import subprocess
import sys
import tempfile

MFMA_WAIT = {"v_mfma_f32_16x16x32_f16": 8, "v_mfma_f32_16x16x32_bf16": 8, "v_mfma_f32_32x32x16_f16": 12, "v_mfma_f32_32x32x16_bf16": 12}
MFMA_WAIT_UNKNOWN = 20
ASM_VALU_TO_MFMA = 2


def regs(tok):
    """v[a:b] / vN -> set of VGPR numbers (AGPRs a[..] are a separate file and not tracked: the product build has none)."""
    out = set()
    for m in re.finditer(r"(?<![\w.])v\[(\d+):(\d+)\]|(?<![\w.\[])v(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def split_ops(l):
    op = l.split()[0]
    rest = l[len(op):]
    return op, [o.strip() for o in rest.split(",")]


def audit_kernel(lines, start, end, name):
    labels = {lines[i].split(":")[0]: i for i in range(start, end) if re.match(r"^\.LBB\w+:", lines[i])}
    # asm regions
    in_asm = [False] * (end - start)
    flag = False
    for i in range(start, end):
        s = lines[i].strip()
        if s.startswith(";;#ASMSTART"):
            flag = True
        elif s.startswith(";;#ASMEND"):
            flag = False
        in_asm[i - start] = flag
    bad = set()
    n_asm = n_mfma = 0
    counted = set()
    work = [(start, ())]
    visited = set()
    while work:
        if len(visited) > 400000:
            print(f"{name}: state space too large: audit incomplete")
            return 2, 0, 0, []
        i, st = work.pop()
        pend = dict(st)            # VGPR -> wait states still owed before a non-MFMA access
        apend = {}                 # VGPR written by an asm VALU -> wait states owed before an MFMA operand read (path-local, short)
        while i < end:
            raw = lines[i]
            l = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else ""
            idx = i
            i += 1
            if not l or l.startswith("."):
                if l.endswith(":") and not l.startswith(".Lfunc"):
                    key = (idx, tuple(sorted(pend.items())))
                    if key in visited:
                        break
                    visited.add(key)
                continue
            if l.endswith(":"):
                continue
            op, ops = split_ops(l)
            if op == "s_endpgm":
                break
            ws = 1
            if op == "s_nop":
                ws = int(ops[0], 0) + 1
            inasm = in_asm[idx - start]
            used = regs(l[len(op):])
            if op.startswith("v_mfma") or op.startswith("v_smfmac"):
                if idx not in counted:
                    counted.add(idx)
                    n_mfma += 1
                # operand reads of registers an asm VALU has just written
                src = set()
                for o in ops[1:]:
                    src |= regs(o)
                hit = [r for r in src if apend.get(r, 0) > 0]
                if hit:
                    bad.add((idx + 1, l, "MFMA reads v%s %d wait state(s) too early after an inline-asm VALU write" % (sorted(hit)[:4], max(apend[r] for r in hit))))
                if inasm:
                    bad.add((idx + 1, l, "MFMA inside inline asm: hipcc pads nothing behind it"))
                dst = regs(ops[0])
                need = MFMA_WAIT.get(op.split("_e64")[0], MFMA_WAIT_UNKNOWN)
                # the MFMA itself is one wait state for everything already pending
                for r in list(pend):
                    pend[r] -= 1
                    if pend[r] <= 0:
                        del pend[r]
                for r in list(apend):
                    apend[r] -= 1
                    if apend[r] <= 0:
                        del apend[r]
                for r in dst:
                    pend[r] = need
                continue
            if inasm and not op.startswith("s_"):
                if idx not in counted:
                    counted.add(idx)
                    n_asm += 1
                chk = used
                if re.match(r"(global_load|buffer_load|scratch_load|flat_load|ds_read|ds_load)", op) and "_lds_" not in op and " lds" not in l:
                    # a load's DESTINATION is written hundreds of cycles later, long after the matrix pipe has retired (hipcc pads only the
                    # registers a memory instruction READS: address and store data); check the operands behind the destination
                    chk = set()
                    for o in ops[1:]:
                        chk |= regs(o)
                hit = [r for r in chk if pend.get(r, 0) > 0]
                if hit:
                    bad.add((idx + 1, l, "inline-asm access of v%s %d wait state(s) before the MFMA result is there" % (sorted(hit)[:4], max(pend[r] for r in hit))))
            if inasm and op.startswith("v_") and ops:
                for r in regs(ops[0]):
                    apend[r] = ASM_VALU_TO_MFMA + 1          # + 1: this instruction's own wait state is taken off below
            for d in (pend, apend):
                for r in list(d):
                    d[r] -= ws
                    if d[r] <= 0:
                        del d[r]
            if op == "s_branch":
                i = labels[ops[0].split()[0]]
                continue
            if op.startswith("s_cbranch"):
                work.append((labels[ops[0].split()[0]], tuple(sorted(pend.items()))))
                continue
    return (1 if bad else 0), n_asm, n_mfma, sorted(bad)


def calibrate():
    src = r"""
#include <hip/hip_runtime.h>
typedef _Float16 f16; typedef __bf16 bf16;
typedef f16 f16x8 __attribute__((ext_vector_type(8))); typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4))); typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CAL(NAME, ACC, N, BUILTIN, VT) __global__ void NAME(const VT* A, const VT* B, float* o) { ACC c; for (int i = 0; i < N; i++) c[i] = 0; \
    c = BUILTIN(A[threadIdx.x], B[threadIdx.x], c, 0, 0, 0); o[threadIdx.x] = c[0] * 3.0f; }
CAL(cal_16x16x32_f16, f32x4, 4, __builtin_amdgcn_mfma_f32_16x16x32_f16, f16x8)
CAL(cal_32x32x16_f16, f32x16, 16, __builtin_amdgcn_mfma_f32_32x32x16_f16, f16x8)
CAL(cal_16x16x32_bf16, f32x4, 4, __builtin_amdgcn_mfma_f32_16x16x32_bf16, bf16x8)
CAL(cal_32x32x16_bf16, f32x16, 16, __builtin_amdgcn_mfma_f32_32x32x16_bf16, bf16x8)
"""
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "cal.hip")
        open(p, "w").write(src)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", os.path.join(d, "cal.s"), p],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(os.path.join(d, "cal.s")).read().split("\n")
    ok = True
    cur = None
    for i, l in enumerate(lines):
        t = l.split(";")[0].strip()
        if t.startswith("v_mfma"):
            cur = (t.split()[0], 0, regs(t.split()[1]))
            continue
        if cur:
            op = t.split()[0] if t else ""
            if op == "s_nop":
                cur = (cur[0], cur[1] + int(t.split()[1], 0) + 1, cur[2])
            elif op.startswith("v_") and regs(t) & cur[2]:
                want = MFMA_WAIT.get(cur[0])
                print(f"{cur[0]}: hipcc leaves {cur[1]} wait states before the first VALU read of the result; table says {want}")
                ok = ok and want == cur[1]
                cur = None
            elif op:
                cur = (cur[0], cur[1] + 1, cur[2])
    return 0 if ok else 1


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--calibrate":
        return calibrate()
    path = sys.argv[1]
    want = sys.argv[2:]
    lines = open(path).read().split("\n")
    rc = 0
    tot_k = tot_asm = tot_mfma = tot_bad = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if not m or (want and not any(w in m.group(1) for w in want)):
            continue
        name = m.group(1)
        end = next((j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")), None)
        if end is None:
            continue
        res = audit_kernel(lines, i, end, name)
        if res[0] == 2:
            rc = 2
            continue
        code, n_asm, n_mfma, bad = res
        tot_k += 1
        tot_asm += n_asm
        tot_mfma += n_mfma
        tot_bad += len(bad)
        if n_asm and n_mfma or bad:
            print(f"{name}: {n_mfma} MFMAs, {n_asm} inline-asm register instructions, {len(bad)} hazards")
        for b in bad[:12]:
            print("  line %d: %s   <- %s" % b)
        if bad:
            rc = max(rc, 1)
    print(f"== {tot_k} kernels, {tot_mfma} MFMAs, {tot_asm} inline-asm register instructions checked, {tot_bad} hazards")
    return rc


if __name__ == "__main__":
    sys.exit(main())
