#!/usr/bin/env python3
"""Audit of hand-counted register loads in a gfx950 .s file (hipcc -save-temps).

The fused layer tail streams its weight fragments with inline-asm `global_load_dwordx4 vdst, voff, s[base]` and waits with
hand-counted `s_waitcnt vmcnt(N)`.  hipcc neither counts these loads nor protects their destination registers (cdna guide 5.7):
a register copy, a spill or an AGPR park between a load and the wait that retires it would silently move garbage.  This script
walks a kernel's instruction stream in file order with a FIFO model of the vector-memory queue (in-order completion: loads,
stores, LDS-DMA all count) and reports every instruction that reads or writes a register of a load still in flight.  The walk
explores BOTH outcomes of every conditional branch (a worklist over (program counter, queue state), memoised), so loop bodies
are visited with every queue state that can reach them; paths the scalar conditions make impossible are walked too, which can
only add reports, never hide one.

  python tools/audit_stream_isa.py kernel.s k_layer_tail
"""
import re
import sys


def regs(tok):
    """v[a:b] / vN -> set of VGPR numbers."""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def main():
    path, kname = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kname, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    fifo = []                     # entries: (set of dest regs for stream loads | None for other VM ops, line number)
    bad = []
    nstream = nwait = 0
    spill = sum(1 for l in lines[start:end] if "scratch_" in l)
    agpr = sum(1 for l in lines[start:end] if "v_accvgpr" in l)
    labels = {lines[i].split(":")[0]: i for i in range(start, end) if re.match(r"^\.LBB\w+:", lines[i])}
    def is_stream_line(l):
        t = l.split(";")[0].strip()
        if not t.startswith("global_load_dwordx4"):
            return False
        ops = [o.strip() for o in t[len("global_load_dwordx4"):].split(",")]
        return len(ops) >= 3 and ops[2].split()[0].startswith("s[")
    last_stream = max((i for i in range(start, end) if is_stream_line(lines[i])), default=start)
    work = [(start, ())]
    visited = set()
    leftover = set()
    while work:
        if len(visited) > 200000:
            print("state space too large (more than 200000 (block, queue) states): audit incomplete")
            return 2
        i, fifo_t = work.pop()
        fifo = list(fifo_t)
        while i < end:
            l = lines[i].split(";")[0].strip()
            i += 1
            if i > last_stream + 1 and not any(fifo):     # past the last stream load with none in flight: nothing left to check on this path
                leftover.add(len(fifo))
                break
            if not l or l.endswith(":") or l.startswith("."):
                if l.endswith(":"):                       # block entry: memoise on (pc, queue)
                    key = (i, tuple(fifo))
                    if key in visited:
                        break
                    visited.add(key)
                continue
            op = l.split()[0]
            if op == "s_endpgm":
                leftover.add(len(fifo))
                break
            if op == "s_branch":
                i = labels[l.split()[1]]
                continue
            if op.startswith("s_cbranch"):
                work.append((labels[l.split()[1]], tuple(fifo)))
                continue
            m = re.match(r"s_waitcnt\s+(.*)", l)
            if m:
                v = re.search(r"vmcnt\((\d+)\)", m.group(1))
                if v:
                    nwait += 1
                    del fifo[:max(0, len(fifo) - int(v.group(1)))]
                continue
            inflight = set()
            for d in fifo:
                inflight |= set(d)
            operands = l[len(op):]
            used = regs(operands)
            if inflight & used:
                bad.append((i, l, tuple(sorted(inflight & used)[:4])))
            if op.startswith("global_load_lds") or op.startswith("buffer_load") or op.startswith("global_store") or \
                    op.startswith("global_atomic") or op.startswith("buffer_store") or op.startswith("scratch_"):
                fifo.append(())
            elif op.startswith("global_load"):
                ops = [o.strip() for o in operands.split(",")]
                # (dwordx2 too: the backward tail's `pre` values come in by hand-counted 8-byte loads, csrc/mst_tail_bwd.h tailb_load8)
                is_stream = op in ("global_load_dwordx4", "global_load_dwordx2") and len(ops) >= 3 and ops[2].split()[0].startswith("s[")
                if is_stream:
                    nstream += 1
                fifo.append(tuple(sorted(regs(ops[0]))) if is_stream else ())
    bad = sorted(set(bad))
    fifo = [0] * (max(leftover) if leftover else 0)
    print(f"{kname}: {nstream} stream loads, {nwait} vmcnt waits, scratch instructions {spill}, v_accvgpr {agpr}, "
          f"{len(fifo)} VM ops still in flight at s_endpgm, {len(bad)} hazards")
    for b in bad[:40]:
        print("  line %d: %s   <- in-flight v%s" % (b[0], b[1], list(b[2])))
    # --allow-spills (the training kernels): scratch traffic costs time -- every reload is a vmcnt(0) -- but is only WRONG when it touches a
    # register of a load in flight, which is what `hazards` counts
    return 1 if bad or agpr or (spill and "--allow-spills" not in sys.argv) else 0


if __name__ == "__main__":
    sys.exit(main())
