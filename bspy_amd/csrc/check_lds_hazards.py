"""Build-time guard for the kernels that read LDS through inline asm (eval_stream, jac_stream,
eval_rowrot, jac_rowrot).

hipcc does not know that the destination of an `asm volatile("ds_read_b64 ...")` is still in
flight until the matching `s_waitcnt lgkmcnt(N)` (also inline asm): it is free to copy, move or
overwrite that register in between.  This script replays the device assembly of those kernels
with the hardware's rule - LDS operations return in order, `lgkmcnt(N)` waits until at most N are
outstanding - and fails the build when

  * any instruction reads or writes a VGPR that is the destination of an outstanding LDS read, or
  * reads pile up around a loop (the analysis is a forward data-flow over the control-flow graph;
    the states of a block are those of its incoming paths, a path that has waited less standing
    for the paths whose outstanding reads are a suffix of its own).

usage: check_lds_hazards.py <device .s> [--allow-none] [kernel-name-substring ...]
"""
import re
import sys

KERNELS = ("eval_stream", "jac_stream", "eval_rowrot", "jac_rowrot", "curv_rowrot", "eval_uni", "jac_uni", "curv_uni",
           "eval_cellsort", "eval_slab2", "eval_rec32")      # regular expressions on the mangled name
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
WAIT = re.compile(r"lgkmcnt\((\d+)\)")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lds_dest(mnem, ops):
    """destination VGPRs of an LDS instruction (empty for stores / atomics without return)"""
    if mnem.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) \
            or "_rtn" in mnem:
        return regs(ops.split(",")[0])
    return set()


def used_regs(mnem, ops):
    """VGPRs an instruction touches.  Packed fp32 math names register pairs but reads only the
    halves op_sel / op_sel_hi select (defaults 0 / 1) - count those only."""
    if not mnem.startswith("v_pk_") or not mnem.endswith("f32"):
        return regs(ops)
    fields = [f.strip() for f in re.split(r",(?![^\[]*\])", ops.split(" op_sel")[0].split(" neg_")[0])]
    sel = {"op_sel": None, "op_sel_hi": None}
    for key in sel:
        m = re.search(key + r":\[([01,]+)\]", ops)
        if m:
            sel[key] = [int(x) for x in m.group(1).split(",")]
    out = regs(fields[0])
    for i, f in enumerate(fields[1:]):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", f)
        if not m:
            out |= regs(f)
            continue
        lo = int(m.group(1))
        a = sel["op_sel"][i] if sel["op_sel"] and i < len(sel["op_sel"]) else 0
        b = sel["op_sel_hi"][i] if sel["op_sel_hi"] and i < len(sel["op_sel_hi"]) else 1
        out |= {lo + a, lo + b}
    return out


def split_blocks(lines):
    """basic blocks of one kernel: list of dicts {labels, insts [(ln, mnem, ops, text)], succ [label | index]}"""
    blocks = [{"labels": [], "insts": []}]
    for ln, raw in lines:
        text = raw.split(";")[0].strip()
        if not text or text.startswith("//") or (text.startswith(".") and not text.endswith(":")):
            continue
        if text.endswith(":"):
            if blocks[-1]["insts"]:
                blocks.append({"labels": [], "insts": []})
            blocks[-1]["labels"].append(text[:-1])
            continue
        parts = text.split(None, 1)
        mnem, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        blocks[-1]["insts"].append((ln, mnem, ops, text))
        if mnem.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            blocks.append({"labels": [], "insts": []})
    index = {lab: i for i, b in enumerate(blocks) for lab in b["labels"]}
    for i, b in enumerate(blocks):
        succ = []
        last = b["insts"][-1] if b["insts"] else None
        if last and last[1].startswith("s_cbranch"):
            succ = [index.get(last[2].strip()), i + 1]
        elif last and last[1] == "s_branch":
            succ = [index.get(last[2].strip())]
        elif last and last[1].startswith(("s_endpgm", "s_setpc")):
            succ = []
        else:
            succ = [i + 1]
        b["succ"] = [x for x in succ if x is not None and x < len(blocks)]
    return blocks


def run_block(block, fifo, errors=None):
    """replay one basic block from the in-state `fifo`; returns the out-state"""
    for ln, mnem, ops, text in block["insts"]:
        if mnem == "s_waitcnt":
            m = WAIT.search(ops)
            if m:
                keep = int(m.group(1))
                fifo = () if keep == 0 else (fifo[-keep:] if len(fifo) > keep else fifo)
            elif "cnt" not in ops:
                try:
                    if int(ops, 0) == 0:
                        fifo = ()
                except ValueError:
                    pass
            continue
        if errors is not None:
            used = used_regs(mnem, ops)
            for dest, dln, dtext in fifo:
                hit = set(dest) & used
                if hit:
                    errors.append((ln, f"{text}   touches v{sorted(hit)[0]} of the LDS read still in flight",
                                   f"line {dln}: {dtext}"))
                    break
        if mnem.startswith("ds_"):
            fifo = fifo + ((tuple(sorted(lds_dest(mnem, ops))), ln, text),)
        elif mnem.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            fifo = fifo + (((), ln, text),)                      # scalar memory also counts in lgkmcnt
        # lgkmcnt is 4 bits: an entry older than the 16 youngest is released by ANY wait, so a
        # leading entry without a destination carries no information (keeps store loops finite)
        while len(fifo) > 16 and not fifo[0][0]:
            fifo = fifo[1:]
    return fifo


def covers(long, short):
    """every read outstanding in `short` is outstanding in `long` in the same place from the young end: whatever
    holds after replaying `long` holds after `short`"""
    if len(long) < len(short):
        return False
    if not any(f[0] for f in long) and not any(f[0] for f in short):
        return True                                # only stores / scalar loads pending: counts, no registers
    return not short or long[len(long) - len(short):] == short


MAX_STATES = 32


def merge(states, out):
    """in-states of a block (a tuple of alternatives, None = not reached yet) joined with one more incoming state:
    a state that another one covers is dropped.  Paths whose outstanding reads are unrelated (the compiler's own
    reads in the branches of compiler-managed code) are kept side by side and each is replayed."""
    if states is None:
        return (out,)
    if any(covers(st, out) for st in states):
        return states
    return tuple(st for st in states if not covers(out, st)) + (out,)


def check_kernel(name, lines):
    """forward data-flow over the kernel's control-flow graph to a fixed point, then one checking pass"""
    errors = []
    blocks = split_blocks(lines)
    state = [None] * len(blocks)
    state[0] = ((),)
    work = [0]
    rounds = 0
    while work and not errors:
        rounds += 1
        if rounds > 100000:
            errors.append((0, "data-flow did not converge", ""))
            break
        i = work.pop()
        for st in state[i]:
            out = run_block(blocks[i], st)
            if len(out) > 64:
                errors.append((blocks[i]["insts"][0][0], "LDS operations pile up along a loop (never awaited)", ""))
                break
            for j in blocks[i]["succ"]:
                m = merge(state[j], out)
                if len(m) > MAX_STATES:
                    ln = blocks[j]["insts"][0][0] if blocks[j]["insts"] else 0
                    errors.append((ln, "too many unrelated sets of outstanding LDS reads reach " + "/".join(blocks[j]["labels"]), ""))
                    break
                if m != state[j]:
                    state[j] = m
                    if j not in work:
                        work.append(j)
            if errors:
                break
    if not errors:
        for b, sts in zip(blocks, state):
            for st in sts or ():
                run_block(b, st, errors)
        errors = sorted(set(errors))
    return errors


def main():
    args = [a for a in sys.argv[1:] if a != "--allow-none"]
    allow_none = "--allow-none" in sys.argv[1:]          # translation units without asm-LDS kernels
    path = args[0]
    wanted = tuple(args[1:]) or KERNELS
    kernels = {}
    cur = None
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            if cur is None:
                m = re.match(r"^(_Z\w+):", raw)
                if m and any(re.search(k, m.group(1)) for k in wanted):
                    cur = m.group(1)
                    kernels[cur] = []
                continue
            kernels[cur].append((ln, raw))
            if "s_endpgm" in raw.split(";")[0]:
                cur = None
    if not kernels:
        print("check_lds_hazards: no asm-LDS kernel found in", path)
        return 0 if allow_none else 1
    bad = 0
    for name, lines in kernels.items():
        errs = check_kernel(name, lines)
        for ln, what, where in errs[:5]:
            print(f"LDS HAZARD in {name}\n  line {ln}: {what}\n  {where}", flush=True)
        bad += len(errs)
    if bad:
        print(f"check_lds_hazards: {bad} hazard(s) in {len(kernels)} kernels")
        return 1
    print(f"no LDS read hazards in {len(kernels)} asm-LDS kernels")
    return 0


if __name__ == "__main__":
    sys.exit(main())
