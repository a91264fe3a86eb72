"""Build-time guard: the kernels that read LDS through inline asm (eval_stream, jac_stream, eval_rowrot, jac_rowrot)
must not use scratch: a spilled asm destination would be stored before its data arrived.
Usage: python check_spills.py <resource-usage log>"""
import re
import sys

log = open(sys.argv[1]).read()
bad = []
for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", log, re.S):
    name, scratch = m.group(1), int(m.group(2))
    if "eval_slab2IdLi6" in name:       # fp64, order 6: plain C++ instantiation, no asm LDS reads (bsk_slab.hpp)
        continue
    if (any(k in name for k in ("eval_stream", "jac_stream", "eval_rowrot", "jac_rowrot", "curv_rowrot", "eval_uni", "jac_uni", "curv_uni", "eval_slab2", "eval_rec32"))) and scratch:
        bad.append((name, scratch))
if bad:
    for name, scratch in bad:
        print(f"SPILL in asm-LDS kernel {name}: {scratch} bytes/lane", file=sys.stderr)
    sys.exit(1)
print("no scratch in asm-LDS kernels")
