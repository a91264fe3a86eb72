"""Instruction mix of a kernel's main loop in the kept device assembly (bspy_amd/csrc/*.device.s).
usage: python tools/isa_mix.py <device .s> <mangled-name substring> [--dump]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, re.M)
name = m.group(1)
a = m.start()
b = s.index("s_endpgm", a)
lines = [l.split(";")[0].strip() for l in s[a:b].split("\n")]
labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":") and l.startswith(".LBB")}
loops = []
for i, l in enumerate(lines):
    mm = re.match(r"s_c?branch\w* (\.LBB\S+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        loops.append((labels[mm.group(1)], i))
lo, hi = max(loops, key=lambda x: x[1] - x[0])
c = collections.Counter()
for l in lines[lo:hi + 1]:
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    c[l.split()[0]] += 1
print(name)
print("loop lines", lo, hi, "VALU", sum(v for k, v in c.items() if k.startswith("v_")),
      "f64", sum(v for k, v in c.items() if k.startswith("v_") and "f64" in k),
      "DS", sum(v for k, v in c.items() if k.startswith("ds_")),
      "SALU", sum(v for k, v in c.items() if k.startswith("s_")))
for k, v in c.most_common():
    print(f"{v:4d} {k}")
if "--dump" in sys.argv:
    print("\n".join(lines[lo:hi + 1]))
