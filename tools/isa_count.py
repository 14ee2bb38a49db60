#!/usr/bin/env python3
"""Static instruction mix of one kernel of a device-only assembly file (hipcc --cuda-device-only -S), per basic block:
    python tools/isa_count.py /tmp/isa/dsss_pg.s pg_segment_kernel
Prints the register / LDS figures of the kernel descriptor and, per label, the instruction classes -- the loop bodies are the
blocks a backward branch targets.  Used to price an issue-bound kernel without a GPU."""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
text = open(path).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l))
end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_") and "f64" in op: return "f64"
    if op.startswith("v_pk_"): return "pk"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur, order = collections.defaultdict(collections.Counter), "entry", ["entry"]
targets = collections.Counter()
for l in text[start + 1:end + 1]:
    s = l.strip()
    if not s or (s.startswith(("//", ".")) and not s.startswith(".LBB")): continue
    m = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):", s)
    if m:
        cur = m.group(1); order.append(cur); continue
    if s.startswith(";"): continue
    op = s.split()[0]
    blocks[cur][cls(op)] += 1
    if op.startswith("s_cbranch") or op == "s_branch":
        targets[s.split()[-1]] += 1
tot = collections.Counter()
seen = set()
for i, b in enumerate(order):
    c = blocks[b]; n = sum(c.values())
    if n == 0: continue
    tot.update(c)
    back = any(t == b for t in targets) and b in order[:i + 1]
    print("%-14s %5d  %s" % (b, n, "  ".join("%s %d" % kv for kv in sorted(c.items()))))
print("total %d  %s" % (sum(tot.values()), dict(tot)))
for l in text[end:end + 140]:
    if re.search(r"next_free_vgpr|group_segment_fixed_size|private_segment_fixed_size|; (NumVgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)", l): print(l.strip())
