"""Instruction mix of a kernel's loops from hipcc's -save-temps assembly (tools/isa_mix.py file.s kernel-substring): per basic-block range
between labels, counts of MFMA / VALU / trans / SALU / s_waitcnt / s_nop / LDS / VMEM / branch instructions.  A CPU-side aid for
sizing an item loop's issue slots against its MFMAs before anything runs on the GPU."""
import re
import sys

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op in ("v_exp_f32_e32", "v_exp_f32", "v_log_f32_e32", "v_rcp_f32_e32", "v_rsq_f32_e32", "v_sqrt_f32_e32"): return "trans"
    if op.startswith("v_"): return "valu"
    if op == "s_waitcnt": return "wait"
    if op == "s_nop": return "nop"
    if op == "s_barrier": return "barrier"
    if op.startswith("s_cbranch") or op == "s_branch": return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    return "other"

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0]:
            start = i; break
    assert start is not None, "kernel not found"
    blocks, cur, name = [], {}, "entry"
    order = []
    for l in lines[start + 1:]:
        t = l.split(";")[0].strip() if not l.strip().startswith(";") else ""
        if t.startswith("s_endpgm"): break
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append((name, cur)); name = t.split(":")[0]; cur = {}
            continue
        if not t or t.startswith((";", ".")): continue
        op = t.split()[0]
        c = classify(op)
        cur[c] = cur.get(c, 0) + 1
    blocks.append((name, cur))
    cols = ["mfma", "valu", "trans", "salu", "wait", "nop", "lds", "vmem", "branch", "barrier", "other"]
    print("%-12s " % "block" + " ".join("%6s" % c for c in cols))
    tot = {}
    for n, b in blocks:
        if sum(b.values()) == 0: continue
        print("%-12s " % n + " ".join("%6d" % b.get(c, 0) for c in cols))
        for c in cols: tot[c] = tot.get(c, 0) + b.get(c, 0)
    print("%-12s " % "TOTAL" + " ".join("%6d" % tot.get(c, 0) for c in cols))

main()
