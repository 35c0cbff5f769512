"""Static instruction mix of one kernel in a hipcc -save-temps .s file, per basic block.

    python tools/isa_count.py file.s <substring of the mangled kernel name> [--blocks]
"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "MFMA"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\w+:", ln) and key in ln.split(":")[0]:
            start = i
            break
    assert start is not None, "kernel not found"
    total = {}
    blocks = []
    cur, cur_name = {}, "entry"
    for ln in lines[start + 1:]:
        t = ln.strip()
        if t.startswith("s_endpgm"):
            break
        if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("; %bb."):
            blocks.append((cur_name, cur))
            cur, cur_name = {}, t.split(":")[0].replace("; %", "")
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        cur[c] = cur.get(c, 0) + 1
        total[c] = total.get(c, 0) + 1
    blocks.append((cur_name, cur))
    print("total:", dict(sorted(total.items())))
    if show_blocks:
        for name, c in blocks:
            if sum(c.values()) >= 8:
                print("%-12s %s" % (name, dict(sorted(c.items()))))


if __name__ == "__main__":
    main()
