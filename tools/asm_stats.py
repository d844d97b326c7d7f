"""Instruction counts by class for every kernel of an AMDGPU assembly file (hipcc -S --cuda-device-only).
usage: python tools/asm_stats.py file.s [name filter]"""
import collections
import re
import sys


def stats(text, flt=""):
    out = []
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end", text, flags=re.M | re.S):
        name, body = m.group(1), m.group(2)
        if flt not in name:
            continue
        c = collections.Counter()
        n = 0
        for line in body.split("\n"):
            line = line.split(";")[0].strip()
            if not line or line.startswith(".") or line.endswith(":"):
                continue
            o = line.split()[0]
            n += 1
            if o.startswith("v_mfma"): c["mfma"] += 1
            elif o.startswith("v_accvgpr"): c["accvgpr"] += 1
            elif "dpp" in o: c["dpp"] += 1
            elif o.startswith("v_"): c["valu"] += 1
            elif o.startswith("s_waitcnt"): c["waitcnt"] += 1
            elif o.startswith("s_nop"): c["nop"] += 1
            elif o.startswith("s_"): c["salu"] += 1
            elif o.startswith("ds_"): c["lds"] += 1
            elif o.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
            else: c["other"] += 1
        out.append((name, n, dict(c)))
    return out


if __name__ == "__main__":
    for name, n, c in stats(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else ""):
        print("%-80s %6d %s" % (name[-80:], n, c))
