"""Build-time check of the hand-written DPP statements in the solve (als_kernels.hip, fmac_bcast).

On gfx9 a VGPR written by a vector ALU instruction may be read through DPP two wait states later at the
earliest, and the hardware does not interlock this.  The compiler inserts the wait states for its own
instructions but does not look inside an asm statement, so a register copy that the register allocator
places directly in front of one of the asm `v_fmac_f32_dpp` statements would make it read a stale value.
This script compiles the kernels to assembly with the Makefile's flags and verifies, for every DPP
instruction (the hand-written `v_fmac_f32_dpp vD, vD, vS` and the compiler's own, which may read a
register that an asm statement wrote), that no vector ALU instruction writes the register read through
DPP in the two wait states before it, and no matrix-core instruction in the 18 before it.

usage: python tools/check_dpp_hazard.py [file.s ...]     (no arguments: compiles recommender_amd/csrc/*.hip)
exit status 0: clean, 1: hazards found (printed with their kernel and line)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "recommender_amd", "csrc")
SOURCES = ["als_kernels.hip", "als_cg.hip", "als_scoring.hip", "als_short_rows.hip"]


def makefile_flags():
    """The CXXFLAGS the library is built with: the environment's if set (make's `?=`), else the Makefile's own line,
    so that a build with other flags or -D switches is checked as built."""
    env = os.environ.get("CXXFLAGS")
    if env is not None:
        return env.split()
    for line in open(os.path.join(CSRC, "Makefile")):
        m = re.match(r"^CXXFLAGS\s*\?=\s*(.*)$", line)
        if m:
            return m.group(1).split()
    raise RuntimeError("no CXXFLAGS line in %s" % os.path.join(CSRC, "Makefile"))


MFMA_STATES = 20
_operand = re.compile(r"^v(\d+)\b|^v\[(\d+):(\d+)\]")


def _mfma_states(op):
    """Wait states a vector read must keep after this matrix-core instruction wrote the register."""
    passes = 16
    if "16x16x4_f32" in op or "16x16x4f32" in op or "32x32x16" in op:
        passes = 8
    elif "16x16x32" in op or "16x16x16" in op:
        passes = 4
    elif "4x4x" in op:
        passes = 2
    return min(MFMA_STATES, passes + 4)


class _Any:
    """Stands for "any register may have been written" (a control-flow join: the code in front is not the only
    predecessor)."""
    def __and__(self, other):
        return bool(other)
    __rand__ = __and__


def _written(operand):
    """VGPR numbers named by an operand; modifiers after the register (`v2 quad_perm:[1`) are ignored."""
    m = _operand.match(operand)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def check(text):
    """Returns a list of (kernel, line number, instruction, offending earlier instruction)."""
    bad = []
    kernel = "?"
    window = []          # (wait states this instruction provides, registers it writes, text, wait states a DPP read needs), newest last
    for no, raw in enumerate(text.split("\n"), 1):
        line = raw.split(";")[0].strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, window = m.group(1), []
            continue
        if re.match(r"^\.?\w+:$", line):
            # a label: branches arrive here from code this scan has not just seen (loop back-edges included), so
            # anything may have been written zero wait states ago
            window.append((0, _Any(), "label " + line, 2))
            continue
        if not line or line.startswith("."):
            continue
        parts = line.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        if op.endswith("_dpp") and len(args) > 1:
            src = _written(args[1])      # the operand read through DPP (for v_fmac_f32_dpp d, d, s the destination itself)
            states = 0
            for ws, wr, txt, need in reversed(window):
                if states >= MFMA_STATES:
                    break
                if wr & src:
                    if states < need:
                        bad.append((kernel, no, line, txt))
                    break
                states += ws
        if op == "s_nop":
            window.append((int(args[0], 0) + 1, set(), line, 0))
        elif op.startswith("v_mfma") and args:
            # a matrix-core result is not interlocked against a vector read either: passes + 4 wait states (the
            # figure LLVM's hazard recognizer uses for gfx950, rounded up), passes = issue cycles / 4
            window.append((1, _written(args[0]), line, _mfma_states(op)))
        elif op.startswith("v_") and args and not op.startswith("v_cmp") and not op.startswith("v_readlane") \
                and not op.startswith("v_readfirstlane"):
            window.append((1, _written(args[0]), line, 2))
        else:
            window.append((1, set(), line, 0))
        if len(window) > MFMA_STATES + 2:
            window.pop(0)
    return bad


def check_m0(text):
    """The LDS-DMA of the stream and score kernels is issued from inline assembly that writes m0 (the DMA's LDS
    address), a register reserved to the compiler, which is told so by a clobber it need not honour.  That is safe as
    long as the compiler itself never keeps a value in m0 in these files: every instruction that names m0 must be the
    `s_mov_b32 m0, <sgpr>` of such an assembly statement."""
    bad, in_asm = [], False
    for no, raw in enumerate(text.split("\n"), 1):
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
        elif line.startswith(";;#ASMEND"):
            in_asm = False
        elif re.search(r"\bm0\b", line.split(";")[0]) and not (in_asm and re.match(r"^s_mov_b32 m0, s\d+$", line.split(";")[0].strip())):
            bad.append((no, line))
    return bad


def assembly_of(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        flags = [f for f in makefile_flags() if f not in ("-shared", "-Wall") and not f.startswith("-Wno-")]
        subprocess.run([hipcc, "--offload-arch=gfx950", *flags, "-I" + os.path.join(ROOT, "include"), "-S",
                        "--cuda-device-only", "-o", out, src], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
        return open(out).read()


def main(argv):
    texts = [(p, open(p).read()) for p in argv] if argv else [(s, assembly_of(os.path.join(CSRC, s))) for s in SOURCES]
    total = n_dpp = 0
    for name, text in texts:
        n_dpp += len(re.findall(r"^\s*v_\w+_dpp\s", text, flags=re.M))
        for kernel, no, ins, prev in check(text):
            total += 1
            print(f"{name}:{no}: {kernel[:60]}: `{ins}` reads a register written by `{prev}` too few wait states before")
        for no, ins in check_m0(text):
            total += 1
            print(f"{name}:{no}: m0 used outside the LDS-DMA assembly statements: `{ins}`")
    print(f"{n_dpp} DPP instructions checked, {total} hazards")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
