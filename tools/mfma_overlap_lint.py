"""Lint for the MFMA hazard round 3 ran into on gfx950 (EXPERIMENTS.md, "fused cross-attention sub-block"): a v_mfma that STARTS an
accumulator (SrcC = the inline constant 0) whose DESTINATION overlaps its own A or B operand, with the dependent next MFMA of that
accumulator (SrcC == that destination) issued right behind it, lost the second MFMA's contribution in part of the destination registers (timing dependent).  Kernels whose
overlapping MFMAs are followed by at least `--distance` other MFMAs before the dependent one have been parity-green for three
rounds (ffblock.hip), and so have pairs with dozens of other instructions in between (ffblock.hip: one other MFMA
+ a ~60-instruction VALU block); the failing pair was 7 instructions apart with no MFMA in between.  The lint flags only pairs that
are close in BOTH measures (fewer than --distance MFMAs and fewer than --instructions instructions in between).

INFORMATIONAL, not a gate: attention.hip contains 98 such pairs and is parity-green - the pattern is necessary for what was
seen, not sufficient (the fourth ingredient is not isolated).  Findings at the end of round 3: xattn.hip 0 (9 in its un-fixed
form), ffblock.hip 0, attention.hip 98.

  python tools/mfma_overlap_lint.py sketch2img_amd/csrc/xattn.hip [--distance 2]      # compiles with hipcc -S, prints findings
"""
import argparse
import re
import subprocess
import sys
import tempfile

MFMA = re.compile(r"^\s*(v_mfma_\w+)\s+(\S+),\s*(\S+),\s*(\S+),\s*(\S+)")


def regs(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return tok[0], int(m.group(1)), int(m.group(2))
    m = re.match(r"([va])(\d+)$", tok)
    return (m.group(1), int(m.group(2)), int(m.group(2))) if m else None


def overlap(a, b):
    return a is not None and b is not None and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


def lint_asm(text, distance=2, instructions=24):
    """-> list of (kernel, line number, overlapping MFMA, dependent MFMA, MFMAs in between)."""
    out, kernel, mf, icount = [], "?", [], 0
    for ln, line in enumerate(text.splitlines(), 1):
        st = line.strip()
        if st and not st.startswith((";", ".")) and not st.endswith(":"):
            icount += 1
        if re.match(r"^[\w.$]+:\s*(;.*)?$", line) and not line.startswith(".L"):
            kernel, mf = line.split(":")[0], []
        if line.lstrip().startswith(("s_barrier", "s_cbranch", "s_branch")) or line.startswith(".L"):
            mf = []                                   # only straight-line neighbourhoods are judged
        m = MFMA.match(line)
        if not m:
            continue
        dst, a, b, c = (regs(m.group(i)) for i in (2, 3, 4, 5))
        for back, (pln, pdst, pline, pic) in enumerate(reversed(mf[-distance:])):
            if overlap(c, pdst) and c == pdst and icount - pic - 1 < instructions:
                out.append((kernel, pln, pline.strip(), line.strip(), back))
        # the failing instruction had an INLINE CONSTANT as SrcC (a fresh accumulator): with a register SrcC the same
        # overlap + close dependent pair is all over attention.hip (252 of them) and parity-green
        fresh = c is None
        mf.append((ln, dst if (fresh and (overlap(dst, a) or overlap(dst, b))) else None, line, icount))
    return out


def compile_to_asm(src, extra=()):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-S",
                        "--cuda-device-only", *extra, src, "-o", f.name], check=True, capture_output=True)
        return open(f.name).read()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--distance", type=int, default=2)
    ap.add_argument("--instructions", type=int, default=24)
    a = ap.parse_args()
    found = lint_asm(compile_to_asm(a.src), a.distance, a.instructions)
    for k, ln, first, second, back in found:
        print(f"{k}: line {ln}: {first}\n    dependent {back} MFMA(s) later: {second}")
    print(f"{len(found)} close dependent MFMA(s) behind a destination / operand overlap")
    sys.exit(1 if found else 0)


if __name__ == "__main__":
    main()
