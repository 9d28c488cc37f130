#!/usr/bin/env python3
"""Build-time check of the lean kernels' reserved registers.

niw_lean_kernel / niw_sub_kernel touch the next tile's x rows with inline-asm `global_load_dword v254 / v255`: two registers nothing reads, loaded
behind the compiler's back so that no wait is ever emitted for them.  That is only safe while the register allocator never places a value of
its own in v254 / v255 -- a touch returning late would overwrite it.  `amdgpu_num_vgpr(254)` asks for that; this script PROVES it on the
generated code: in the device assembly of niw_lean.hip no instruction other than the touches names v254 or v255 (alone or inside a register
range), and no kernel that contains a touch spills vector registers around it unnoticed (reported, not fatal).

    check_reserved_vgprs.py <device assembly .s>        exit status 1 (with the offending lines) if the property does not hold
"""
import re
import sys

RESERVED = (254, 255)


def offending(lines):
    bad, touches = [], 0
    single = re.compile(r"\bv(\d+)\b")
    rng = re.compile(r"\bv\[(\d+):(\d+)\]")
    for no, raw in enumerate(lines, 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith((".", "//")) or line.endswith(":"):
            continue
        regs = {int(m.group(1)) for m in single.finditer(line)}
        for m in rng.finditer(line):
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        hit = [r for r in RESERVED if r in regs]
        if not hit:
            continue
        m = re.match(r"global_load_dword\s+v(\d+),\s*v\[\d+:\d+\],\s*off\s*$", line)
        if m and int(m.group(1)) in RESERVED:
            touches += 1
            continue
        bad.append((no, raw.rstrip()))
    return bad, touches


def main():
    lines = open(sys.argv[1]).read().split("\n")
    bad, touches = offending(lines)
    if touches == 0:
        print("check_reserved_vgprs: no touch load found -- the check does not apply any more (remove it together with the touches)", file=sys.stderr)
        return 1
    if bad:
        print(f"check_reserved_vgprs: v254 / v255 are used outside the {touches} touch loads -- a late touch could overwrite a live value:", file=sys.stderr)
        for no, l in bad[:20]:
            print(f"  line {no}: {l}", file=sys.stderr)
        return 1
    print(f"check_reserved_vgprs: ok ({touches} touch loads, v254 / v255 named nowhere else)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
