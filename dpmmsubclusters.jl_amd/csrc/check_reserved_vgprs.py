#!/usr/bin/env python3
"""Build-time check of the lean kernels' touches (per kernel since round 6).

niw_lean_kernel / niw_sub_kernel touch the next tile's x rows behind the compiler's back (inline asm; its vmcnt bookkeeping must not see them):
`global_load_dword v254 / v255`, two registers nothing reads.  Safe only while the register allocator never places a value of its own there -- a
touch returning late would overwrite it.  amdgpu_num_vgpr(254) asks for that and is honoured only while the kernel fits: round 6's first
change that added pressure made the allocator use them, and this script stopped the build (the kernel that needs the whole register file,
niw_lean_kernel_dir, has no touch since).  Property checked, per kernel of the assembly: in a kernel WITH register touches no other instruction
names v254 or v255 (alone or inside a register range).

An experimental second flavour is recognised as well (docs/experiments/r06_lds_dma_touch.patch: `global_load_lds_dword`, no register destination):
every such load must sit between `s_mov_b32 sN, m0` (save) and `s_mov_b32 m0, sN` (restore, the same sN) with nothing in between but the M0
write, s_nop and the touches themselves.

    check_reserved_vgprs.py <device assembly .s>        exit status 1 (with the offending lines) if a property does not hold
"""
import re
import sys

RESERVED = (254, 255)


def kernels(lines):
    """[(name, first line number, body lines)] of the functions of a device assembly (`name:` ... s_endpgm / s_setpc_b64)."""
    out, name, start, body = [], None, 0, []
    for no, raw in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", raw)
        if m and name is None:
            name, start, body = m.group(1), no, []
            continue
        if name is not None:
            body.append(raw)
            if re.match(r"\s*(s_endpgm|s_setpc_b64)", raw):
                out.append((name, start, body)); name = None
    return out


def offending(lines, first=1):
    """(bad [(line number, text)], register touches, LDS touches) of ONE kernel's lines (or of a fragment, for the tests)."""
    bad, reg_t, lds_t = [], 0, 0
    single = re.compile(r"\bv(\d+)\b")
    rng = re.compile(r"\bv\[(\d+):(\d+)\]")
    saved = None              # the SGPR holding M0 inside a window, else None
    named = []                # lines naming a reserved register outside a touch
    for off, raw in enumerate(lines):
        no = first + off
        line = raw.split(";")[0].strip()
        if not line or line.startswith((".", "//")) or line.endswith(":"):
            continue
        m = re.match(r"s_mov_b32\s+(s\d+),\s*m0$", line)
        if m:
            saved = m.group(1)
            continue
        m = re.match(r"s_mov_b32\s+m0,\s*(\S+)$", line)
        if m and saved is not None:
            if m.group(1) == saved:
                saved = None              # restored: window closed
            continue                      # (the sink's base)
        if line.startswith("global_load_lds_dword"):
            lds_t += 1
            if saved is None:
                bad.append((no, raw.rstrip()))
            continue
        if saved is not None and not line.startswith("s_nop"):
            bad.append((no, raw.rstrip()))          # something else inside the window: it could read or write M0
            saved = None
        regs = {int(x.group(1)) for x in single.finditer(line)}
        for x in rng.finditer(line):
            regs.update(range(int(x.group(1)), int(x.group(2)) + 1))
        if any(r in regs for r in RESERVED):
            m = re.match(r"global_load_dword\s+v(\d+),\s*v\[\d+:\d+\],\s*off\s*$", line)
            if m and int(m.group(1)) in RESERVED:
                reg_t += 1
            else:
                named.append((no, raw.rstrip()))
    if saved is not None:
        bad.append((first + len(lines), "M0 saved and never restored"))
    if reg_t:
        bad += named                               # a kernel WITH register touches must not name them anywhere else
    return bad, reg_t, lds_t


def main():
    lines = open(sys.argv[1]).read().split("\n")
    total_bad, report = [], []
    for name, start, body in kernels(lines):
        bad, reg_t, lds_t = offending(body, start + 1)
        if reg_t or lds_t:
            report.append(f"{name[:48]}: {reg_t} register + {lds_t} LDS touches")
        total_bad += [(name, no, l) for no, l in bad]
    if not report:
        print("check_reserved_vgprs: no touch found -- the check does not apply any more (remove it together with the touches)", file=sys.stderr)
        return 1
    if total_bad:
        print(f"check_reserved_vgprs: {len(total_bad)} line(s) break the touches' invariants (v254 / v255 named outside the register touches of a kernel, or an "
              "LDS touch outside an M0 save / restore window):", file=sys.stderr)
        for name, no, l in total_bad[:20]:
            print(f"  {name[:40]} line {no}: {l}", file=sys.stderr)
        return 1
    print("check_reserved_vgprs: ok (" + "; ".join(report) + ")")
    return 0


if __name__ == "__main__":
    sys.exit(main())
