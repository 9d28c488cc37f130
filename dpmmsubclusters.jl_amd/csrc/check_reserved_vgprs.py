#!/usr/bin/env python3
"""Build-time check of the lean kernels' touches (rewritten in round 6).

niw_lean_kernel / niw_sub_kernel touch the next tile's x rows behind the compiler's back (inline asm; its vmcnt bookkeeping must not see them).
Round 5 loaded them into v254 / v255 and relied on amdgpu_num_vgpr(254) to keep the allocator away; round 6's first change that added register
pressure made the allocator use those registers, and this script -- then checking exactly that -- stopped the build.  The touches now have no
register destination (global_load_lds_dword: M0 + 4 * lane in LDS), so what has to hold in the generated code is:

  * every global_load_lds_dword of niw_lean.hip sits between `s_mov_b32 sN, m0` (save) and `s_mov_b32 m0, sN` (restore, the same sN), with
    nothing in between but the M0 write, s_nop and the touches themselves: the compiler's own uses of M0 never see the sink's address;
  * no load names a fixed high register as in the old scheme (v254 / v255 as a destination of global_load_dword).

    check_reserved_vgprs.py <device assembly .s>        exit status 1 (with the offending lines) if the property does not hold
"""
import re
import sys


def offending(lines):
    """(bad lines, number of touches).  Walks the assembly; an LDS-DMA touch outside a save ... restore window of M0 is bad."""
    bad, touches = [], 0
    saved = None              # the SGPR holding M0 inside a window, else None
    for no, raw in enumerate(lines, 1):
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
            touches += 1
            if saved is None:
                bad.append((no, raw.rstrip()))
            continue
        if saved is not None and not line.startswith("s_nop"):
            bad.append((no, raw.rstrip()))          # something else inside the window: it could read or write M0
            saved = None
        if re.match(r"global_load_dword\s+v25[45],", line):
            bad.append((no, raw.rstrip()))          # the old scheme
    if saved is not None:
        bad.append((len(lines), "M0 saved and never restored"))
    return bad, touches


def main():
    lines = open(sys.argv[1]).read().split("\n")
    bad, touches = offending(lines)
    if touches == 0:
        print("check_reserved_vgprs: no LDS-DMA touch found -- the check does not apply any more (remove it together with the touches)", file=sys.stderr)
        return 1
    if bad:
        print(f"check_reserved_vgprs: {len(bad)} line(s) break the touches' M0 save / restore window:", file=sys.stderr)
        for no, l in bad[:20]:
            print(f"  line {no}: {l}", file=sys.stderr)
        return 1
    print(f"check_reserved_vgprs: ok ({touches} LDS-DMA touches, each inside a save / restore of M0)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
