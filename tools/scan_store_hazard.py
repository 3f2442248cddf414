#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for the buffer-store data hazard: a VALU instruction that writes
a data register of a 16-byte buffer/global store within the next 2 wait states (LLVM does not guard buffer stores whose
soffset is an SGPR). usage: scan_store_hazard.py file.s ..."""
import re
import sys

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.fullmatch(tok.strip().rstrip(","))
    if not m:
        return set()
    if m.group(1):
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def main():
    total = bad = 0
    for path in sys.argv[1:]:
        lines = [l.strip() for l in open(path)]
        for i, l in enumerate(lines):
            if not (l.startswith("buffer_store_dwordx4") or l.startswith("buffer_store_dwordx3") or l.startswith("global_store_dwordx4") or l.startswith("global_store_dwordx3")):
                continue
            total += 1
            ops = l.split(None, 1)[1].split(",")
            data = regs(ops[0]) if l.startswith("buffer") else regs(ops[1])
            waits, j = 0, i + 1
            while waits < 2 and j < len(lines):
                t = lines[j]
                j += 1
                if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                    continue
                if t.startswith("s_nop"):
                    waits += int(t.split()[1]) + 1
                    continue
                if t.startswith("v_") and not t.startswith("v_cmp"):
                    dst = regs(t.split(None, 1)[1].split(",")[0])
                    if dst & data:
                        bad += 1
                        print(f"{path}:{i + 1}: {l}   <-  {t}")
                        break
                waits += 1
    print(f"16-byte stores: {total}, data register overwritten within 2 wait states: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
