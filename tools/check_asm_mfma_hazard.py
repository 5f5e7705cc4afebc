"""Static check of a hipcc --offload-device-only -S listing: inline-asm VALU writes (between ;;#ASMSTART/;;#ASMEND) are
not covered by hipcc's hazard recogniser, so one that overwrites a register an MFMA issued fewer than 8 wait states
earlier still reads as its C operand would corrupt that MFMA (gfx90a+: 7 wait states for the 8-pass 16x16x4 fp32 MFMA).
usage: python tools/check_asm_mfma_hazard.py <file.s> [kernel name substring]   (exit code 1 when an exposure is found)"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main(path, want=""):
    s = open(path).read()
    bad = 0
    for m0 in re.finditer(r"^(\S+):\s*; @\1|^(_Z\S+):", s, re.M):
        name = m0.group(1) or m0.group(2)
        if want not in name or "Lfunc_end" in name:
            continue
        end = s.find(".Lfunc_end", m0.end())
        body = [l.strip() for l in s[m0.end():end].split("\n")]
        inasm, recent, n_asm, n_mfma, renamed = False, [], 0, 0, 0  # recent: (wait states ago, C registers)
        for l in body:
            if l.startswith(";;#ASMSTART"):
                inasm = True
                continue
            if l.startswith(";;#ASMEND"):
                inasm = False
                continue
            if not l or l[0] in ";.":
                continue
            ops = [t.strip() for t in re.split(r"[ ,]+", l, maxsplit=5)]
            if l.startswith("v_mfma"):
                n_mfma += 1
                parts = [t.strip() for t in l.split(None, 1)[1].split(",")]
                d, c = regs(parts[0]), regs(parts[3].split()[0])
                renamed += d != c
                recent = [(a + 1, r) for a, r in recent if a + 1 < 8] + [(0, c)]
                continue
            if inasm and l.startswith("v_"):
                n_asm += 1
                dst = regs(ops[1])
                for ago, c in recent:
                    if dst & c:
                        bad += 1
                        print("%s: '%s' writes v%s, read as C by an MFMA %d wait states earlier" % (name[-40:], l[:60], sorted(dst & c), ago))
            m = re.match(r"s_nop (\d+)", l)
            step = int(m.group(1)) + 1 if m else (16 if l.startswith("s_waitcnt") else 1)
            recent = [(a + step, r) for a, r in recent if a + step < 8]
        if n_mfma:
            print("%-60s %4d MFMAs (%d with D != C), %3d inline-asm VALU ops" % (name[-60:], n_mfma, renamed, n_asm))
    print("exposures: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""))
