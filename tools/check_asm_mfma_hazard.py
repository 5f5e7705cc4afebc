"""Static check of a hipcc --offload-device-only -S listing: inline-asm VALU writes (between ;;#ASMSTART/;;#ASMEND) are
not covered by hipcc's hazard recogniser, so one that overwrites a register an MFMA issued fewer than 8 wait states
earlier still reads as its C operand would corrupt that MFMA (gfx90a+: 7 wait states for the 8-pass 16x16x4 fp32 MFMA).

Conservative on purpose: an ``s_waitcnt`` counts as ONE wait state (one whose counters are already satisfied issues in
a cycle), and every backward branch is followed once -- the body of a loop is replayed with the state its bottom leaves
behind, so an MFMA at the end of an iteration is checked against the inline-asm ops at the top of the next.

usage: python tools/check_asm_mfma_hazard.py <file.s> [kernel name substring]   (exit code 1 when an exposure is found)"""
import re
import sys

WINDOW = 8  # wait states an MFMA's SrcC stays exposed


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


class _State:
    def __init__(self):
        self.inasm = False
        self.recent = []   # (wait states ago, C registers) of the MFMAs still inside the window
        self.n_asm = self.n_mfma = self.renamed = 0
        self.found = []


def _step(l, st, name, count=True):
    """advance the state by one listing line; count=False on a replay (statistics are per static instruction)"""
    if l.startswith(";;#ASMSTART"):
        st.inasm = True
        return
    if l.startswith(";;#ASMEND"):
        st.inasm = False
        return
    if not l or l[0] in ";." or l.endswith(":"):
        return
    if l.startswith("v_mfma"):
        parts = [t.strip() for t in l.split(None, 1)[1].split(",")]
        d, c = regs(parts[0]), regs(parts[3].split()[0])
        if count:
            st.n_mfma += 1
            st.renamed += d != c
        st.recent = [(a + 1, r) for a, r in st.recent if a + 1 < WINDOW] + [(0, c)]
        return
    if st.inasm and l.startswith("v_"):
        ops = [t.strip() for t in re.split(r"[ ,]+", l, maxsplit=5)]
        if count:
            st.n_asm += 1
        dst = regs(ops[1])
        for ago, c in st.recent:
            if dst & c:
                st.found.append("%s: '%s' writes v%s, read as C by an MFMA %d wait states earlier"
                                % (name[-40:], l[:60], sorted(dst & c), ago))
    m = re.match(r"s_nop (\d+)", l)
    step = int(m.group(1)) + 1 if m else 1   # s_waitcnt included: 1
    st.recent = [(a + step, r) for a, r in st.recent if a + step < WINDOW]


def check_body(body, name):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):(\s*;.*)?$", l)   # ".LBB0_3:        ; =>This Inner Loop Header"
        if m:
            labels[m.group(1)] = i
    st = _State()
    for i, l in enumerate(body):
        _step(l, st, name)
        m = re.match(r"s_c?branch\w*\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            # back edge: replay the loop body once with the state the bottom of the loop leaves behind
            saved = (st.inasm, list(st.recent))
            for k in range(labels[m.group(1)], i + 1):
                _step(body[k], st, name, count=False)
            st.inasm, st.recent = saved
    return st


def main(path, want=""):
    s = open(path).read()
    bad = 0
    for m0 in re.finditer(r"^(\S+):\s*; @\1|^(_Z\S+):", s, re.M):
        name = m0.group(1) or m0.group(2)
        if want not in name or "Lfunc_end" in name:
            continue
        end = s.find(".Lfunc_end", m0.end())
        body = [l.strip() for l in s[m0.end():end].split("\n")]
        st = check_body(body, name)
        for line in sorted(set(st.found)):
            print(line)
        bad += len(set(st.found))
        if st.n_mfma:
            print("%-60s %4d MFMAs (%d with D != C), %3d inline-asm VALU ops" % (name[-60:], st.n_mfma, st.renamed, st.n_asm))
    print("exposures: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""))
