"""Static check of the hand-issued fragment reads of csrc/conv_cl16_tr.hip (no GPU needed): tr_lds_read is an
inline-asm `ds_read_b128 v[a:a+3], ...` whose completion the compiler cannot see -- the kernel's own
`s_waitcnt lgkmcnt(0)` (inline asm as well) is the only thing between the request and the first use.  Between such a
request and the next lgkmcnt(0) wait no instruction may name one of its destination registers: a copy, a spill or an MFMA
scheduled in between would read a register whose load has not landed.

    hipcc -S --offload-arch=gfx950 --cuda-device-only <build flags> selavi_amd/csrc/conv_cl16_tr.hip -o /tmp/tr.s
    python tools/tr_asm_check.py /tmp/tr.s [name-pattern]          (or: python tools/tr_asm_check.py --build)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build():
    sys.path.insert(0, ROOT)
    from selavi_amd.build import FLAGS
    out = "/tmp/slv_tr_check.s"
    src = os.path.join(ROOT, "selavi_amd", "csrc", "conv_cl16_tr.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S", src, "-o", out])
    return out


def regs_of(s):
    regs = set(int(r) for r in re.findall(r"\bv(\d+)\b", s))
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", s):
        regs |= set(range(int(a), int(b) + 1))
    return regs


def check(path, pat="conv_cl16_tr"):
    txt = open(path).read().split("\n")
    names = [m.group(1) for l in txt for m in [re.match(r"^(_ZN3slv\S+):", l)] if m and pat in m.group(1)]
    bad = 0
    for name in names:
        start = next(i for i, l in enumerate(txt) if l.startswith(name + ":"))
        end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith(".Lfunc_end"))
        pending = {}            # register number -> line of the request
        in_asm, nreq, nwait = False, 0, 0
        for i in range(start, end):
            raw = txt[i].strip()
            if raw.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if raw.startswith(";;#ASMEND"):
                in_asm = False
                continue
            s = raw.split(";")[0].strip()
            if not s or s.startswith(".") or s.endswith(":"):
                continue
            m = re.match(r"ds_read_b128 v\[(\d+):(\d+)\], (v\d+)", s)
            if m and in_asm:
                dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
                addr = int(m.group(3)[1:])
                if addr in pending:
                    print(f"{name}: address v{addr} of a read is itself an un-waited destination: line {i + 1}: {s}"); bad += 1
                hit = dst & set(pending)
                if hit:
                    print(f"{name}: destination {sorted(hit)} requested again before its wait: line {i + 1}: {s}"); bad += 1
                for r in dst:
                    pending[r] = i + 1
                nreq += 1
                continue
            if s.startswith("s_waitcnt") and "lgkmcnt(0)" in s:
                nwait += 1
                pending.clear()
                continue
            if pending:
                hit = regs_of(s) & set(pending)
                if hit:
                    print(f"{name}: v{sorted(hit)} (requested line {min(pending[r] for r in hit)}) touched before the wait: "
                          f"line {i + 1}: {s}"); bad += 1
        print(f"{name}: {nreq} hand-issued ds_read_b128, {nwait} lgkmcnt(0) waits")
    print("BAD" if bad else "OK", bad)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        sys.exit(1 if check(build()) else 0)
    sys.exit(1 if check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "conv_cl16_tr") else 0)
