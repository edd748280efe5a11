"""Static check of the hand-counted B requests of igemm3.hpp (no GPU needed): between an inline-asm `buffer_load_dword vN`
(not the LDS-DMA form) and the next `s_waitcnt vmcnt(K)` that is not compiler-staged, no instruction may name vN -- a
copy or spill of a register whose load has not been waited for reads garbage.  Usage: tools/x3_asm_check.py file.s [name-pattern]"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else "igemm3"
names = [m.group(1) for l in txt for m in [re.match(r"^(_ZN3slv\S+):", l)] if m and pat in m.group(1)]
bad = 0
for name in names:
    start = next(i for i, l in enumerate(txt) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith(".Lfunc_end"))
    pending = {}          # reg -> line of the request
    nreq = nwait = 0
    for i in range(start, end):
        s = txt[i].strip().split(";")[0]
        if not s or s.startswith("."):
            continue
        m = re.match(r"buffer_load_dword (v\d+), ", s)
        if m and " lds" not in s:
            regs_used = set(re.findall(r"\bv(\d+)\b", s.split(",", 1)[1]))
            for r in list(pending):
                if r[1:] in regs_used:
                    print(f"{name}: {r} (requested line {pending[r]}) used as address before its wait: line {i}: {s}"); bad += 1
            pending[m.group(1)] = i
            nreq += 1
            continue
        if s.startswith("s_waitcnt") and "vmcnt" in s:
            nwait += 1
            pending.clear()       # (every vmcnt wait in these kernels is one of the hand-counted ones; which registers it
            continue              #  covers is the kernel's invariant, not checked here)
        if pending:
            regs = set("v" + r for r in re.findall(r"\bv(\d+)\b", s))
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", s):
                regs |= set(f"v{k}" for k in range(int(a), int(b) + 1))
            hit = regs & set(pending)
            if hit:
                print(f"{name}: {sorted(hit)} touched before a wait: line {i}: {s}"); bad += 1
    print(f"{name}: {nreq} asm requests, {nwait} vmcnt waits")
print("BAD" if bad else "OK", bad)
