"""Instruction mix of the steady-state loop of an igemm_kernel instantiation (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only \
          selavi_amd/csrc/conv_fwd.hip -o /tmp/conv_fwd.s
    python tools/isa_loop_mix.py /tmp/conv_fwd.s 'ILi0ELi9ELi2ELb1ELi1ELi1ELi0ELi0ELi0E'   # MODE,MT,NT,VA,PRO,KORD,EPI,MF,VB

Prints, for every kernel whose mangled name contains the pattern, the basic block with the most MFMAs: counts per
instruction class, the most frequent VALU/SALU opcodes, registers / LDS / scratch of the kernel."""
import collections
import re
import sys


def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_load", "s_buffer")): return "smem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    return "other"


def main(path, pattern):
    txt = open(path).read()
    lines = txt.split("\n")
    meta = {m.group(1): m.group(2) for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S)}
    for name, body_meta in meta.items():
        if pattern not in name or ("igemm_kernel" not in name and "igemm3" not in name and "conv_cl16_kernel" not in name and "conv_cl16_s3_kernel" not in name and "cl16_wgrad_kernel" not in name):
            continue
        start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        blocks, cur, label = [], [], "entry"
        for l in lines[start:end]:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                blocks.append((label, cur)); cur = []; label = m.group(1)
            else:
                cur.append(l)
        blocks.append((label, cur))
        best = None
        for label, ls in blocks:
            ops = [l.split()[0] for l in ls if l.startswith("\t") and not l.strip().startswith((".", ";")) and l.split()]
            c = collections.Counter(cat(o) for o in ops)
            if c["mfma"] and (best is None or c["mfma"] > best[1]["mfma"]):
                best = (label, c, ops)
        g = lambda k: re.search(k + r" (\d+)", body_meta).group(1)
        print(name)
        print("  vgpr", g("next_free_vgpr"), "sgpr", g("next_free_sgpr"), "lds", g("group_segment_fixed_size"),
              "scratch", g("private_segment_fixed_size"))
        if best:
            label, c, ops = best
            print("  loop block", label, dict(c), "total", len(ops))
            oc = collections.Counter(o for o in ops if cat(o) in ("valu", "salu"))
            print("  top valu/salu:", oc.most_common(10))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
