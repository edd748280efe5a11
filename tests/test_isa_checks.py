"""Static ISA checks of hand-scheduled kernels (no GPU: hipcc cross-compiles gfx950 to assembly here).

csrc/conv_cl16_tr.hip issues its fragment reads as inline-asm ds_read_b128 with a hand-placed s_waitcnt lgkmcnt(0): correct
only while the compiler neither copies nor spills nor schedules a use of the destination registers in front of that wait
(ADVICE r5).  tools/tr_asm_check.py walks the generated assembly of every instantiation."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tr_kernel_fragment_reads_are_not_touched_before_their_wait():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tr_asm_check.py"), "--build"], capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK 0"), p.stdout[-3000:] + p.stderr[-2000:]
    assert "60 hand-issued ds_read_b128" in p.stdout          # the layer-1 temporal instantiations are among those checked
