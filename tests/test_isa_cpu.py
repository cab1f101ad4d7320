"""Code-object checks that need no GPU: properties of the compiled gfx950 ISA that a measured hardware hazard depends on."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


@pytest.mark.skipif(not os.path.exists(LLVM + "llvm-objdump"), reason="no ROCm LLVM tools")
def test_no_wide_buffer_store_with_sgpr_offset():
    """common.h (buffer_store_b128): on MI355X a `buffer_store_dwordx4 ... offen` with an SGPR offset directly followed by a VALU write
    of its data registers stored the NEW value in part of the lanes at full occupancy (tools/probe/diag_attn64b.py) -- LLVM exempts
    exactly that form from its wait state.  No 96/128-bit MUBUF store of the library may carry an SGPR offset."""
    import __graft_entry__ as G
    G.build()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_store_hazard.py"), "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"96/128-bit buffer stores with an SGPR offset: (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    assert int(m.group(1)) == 0, out.stdout[-3000:]
    # and no buffer store is directly followed by a VALU write of its data (distance 1)
    direct = [l for l in out.stdout.splitlines() if "buffer_store_dwordx" in l and "<- +1: v_" in l]
    assert not direct, "\n".join(direct[:10])
