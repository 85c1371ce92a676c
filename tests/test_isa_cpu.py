"""The built library must not contain packed-f32 VALU instructions (on gfx950 they read stale registers in lanes
48-63 when they consume freshly loaded data beside an MFMA-heavy wave of another kernel -- found as "the 128-column conv tile
corrupts its neighbours" in round 3, root-caused in round 4 with tools/canary.hip).  Runs on the CPU build box: disassembly only."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
LIB = os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd', 'lib', 'libvpmi.so')


@pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'), reason='needs llvm-objdump')
def test_library_has_no_packed_f32_instructions():
    import isa_scan
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    text = isa_scan.disassemble(LIB)
    assert text.count('v_mfma_') > 1000, 'disassembly looks empty'            # the scan saw real kernels
    hits, npk = isa_scan.scan(text)
    assert sum(npk.values()) == 0, f'packed-f32 VALU instructions in {dict(npk.most_common(5))}'


def test_scanner_recognises_the_pattern():
    import isa_scan
    text = '''0000000000001000 <kern>:
\tglobal_load_dwordx4 v[14:17], v[12:13], off
\tglobal_load_dwordx4 v[18:21], v[46:47], off
\ts_waitcnt vmcnt(1)
\tv_pk_fma_f32 v[2:3], v[14:15], v[46:47], v[2:3] op_sel_hi:[1,0,1]
\ts_waitcnt vmcnt(0)
\tv_pk_fma_f32 v[2:3], v[18:19], v[46:47], v[2:3]
'''
    hits, npk = isa_scan.scan(text)
    assert npk['kern'] == 2 and len(hits['kern']) == 1 and 'v[14:15]' in hits['kern'][0]
