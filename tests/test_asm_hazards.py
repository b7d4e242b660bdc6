"""The kernels' ISA must not hide a matrix-pipe hazard behind inline assembly (tools/asm_hazard_scan.py): the compiler
inserts the VALU -> MFMA and MFMA -> VALU wait states for its own instructions only, and a missing one passes every isolated kernel
test (the stale read needs another kernel on the chip to show).  Static, no GPU: hipcc -S of every MFMA-issuing source file."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('asm_hazard_scan', os.path.join(ROOT, 'tools', 'asm_hazard_scan.py'))
scan_mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scan_mod)

BROKEN = """
_Z6kernelv:
	v_mov_b32_e32 v3, v9
	;;#ASMSTART
	v_and_b32 v114, v114, v113
	;;#ASMEND
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_32x32x2_f32 v[0:15], v116, v114, v[0:15]
	v_add_u32_e32 v40, s7, v25
	;;#ASMSTART
	v_bfi_b32 v1, v77, v1, v3
	;;#ASMEND
	s_endpgm
"""
FIXED = """
_Z6kernelv:
	;;#ASMSTART
	v_bfe_i32 v113, v59, 0, 1
	;;#ASMEND
	v_and_b32_e32 v114, v114, v113
	s_nop 0
	v_mfma_f32_32x32x2_f32 v[0:15], v116, v114, v[0:15]
	s_nop 15
	s_nop 7
	;;#ASMSTART
	buffer_store_dword v1, v47, s[28:31], 0 offen
	;;#ASMEND
	s_endpgm
"""
# the hazard sits across a loop back-edge: the assembly store at the top of the body reads what the MFMA at the bottom of the previous trip wrote
LOOP = """
_Z6kernelv:
	s_mov_b32 s4, 0
.LBB0_1:
	;;#ASMSTART
	buffer_store_dword v1, v47, s[28:31], 0 offen
	;;#ASMEND
	s_add_i32 s4, s4, 1
	s_cmp_lt_i32 s4, 8
	v_mfma_f32_32x32x2_f32 v[0:15], v116, v114, v[0:15]
	s_cbranch_scc1 .LBB0_1
	s_endpgm
"""


def test_scanner_follows_back_edges(tmp_path):
    loop = tmp_path / 'loop.s'
    loop.write_text(LOOP)
    found = scan_mod.scan(str(loop))
    assert len(found) == 1 and 'reads the result of' in found[0]
    loop.write_text(LOOP.replace('\ts_cbranch_scc1 .LBB0_1', '\ts_nop 15\n\ts_nop 7\n\ts_cbranch_scc1 .LBB0_1'))
    assert scan_mod.scan(str(loop)) == []


def test_scanner_sees_both_hazards(tmp_path):
    bad, good = tmp_path / 'bad.s', tmp_path / 'good.s'
    bad.write_text(BROKEN)
    good.write_text(FIXED)
    found = scan_mod.scan(str(bad))
    assert len(found) == 2 and 'reads an operand that inline assembly' in found[0] and 'reads the result of' in found[1]
    assert scan_mod.scan(str(good)) == []


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc (cross-compiles gfx950 without a GPU)')
def test_no_hazard_hidden_behind_inline_assembly(tmp_path):
    srcs = [os.path.join(scan_mod.CSRC, f) for f in sorted(os.listdir(scan_mod.CSRC))
            if f.endswith('.hip') and '_mfma_' in open(os.path.join(scan_mod.CSRC, f)).read()]
    assert len(srcs) >= 5
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=6) as ex:
        listings = list(ex.map(lambda s: scan_mod.compile_listing(s, str(tmp_path)), srcs))
    findings = [f for p in listings for f in scan_mod.scan(p)]
    assert findings == [], '\n'.join(findings[:10])
