"""The compiled GEMM kernels may not touch the destination registers of an asm-issued transposed LDS read before an lgkmcnt(0) has covered
them, and no branch or join may lie between the read and that wait (tools/check_async_reads.py; the cause of round 2's plain-bf16 wrong
results). Runs on the CPU box: hipcc cross-compiles gemm.hip to gfx950 assembly."""
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'tools'))


@pytest.mark.skipif(not Path('/opt/rocm/bin/hipcc').exists() or shutil.which('c++filt') is None, reason='needs hipcc')
def test_no_instruction_reads_a_pending_transposed_fragment(tmp_path):
    import check_async_reads as chk
    out = tmp_path / 'gemm.s'
    chk.compile_to_asm(ROOT / 'exorl_amd' / 'csrc' / 'gemm.hip', out)
    funcs = chk.split_functions(out)
    checked = 0
    for name, lines in funcs.items():
        n_tr, findings = chk.check_kernel(name, lines)
        if n_tr:
            checked += 1
            assert not findings, (name, findings[:3])
    assert checked >= 8        # every k-image launch form of the product build (plain and split planes, 128- and 64-wide tiles, mixed)


def test_checker_flags_the_round2_pattern():
    """The pattern that was found, as a three-line listing: a copy of the read's destination in front of the wait."""
    import check_async_reads as chk
    lines = [(1, 'ds_read_b64_tr_b16 v[2:3], v40 offset:0', True), (2, 's_cbranch_scc0 .LBB0_1', False), (3, 'v_mov_b64_e32 v[36:37], v[2:3]', False),
             (4, 's_waitcnt lgkmcnt(0)', False), (5, 'v_mfma_f32_32x32x16_bf16 a[0:15], v[36:39], v[10:13], 0', False)]
    n, findings = chk.check_kernel('k', lines)
    assert n == 1 and [f[0] for f in findings] == [2, 3]
    ok = [(1, 'ds_read_b64_tr_b16 v[2:3], v40 offset:0', True), (2, 's_waitcnt lgkmcnt(0)', False), (3, 'v_mov_b64_e32 v[36:37], v[2:3]', False)]
    assert chk.check_kernel('k', ok) == (1, [])
