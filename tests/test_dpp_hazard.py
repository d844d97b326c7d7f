"""The hand-written `v_fmac_f32_dpp` statements of the solve rely on the two wait states between a vector write of a
register and a DPP read of it, which the compiler does not provide inside an asm statement: the generated code of
the shipped kernels must keep that distance (tools/check_dpp_hazard.py)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(ROOT, "tools", "check_dpp_hazard.py"))
chk = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(chk)

DPP = "\tv_fmac_f32_dpp v5, v5, v9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"


def test_checker_flags_a_write_right_before_the_dpp_read():
    assert chk.check("_Zk:\n\tv_mov_b32_e32 v5, v7\n" + DPP)
    assert chk.check("_Zk:\n\tv_accvgpr_read_b32 v5, a3\n\tv_add_f32_e32 v1, v2, v3\n" + DPP)
    assert chk.check("_Zk:\n\tv_pk_add_f32 v[4:5], v[0:1], v[2:3]\n" + DPP)


def test_checker_accepts_two_wait_states():
    assert not chk.check("_Zk:\n\tv_mov_b32_e32 v5, v7\n\ts_nop 1\n" + DPP)
    assert not chk.check("_Zk:\n\tv_mov_b32_e32 v5, v7\n\tv_add_f32_e32 v1, v2, v3\n\tv_add_f32_e32 v2, v2, v3\n" + DPP)
    assert not chk.check("_Zk:\n\tv_mov_b32_e32 v6, v7\n" + DPP)                 # another register
    assert not chk.check("_Zk:\n\tv_mov_b32_e32 v9, v7\n" + DPP)                 # the plain (non-DPP) source


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_shipped_solve_kernels_keep_the_distance():
    text = chk.assembly_of(os.path.join(chk.CSRC, "als_kernels.hip"))
    assert text.count("v_fmac_f32_dpp") > 1000            # the statements are there
    assert chk.check(text) == []


def test_checker_flags_a_matrix_core_result_read_too_early():
    mfma = "\tv_mfma_f32_16x16x4_f32 v[4:7], v1, v2, v[4:7]\n"
    assert chk.check("_Zk:\n" + mfma + "\tv_add_f32_e32 v1, v2, v3\n" * 5 + DPP)
    assert not chk.check("_Zk:\n" + mfma + "\tv_add_f32_e32 v1, v2, v3\n" * 12 + DPP)      # 8 passes + 4
    assert not chk.check("_Zk:\n" + mfma + "\ts_nop 7\n\ts_nop 3\n" + DPP)
    big = "\tv_mfma_f32_32x32x2_f32 v[0:15], v20, v21, v[0:15]\n"                               # 16 passes
    assert chk.check("_Zk:\n" + big + "\tv_add_f32_e32 v30, v31, v32\n" * 12 + DPP)
    assert not chk.check("_Zk:\n" + big + "\tv_add_f32_e32 v30, v31, v32\n" * 20 + DPP)


def test_checker_reads_two_operand_dpp_instructions():
    """`v_mov_b32_dpp v1, v2 quad_perm:[1,0,3,2] ...`: the register read through DPP is followed by modifiers, not
    by a comma; the compiler's own DPP moves are of this form and must be checked like the others."""
    mov = "\tv_mov_b32_dpp v1, v2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
    assert chk.check("_Zk:\n\tv_mov_b32_e32 v2, v7\n" + mov)
    assert not chk.check("_Zk:\n\tv_mov_b32_e32 v2, v7\n\ts_nop 1\n" + mov)
    assert not chk.check("_Zk:\n\tv_mov_b32_e32 v3, v7\n" + mov)


def test_checker_is_conservative_at_labels():
    """A DPP read right behind a label may follow a write on the other path into the label."""
    assert chk.check("_Zk:\n\tv_mov_b32_e32 v1, v7\n\ts_nop 1\n.LBB0_3:\n" + DPP)
    assert not chk.check("_Zk:\n.LBB0_3:\n\tv_add_f32_e32 v1, v2, v3\n\tv_add_f32_e32 v2, v2, v3\n" + DPP)


def test_flags_come_from_the_makefile():
    flags = chk.makefile_flags()
    assert "-O3" in flags and "-fno-slp-vectorize" in flags


def test_m0_lint_flags_compiler_uses():
    """m0 may be written only by the `s_mov_b32 m0, <sgpr>` of the LDS-DMA assembly statements."""
    lint = chk
    ok = ";;#ASMSTART\n\ts_mov_b32 m0, s12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[2:3], off\n;;#ASMEND\n"
    assert lint.check_m0(ok) == []
    assert len(lint.check_m0("\ts_mov_b32 m0, s12\n")) == 1                      # outside an assembly statement
    assert len(lint.check_m0(";;#ASMSTART\n\ts_mov_b32 m0, v3\n;;#ASMEND\n")) == 1     # not from a scalar register
    assert len(lint.check_m0("\tv_movrels_b32 v1, v2 ; uses m0\n\ts_movk_i32 m0, 0x3800\n")) == 1
