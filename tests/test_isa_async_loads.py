"""Static guard for the asynchronous-load hazard class of the marching kernels (CPU only: the shipped .so is disassembled).

wl_dev.h issues `global_load_dword*` from inline asm and guards the data with hand-counted `s_waitcnt vmcnt(N)`.  Two things
have to hold in the GENERATED code and nothing else verifies them (round 2 and 3 each shipped a one-in-a-thousand wrong
transform from this class, found by luck):
  (a) no instruction reads, copies, spills or overwrites a destination VGPR between the load and a wait that covers it,
  (b) the wait counts younger LOADS only.
`tools/isa_check.py` checks both on every path of every kernel.  Round 4: the first run of it found two more latent cases
(k_fwd2d_lds_long: phi copies of ring registers BEFORE the drain-side wait; 2-tap pair kernels: the last prefetches still in
flight while the compiler reuses their registers) -- fixed, and pinned below through synthetic listings of the same shapes.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "wavelets.jl_amd", "libwavelets_mi355x.so")

# the toolchain the hand-placed kernels were validated with (ISA check green + GPU soak): a different compiler is not an
# error -- the check below IS the validation -- but the report says so
VALIDATED_HIPCC = "7.2.26015"


def _mod():
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)           # (a plain import: the checker's worker processes must be able to import it too)
    import isa_check
    return isa_check


def _listing(body):
    """objdump-style text of one function from (mnemonic, operands, encoding) triples; branch targets as instruction indices"""
    lines = ["0000000000001000 <k_test>:"]
    for k, (mn, ops, enc, *tgt) in enumerate(body):
        t = (" <k_test+0x%x>" % (4 * tgt[0])) if tgt else ""
        lines.append("\t%s %s // %012X: %08X%s" % (mn, ops, 0x1000 + 4 * k, enc, t))
    return "\n".join(lines) + "\n"


def _waitcnt(vm):
    return ("s_waitcnt", "vmcnt(%d)" % vm, 0xBF8C0F70 | (vm & 0xF) | ((vm >> 4) << 14))


def _check(body, strict=True):
    m = _mod()
    funcs = m.parse_functions(_listing(body))
    return m.check_function(funcs["k_test"], strict=strict)


LOAD_A = ("global_load_dwordx4", "v[10:13], v[2:3], off", 0xDC5C8000)
LOAD_B = ("global_load_dwordx4", "v[14:17], v[2:3], off", 0xDC5C8000)
STORE = ("global_store_dwordx4", "v[4:5], v[20:23], off", 0xDC7C8000)
USE_A = ("v_pk_mul_f32", "v[30:31], s[4:5], v[10:11]", 0xD3B1401E)
END = ("s_endpgm", "", 0xBF810000)


def test_checker_accepts_a_covered_load():
    assert _check([LOAD_A, LOAD_B, _waitcnt(1), USE_A, END]) == []
    assert _check([LOAD_A, _waitcnt(0), USE_A, END]) == []
    # never used: fine
    assert _check([LOAD_A, END]) == []


def test_checker_flags_read_before_wait_and_copies():
    assert _check([LOAD_A, USE_A, _waitcnt(0), END])
    assert _check([LOAD_A, ("v_mov_b64_e32", "v[40:41], v[12:13]", 0x7E000000), _waitcnt(0), END])        # a phi copy
    assert _check([LOAD_A, ("scratch_store_dwordx4", "off, v[10:13], off", 0xDC000000), _waitcnt(0), END])  # a spill
    assert _check([LOAD_A, ("v_add_u32_e32", "v11, s2, v1", 0x68000000), _waitcnt(0), END])                # register reused


def test_checker_flags_a_reloaded_sgpr_base_without_wait_states():
    """Round 6 (wl_fwd3d.hip: scalar-base loads and stores from inline asm): 5 wait states between a VALU write of an SGPR and a
    vector-memory instruction that reads it as its base."""
    m = _mod()
    rl0 = ("v_readlane_b32", "s2, v255, 3", 0xD2890002)
    rl1 = ("v_readlane_b32", "s3, v255, 4", 0xD2890003)
    ld = ("global_load_dwordx4", "v[10:13], v1, s[2:3]", 0xDC5C8000)
    nop4 = ("s_nop", "4", 0xBF800004)
    salu = ("s_add_u32", "s2, s4, s6", 0x80020604)
    f = lambda body: m.check_sgpr_base_hazard(m.parse_functions(_listing(body))["k_test"])
    assert f([rl0, rl1, ld, _waitcnt(0), END])
    assert f([rl0, rl1, nop4, ld, _waitcnt(0), END]) == []
    assert f([rl0, rl1, salu, ("s_addc_u32", "s3, s5, s7", 0x82030705), ld, _waitcnt(0), END]) == []     # the base is SALU-made
    assert f([rl0, rl1, ("global_load_dwordx4", "v[10:13], v[2:3], off", 0xDC5C8000), _waitcnt(0), END]) == []


def test_checker_does_not_count_stores_in_strict_mode():
    body = [LOAD_A, STORE, _waitcnt(1), USE_A, END]
    assert _check(body, strict=True)            # the store may be acknowledged first: vmcnt(1) proves nothing about the load
    assert _check(body, strict=False) == []     # LLVM's in-order model for compiler-placed waits
    assert _check([LOAD_A, STORE, LOAD_B, _waitcnt(1), USE_A, END], strict=True) == []


def test_checker_follows_branches_and_flags():
    # loop: the load is consumed one trip later, behind a wait that counts the younger load of that trip
    body = [LOAD_A, LOAD_B, _waitcnt(2), ("s_cbranch_scc1", "65533", 0xBF85FFFD, 0), USE_A, END]
    assert _check(body)                          # falls out of the loop with vmcnt(2): not covered
    # the gload16_if / wait_vm_sel pair: a skipped load must not be paired with the loading side's wait
    sel = [
        ("s_cmp_eq_u32", "s8, 0", 0xBF068008), ("s_cbranch_scc1", "1", 0xBF850001, 3), LOAD_B,               # 0 1 2
        ("s_cmp_eq_u32", "s8, 0", 0xBF068008), ("s_cbranch_scc1", "2", 0xBF850002, 7),                       # 3 4
        _waitcnt(1), ("s_branch", "1", 0xBF820001, 8), _waitcnt(0),                                          # 5 6 7
        USE_A, END,                                                                                          # 8 9
    ]
    assert _check([LOAD_A] + [(mn, ops, enc, *[t + 1 for t in tg]) for (mn, ops, enc, *tg) in sel]) == []
    # ... while the round-3 shape (copy on the drain side BEFORE its wait) is flagged
    bad = [
        LOAD_A, ("s_cmp_eq_u32", "s8, 0", 0xBF068008), ("s_cbranch_scc1", "2", 0xBF850002, 5),
        _waitcnt(0), ("s_branch", "2", 0xBF820002, 7),
        ("v_mov_b64_e32", "v[40:41], v[10:11]", 0x7E000000), _waitcnt(0),
        USE_A, END,
    ]
    assert _check(bad)


def test_library_has_no_async_load_hazard():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    m = _mod()
    stats, report = m.check_library(LIB)
    assert stats["hand_placed_kernels"] >= 80, stats          # the five families are in the object (names did not drift)
    assert stats["loads"] > 10000, stats
    msg = "\n".join("%s\n    %s" % (k, "\n    ".join(v)) for k, v in report.items())
    try:
        ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.split("\n")[0]
    except OSError:
        ver = "unknown"
    assert not report, "asynchronous-load hazards in the shipped code (%s; validated with %s):\n%s" % (ver, VALIDATED_HIPCC, msg)


def test_instruction_mix_tool_reports_the_headline_kernels():
    """tools/isa_stats.py (round 4): the static instruction mix of the shipped kernels.  Pins two facts the design relies on:
    the 12 ... 20-tap inverse level kernel moves NO spilled SGPRs through lanes (its detail taps are the negated scaling taps), and
    no dominant kernel uses scratch except the inverse pair's documented 68 bytes."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_stats.py")], capture_output=True, text=True, check=True).stdout
    rows = {}
    for line in out.splitlines()[2:]:
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        rows[cells[0].strip("`")] = [int(v) for v in cells[1:]]
    assert any(k.startswith("k_fwd2d_pair<8, 2, 1, 0, 0>") for k in rows) and any(k.startswith("k_fwd2d_lds<8, 1, 0>") for k in rows)
    for k, v in rows.items():
        # columns: instructions, VALU, packed, DPP, lane moves, SALU, LDS, vmem, barriers, waitcnt, VGPRs, SGPR spills, scratch
        if k.startswith("k_inv2d_lds_long<float, 16"):
            assert v[4] == 0 and v[11] == 0, (k, v)
        if not k.startswith("k_inv2d_pair"):
            assert v[12] == 0, (k, v)
