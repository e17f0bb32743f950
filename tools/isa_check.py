#!/usr/bin/env python3
"""Static check of the shipped gfx950 code for the asynchronous-load hazard class (no GPU needed).

The marching kernels (wl_fwd2d.hip, wl_pair2d.hip, wl_fwd2d64.hip, wl_pair2d64.hip, wl_fwd2d_long.hip) issue their column loads
from inline asm (`gload16 / gload4 / gload16_if`, wl_dev.h) and guard the data with hand-counted `s_waitcnt vmcnt(N)`
(`wait_vm<N>`).  The compiler does not know that those asm statements are loads, so nothing but this check stops it from
copying, spilling or reading a destination register while the load is still in flight (round 3 found one such case by luck: a
phi copy around an `if`), and nothing but this check verifies the counting rule: vector-memory LOADS return in order among
themselves, STORES may be acknowledged before an older load returns, so a load is known to have landed only at a
`s_waitcnt vmcnt(n)` with n <= the number of younger LOADS issued since (stores must not be counted towards n).

For every vector-memory load of every kernel in the library the checker walks all control-flow paths from the load (branch
targets are taken from the disassembly) until such a wait, and reports
  * any instruction on the way that reads or writes one of the load's destination VGPRs,
  * a path that reaches `s_endpgm` is fine (the data was never used), a path into an indirect jump is reported.
The other kernels (compiler-placed loads and waits) are checked against LLVM's own model for gfx9-family targets -- loads and
stores share one in-order vmcnt, so every younger vector-memory operation counts -- which catches a mis-assembled or
hand-edited object, not a compiler bug.

Usage: isa_check.py [path/to/libwavelets_mi355x.so] [--filter substring] [--keep DIR]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("WL_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_re_func = re.compile(r"^([0-9a-f]{8,16}) <([^>]+)>:\s*$")
_re_inst = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*([0-9A-Fa-f]{8})(?:\s+([0-9A-Fa-f]{8}))?(?:.*<([^>+]+)\+0x([0-9a-fA-F]+)>)?")
_re_vreg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")

LOAD_PREFIXES = ("global_load_", "buffer_load_", "flat_load_", "scratch_load_", "tbuffer_load_")
VMEM_PREFIXES = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_")
# kernels whose loads are issued from inline asm and guarded by hand-counted waits: the strict rule applies
HAND_PLACED = ("k_fwd2d_lds", "k_fwd2d_pair", "k_fwd3d_one", "k_inv3d_one")


def extract_code_objects(so_path, workdir):
    """-> list of paths of the gfx950 code objects embedded in the shared library (one per translation unit)"""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path])
    data = open(fat, "rb").read()
    offs = []
    i = data.find(MAGIC)
    while i >= 0:
        offs.append(i)
        i = data.find(MAGIC, i + 1)
    out = []
    for k, o in enumerate(offs):
        e = offs[k + 1] if k + 1 < len(offs) else len(data)
        b = os.path.join(workdir, "bundle%d.bin" % k)
        open(b, "wb").write(data[o:e])
        co = os.path.join(workdir, "co%d.elf" % k)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                               "--input=" + b, "--output=" + co])
        if os.path.getsize(co) > 0:
            out.append(co)
    return out


def disassemble(co_path):
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co_path], check=True, capture_output=True, text=True).stdout


class Inst:
    __slots__ = ("addr", "mn", "ops", "enc", "target", "regs", "dst", "srcs")

    def __init__(self, addr, mn, ops, enc, target):
        self.addr, self.mn, self.ops, self.enc, self.target = addr, mn, ops, enc, target
        self.regs = None   # set of VGPR numbers mentioned anywhere in the operand list
        self.dst = None    # destination VGPRs of a load
        self.srcs = None   # ... and the VGPRs of its address operands


def _vregs(text):
    s = set()
    for m in _re_vreg.finditer(text):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


def parse_functions(text, name_filter=None):
    """-> {symbol: [Inst]} for the kernels (and device functions) of one disassembled code object"""
    funcs = {}
    cur = None
    base = {}
    for line in text.splitlines():
        m = _re_func.match(line)
        if m:
            name = m.group(2)
            base[name] = int(m.group(1), 16)
            cur = [] if (name_filter is None or name_filter in name) else None
            if cur is not None:
                funcs[name] = cur
            continue
        if cur is None:
            continue
        m = _re_inst.match(line)
        if not m:
            continue
        mn, ops, addr, enc = m.group(1), m.group(2), int(m.group(3), 16), int(m.group(4), 16)
        target = None
        if m.group(6) is not None and (mn.startswith("s_cbranch") or mn == "s_branch"):
            target = (m.group(6), int(m.group(7), 16))
        ins = Inst(addr, mn, ops, enc, target)
        ins.regs = _vregs(ops)
        if mn.startswith(LOAD_PREFIXES):
            ins.dst = _vregs(ops.split(",")[0])
            ins.srcs = _vregs(",".join(ops.split(",")[1:]))
        cur.append(ins)
    # resolve branch targets to instruction indices
    for name, insts in funcs.items():
        index = {ins.addr: i for i, ins in enumerate(insts)}
        for ins in insts:
            if ins.target is not None:
                sym, off = ins.target
                ins.target = index.get(base.get(sym, -1) + off) if sym == name else None
                if ins.target is None:
                    ins.target = -1      # leaves the function: reported if a load is in flight
    return funcs


def _vmcnt(ins):
    """vmcnt field of an s_waitcnt (gfx9 encoding: simm16[3:0] | simm16[15:14] << 4); 63 = no wait"""
    simm = ins.enc & 0xFFFF
    return (simm & 0xF) | (((simm >> 14) & 3) << 4)


_re_sreg = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def _sdst(ins):
    """SGPRs written by an instruction (first operand of SALU / VALU-to-SGPR forms; compares write only scc / vcc)"""
    mn = ins.mn
    if mn.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_nop", "s_endpgm", "s_bitcmp")):
        return ()
    if not (mn.startswith("s_") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane") or mn.startswith("v_cmp")
            or "s[" in ins.ops or re.search(r"\bs\d", ins.ops)):
        return ()
    first = ins.ops.split(",")[0]
    out = []
    for m in _re_sreg.finditer(first):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    # VOP3 forms with an SGPR carry / mask as SECOND operand (v_mad_u64_u32 v[..], s[6:7], ...; v_add_co_u32 v, s[..], ...)
    parts = ins.ops.split(",")
    if mn.startswith("v_") and len(parts) > 1:
        for m in _re_sreg.finditer(parts[1]):
            if m.group(1) is not None:
                out.append(int(m.group(1)))
            else:
                out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


_re_pair_const = re.compile(r"^s\[(\d+):(\d+)\], (-1|0)$")
_re_vcc_mask = re.compile(r"^vcc, exec, s\[(\d+):(\d+)\]$")
_re_flag_cmp = re.compile(r"^s(\d+), 0$")


def _kill(facts, ins):
    """drop the facts an instruction invalidates"""
    if not facts:
        return facts
    w = _sdst(ins)
    vcc_written = "vcc" in ins.ops
    if not w and not vcc_written:
        return facts
    out = []
    for k, v in facts:
        if k[0] == "z" and k[1] in w:
            continue
        if k[0] == "p" and (k[1] in w or k[1] + 1 in w):
            continue
        if k[0] == "vcc" and vcc_written:
            continue
        out.append((k, v))
    return tuple(out)


def _with(facts, key, val):
    d = dict(facts)
    d[key] = val
    return tuple(sorted(d.items()))


def _facts_before(insts, li, joins):
    """What is known when control reaches insts[li]: scan the straight-line code above it for exit-flag constants and flag
    tests that are still valid.  joins: {target index: [indices of the branches that jump there]}.  The scan passes a join
    only if every branch into it is a forward skip from inside the scanned code; a fact found at index i is used only if all
    those skips start after i (a skip that starts before i bypasses the instruction that established the fact)."""
    facts = ()
    killed = set()
    limit = 1 << 30                      # lowest index of a branch into one of the joins passed so far
    i = li - 1
    while i >= 0:
        ins = insts[i]
        mn = ins.mn
        if mn in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            break
        if i < limit:
            if mn == "s_mov_b64":
                m = _re_pair_const.match(ins.ops)
                if m and int(m.group(2)) == int(m.group(1)) + 1:
                    lo = int(m.group(1))
                    if lo not in killed and lo + 1 not in killed and ("p", lo) not in dict(facts):
                        facts = _with(facts, ("p", lo), m.group(3) == "-1")
            if mn in ("s_cbranch_scc1", "s_cbranch_scc0") and i > 0 and insts[i - 1].mn == "s_cmp_eq_u32" and \
                    isinstance(ins.target, int) and ins.target > li:
                m = _re_flag_cmp.match(insts[i - 1].ops)
                if m and int(m.group(1)) not in killed and ("z", int(m.group(1))) not in dict(facts):
                    # we are on the fall-through side: scc1 not taken <=> the register is not zero
                    facts = _with(facts, ("z", int(m.group(1))), mn == "s_cbranch_scc0")
        killed.update(_sdst(ins))
        if i in joins:
            srcs = joins[i]
            if any(b >= i for b in srcs):
                break                    # a loop header: whatever lies above holds for the first trip only
            limit = min(limit, min(srcs))
        i -= 1
    return facts


def _joins(insts):
    j = {}
    for k, ins in enumerate(insts):
        if isinstance(ins.target, int) and ins.target >= 0:
            j.setdefault(ins.target, []).append(k)
    return j


def walk_from(insts, li, strict, on_violation, trace=None, joins=None):
    """All control-flow paths from the load insts[li] until a covering wait.  Paths are sensitive to the two kinds of
    wave-uniform conditions the hand-placed code and the compiler's structurised loops produce:
      * `s_cmp_eq_u32 sN, 0` + `s_cbranch_scc0/1` -- the flags of gload16_if / wait_vm_sel: whether sN is zero is remembered
        until sN is rewritten, so a skipped load is never paired with the wait of the loading side;
      * `s_mov_b64 s[a:b], -1 | 0` ... `s_and_b64 / s_andn2_b64 vcc, exec, s[a:b]` + `s_cbranch_vccz/vccnz` -- the exit flags
        of a structurised multi-exit loop ("leave" is a jump to the loop header with the flag set)."""
    count_prefixes = LOAD_PREFIXES if strict else VMEM_PREFIXES
    n = len(insts)
    ld = insts[li]
    dst = ld.dst
    best = {}
    stack = [(li + 1, 0, _facts_before(insts, li, joins or {}), None)]
    while stack:
        i, younger, facts, parent = stack.pop()
        while True:
            if i >= n:
                break
            key = (i, facts)
            if best.get(key, 1 << 30) <= younger:
                break
            best[key] = younger
            if trace is not None:
                trace[key] = parent
                parent = key
            ins = insts[i]
            mn = ins.mn
            if mn == "s_waitcnt":
                if _vmcnt(ins) <= younger:
                    break                       # the load has landed on this path
                i += 1
                continue
            if mn == "s_endpgm":
                break
            # (a younger LOAD into the same registers is harmless: loads return in order, the younger one lands last)
            touched = ins.srcs if ins.dst is not None else ins.regs
            if touched and (touched & dst):
                if on_violation(ins, touched & dst, younger, key):
                    return
                break
            if mn.startswith(count_prefixes):
                younger += 1
            if mn == "s_branch":
                if ins.target is None or ins.target < 0:
                    on_violation(ins, None, younger, key)
                    break
                i = ins.target
                continue
            if mn.startswith("s_cbranch"):
                if ins.target is None or ins.target < 0:
                    on_violation(ins, None, younger, key)
                    break
                d = dict(facts)
                decided = None                  # True: taken, False: not taken
                if mn in ("s_cbranch_scc1", "s_cbranch_scc0") and i > 0 and insts[i - 1].mn == "s_cmp_eq_u32":
                    m = _re_flag_cmp.match(insts[i - 1].ops)
                    if m:
                        reg = int(m.group(1))
                        known = d.get(("z", reg))
                        taken_means_zero = (mn == "s_cbranch_scc1")
                        for zero in ((known,) if known is not None else (True, False)):
                            nxt = ins.target if (zero == taken_means_zero) else i + 1
                            stack.append((nxt, younger, _with(facts, ("z", reg), zero), parent))
                        break
                if mn in ("s_cbranch_vccnz", "s_cbranch_vccz") and ("vcc",) in d:
                    decided = (d[("vcc",)] == (mn == "s_cbranch_vccnz"))
                if decided is None:
                    stack.append((ins.target, younger, facts, parent))
                    i += 1
                elif decided:
                    i = ins.target
                else:
                    i += 1
                continue
            if mn in ("s_setpc_b64", "s_swappc_b64"):
                on_violation(ins, None, younger, key)
                break
            # ---- facts ----
            if mn == "s_mov_b64":
                m = _re_pair_const.match(ins.ops)
                facts = _kill(facts, ins)
                if m and int(m.group(2)) == int(m.group(1)) + 1:
                    facts = _with(facts, ("p", int(m.group(1))), m.group(3) == "-1")
            elif mn in ("s_and_b64", "s_andn2_b64") and ins.ops.startswith("vcc, exec, s["):
                m = _re_vcc_mask.match(ins.ops)
                known = dict(facts).get(("p", int(m.group(1)))) if m else None
                facts = _kill(facts, ins)
                if known is not None:           # exec is non-zero in a running wave
                    facts = _with(facts, ("vcc",), known if mn == "s_and_b64" else (not known))
            else:
                facts = _kill(facts, ins)
            i += 1


def check_function(insts, strict=True, max_report=5):
    """-> list of violation strings.  strict: only younger LOADS count towards a covering vmcnt (the rule of the hand-placed
    families); otherwise every younger vector-memory operation counts (LLVM's own model for gfx9-family targets, where loads
    and stores share one in-order counter)."""
    bad = []
    joins = _joins(insts)
    for li, ld in enumerate(insts):
        if ld.dst is None or not ld.dst:
            continue

        def on_violation(ins, regs, younger, key, ld=ld):
            if regs is None:
                bad.append("load at 0x%x (%s %s) in flight across a jump out of the function at 0x%x" % (ld.addr, ld.mn, ld.ops, ins.addr))
            else:
                bad.append("%s at 0x%x touches v%s of the load at 0x%x (%s %s) before a covering s_waitcnt (younger %s: %d)"
                           % (ins.mn + " " + ins.ops, ins.addr, sorted(regs), ld.addr, ld.mn, ld.ops,
                              "loads" if strict else "memory operations", younger))
            return len(bad) >= max_report

        walk_from(insts, li, strict, on_violation, joins=joins)
        if len(bad) >= max_report:
            break
    return bad


_re_spair_last = re.compile(r"s\[(\d+):(\d+)\]")


def check_sgpr_base_hazard(insts, max_report=5):
    """Vector-memory instructions that take their base from an SGPR pair (`global_load_dwordx4 v[..], v1, s[2:3]`: the scalar-base
    forms of wl_fwd3d.hip, issued from inline asm where hipcc's hazard recogniser does not look): gfx9-family parts need 5 wait
    states between a VALU write of an SGPR (v_readlane_b32 / v_readfirstlane_b32 -- the reload of a spilled SGPR) and a
    vector-memory instruction reading it.  -> list of violation strings.  (Straight-line look-back; s_nop N counts N + 1.)"""
    bad = []
    for k, ins in enumerate(insts):
        if not ins.mn.startswith(("global_load", "global_store", "global_atomic")):
            continue
        parts = [x.strip() for x in ins.ops.split(",")]
        base = None
        for x in parts[1:]:
            m = _re_spair_last.match(x.split()[0]) if x else None
            if m:
                base = set(range(int(m.group(1)), int(m.group(2)) + 1))
        if not base:
            continue
        states, j, pending = 0, k - 1, set(base)
        while j >= 0 and states < 5 and pending:
            p = insts[j]
            if p.mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            w = pending & set(_sdst(p))
            pending -= w                                  # (the youngest writer of each register is the one that matters)
            if p.mn.startswith(("v_readlane", "v_readfirstlane")) and w:
                bad.append("%s %s at 0x%x reads an SGPR base written by %s %s %d wait state(s) earlier (5 required)"
                           % (ins.mn, ins.ops, ins.addr, p.mn, p.ops, states))
                break
            if p.mn == "s_nop":
                states += (p.enc & 0xF) + 1
            else:
                states += 1
            j -= 1
        if len(bad) >= max_report:
            break
    return bad


def trace_path(insts, load_addr, strict=True):
    """debugging aid: the branch decisions of the first violating path from the load at load_addr"""
    li = [k for k, i in enumerate(insts) if i.addr == load_addr][0]
    tr = {}
    hit = []

    def on_violation(ins, regs, younger, key):
        hit.append(key)
        return True

    walk_from(insts, li, strict, on_violation, trace=tr, joins=_joins(insts))
    if not hit:
        return []
    path, k = [], hit[0]
    while k is not None:
        path.append(k)
        k = tr[k]
    path.reverse()
    out, prev = [], None
    for k in path:
        ins = insts[k[0]]
        jumped = prev is not None and k[0] != prev + 1
        if jumped or ins.mn.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_mov_b64", "s_and_b64", "s_andn2_b64") + LOAD_PREFIXES) or k == hit[0]:
            out.append("%s0x%x %s %s   %s" % ("-> " if jumped else "   ", ins.addr, ins.mn, ins.ops, dict(k[1]) if k[1] else ""))
        prev = k[0]
    return out


def _check_code_object(args):
    co, name_filter = args
    stats = {"functions": 0, "loads": 0, "instructions": 0, "hand_placed_kernels": 0}
    report = {}
    for name, insts in parse_functions(disassemble(co), name_filter).items():
        stats["functions"] += 1
        stats["instructions"] += len(insts)
        stats["loads"] += sum(1 for i in insts if i.dst)
        hand = any(f in name for f in HAND_PLACED)
        if hand:
            stats["hand_placed_kernels"] += 1
        v = check_function(insts, strict=hand) + check_sgpr_base_hazard(insts)
        if v:
            report[name] = v
    return stats, report


def check_library(so_path, name_filter=None, keep=None, jobs=None):
    """-> (stats dict, {symbol: [violations]}); so_path: the shared library or one of its object files"""
    tmp = keep or tempfile.mkdtemp(prefix="wl_isa_")
    os.makedirs(tmp, exist_ok=True)
    stats = {"code_objects": 0, "functions": 0, "loads": 0, "instructions": 0, "hand_placed_kernels": 0}
    report = {}
    cos = extract_code_objects(so_path, tmp)
    stats["code_objects"] = len(cos)
    jobs = jobs or min(len(cos), os.cpu_count() or 1, 16)
    if jobs > 1:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=jobs) as ex:
            results = list(ex.map(_check_code_object, [(co, name_filter) for co in cos]))
    else:
        results = [_check_code_object((co, name_filter)) for co in cos]
    for st, rep in results:
        for k, v in st.items():
            stats[k] += v
        report.update(rep)
    if keep is None:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return stats, report


def main(argv):
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "..", "wavelets.jl_amd", "libwavelets_mi355x.so")
    flt, keep, trace = None, None, None
    args = list(argv[1:])
    while args:
        a = args.pop(0)
        if a == "--filter":
            flt = args.pop(0)
        elif a == "--keep":
            keep = args.pop(0)
        elif a == "--trace":                    # --trace 0xADDR (with --filter naming one kernel): print the offending path
            trace = int(args.pop(0), 16)
        else:
            so = a
    if trace is not None:
        tmp = tempfile.mkdtemp(prefix="wl_isa_")
        for co in extract_code_objects(so, tmp):
            for name, insts in parse_functions(disassemble(co), flt).items():
                if any(i.addr == trace for i in insts):
                    print(name)
                    print("\n".join(trace_path(insts, trace, any(f in name for f in HAND_PLACED))))
        return 0
    stats, report = check_library(so, flt, keep)
    print(stats)
    for name, v in report.items():
        print(name)
        for s in v:
            print("   ", s)
    return 1 if report else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
