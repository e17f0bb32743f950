"""Static instruction mix of the shipped kernels (CPU only: disassembles the .so like tools/isa_check.py).

    python tools/isa_stats.py [--filter REGEX] [--so PATH]            -> markdown table on stdout

Per kernel: instructions, VALU (of which packed Float32, DPP, v_readlane / v_writelane = SGPR-spill traffic), SALU, LDS, vector
memory, barriers; registers / scratch / SGPR spills from the code-object metadata.  Not a performance model -- a way to see
what a kernel spends its issue slots on (round 4: the 16-tap inverse level kernel reloaded 78 spilled SGPRs per step through
v_readlane_b32; taking the detail taps from the scaling taps removed them: 164 -> 150 us)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_check as ic                                                     # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    if not names:
        return []
    return subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "wavelets.jl_amd", "libwavelets_mi355x.so"))
    ap.add_argument("--filter", default=r"k_fwd2d_pair<8, 2, 1, 0, 0>|k_fwd3d_one<(float, 4|double, 2), 8, (2|4)>|k_inv3d_one<float, 4, 8, 2>|k_level3_lds<float, 8, 4, [01]>|k_fwd2d_lds<8, 1, 0>|k_inv2d_pair<8|k_inv2d_lds_long<float, (10|16), 1, [23], 0, [12]>|"
                                       r"k_fwd2d_lds_long<16, 20, 1>|k_fwd1d_multi<float, 8, 1>|k_lift1d_fwd3<float, 0, 1>|k_lift2d_fwd<float, 0, 8>|"
                                       r"k_fwd2d_tileB<8>|k_fwd2d_pair64<8, 2, 1>|k_inv2d_stream<float, 10, 2>")
    args = ap.parse_args()
    pat = re.compile(args.filter)
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for co in ic.extract_code_objects(args.so, d):
            notes = subprocess.run([os.path.join(ic.LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            meta = {}
            for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
                meta[m.group(1)] = (int(m.group(4)), int(m.group(2)), int(m.group(3)))
            txt = ic.disassemble(co)
            heads = list(re.finditer(r"^[0-9a-f]+ <(\w+)>:\n", txt, re.M))
            names = [h.group(1) for h in heads]
            dem = demangle(names)
            for k, h in enumerate(heads):
                dn = re.sub(r"\(.*$", "", dem[k]).replace("void wl::", "").replace("(anonymous namespace)::", "")
                if not pat.search(dn):
                    continue
                body = txt[h.end():heads[k + 1].start() if k + 1 < len(heads) else len(txt)]
                ops = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith("//")]
                c = collections.Counter(ops)
                valu = sum(v for o, v in c.items() if o.startswith("v_"))
                pk = sum(v for o, v in c.items() if o.startswith("v_pk_") and o.endswith("f32"))
                dpp = sum(v for o, v in c.items() if o.endswith("_dpp"))
                lane = c["v_readlane_b32"] + c["v_writelane_b32"]
                salu = sum(v for o, v in c.items() if o.startswith("s_") and o not in ("s_waitcnt", "s_barrier", "s_nop"))
                lds = sum(v for o, v in c.items() if o.startswith("ds_"))
                vmem = sum(v for o, v in c.items() if o.startswith(("global_", "buffer_", "scratch_", "flat_")))
                vg, scr, ssp = meta.get(h.group(1), (0, 0, 0))
                rows.append((dn, len(ops), valu, pk, dpp, lane, salu, lds, vmem, c["s_barrier"], c["s_waitcnt"], vg, ssp, scr))
    print("| kernel | instructions | VALU | packed f32 | DPP | read/writelane | SALU | LDS | vector memory | barriers | waitcnt | VGPRs | SGPR spills | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in sorted(rows):
        print("| `%s` | %s |" % (r[0], " | ".join(str(v) for v in r[1:])))


if __name__ == "__main__":
    main()
