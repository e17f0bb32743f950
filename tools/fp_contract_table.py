"""gpurun_out/fpc/{wlbench.log, matrix_exact.md, matrix_fused.md} (tools/fp_contract_price.sh) -> markdown on stdout.
    python tools/fp_contract_table.py > profiles/r05_fp_contract_price.md"""
import collections
import json
import os
import sys

D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fpc")
cur, rows = None, collections.OrderedDict()
for ln in open(os.path.join(D, "wlbench.log")):
    if ln.startswith("==="):
        cur = ln.split()[1]
    elif ln.startswith("{"):
        d = json.loads(ln)
        key = (tuple(d["n"][:d["nd"]]), d["L"], d["filt"], d["dtype"], d["fw"], d["kernel"])
        rows.setdefault(key, {}).setdefault(cur, []).append(d["avg_us"])

print("# What bit-exactness costs on the round-4/5 kernels — one MI355X, `tools/fp_contract_price.sh`\n")
print("Same sources, same box, same run.  **exact** = the product build (`-ffp-contract=off`, every product and sum rounded separately,")
print("results bit-identical to the reference's CPU path); **fused** = `libwavelets_mi355x_fma.so` (`make FMA=1`: `-ffp-contract=fast`,")
print("`a*b+c` may contract to one FMA; agrees with the reference to SURVEY.md §8(c)'s tolerances, `tests/test_gpu_fused.py`).\n")
print("## Back-to-back calls (`tools/wlbench.bin mode=seq`, one event pair around 200–300 calls, two repetitions each)\n")
print("| case | L | T | dominant kernel | exact µs | fused µs | gain |")
print("|---|---|---|---|---|---|---|")
for (n, L, filt, dt, fw, kern), v in rows.items():
    e, f = v.get("exact", []), v.get("fused", [])
    if not e or not f:
        continue
    shape = " x ".join(str(k) for k in n)
    nm = ("dwt " if fw else "idwt ") + filt + " " + shape
    print(f"| {nm} | {L} | {dt} | {kern} | {' / '.join('%.1f' % t for t in e)} | {' / '.join('%.1f' % t for t in f)} | {100 * (min(e) / min(f) - 1):+.1f} % |")


def table(path):
    out, sect = collections.OrderedDict(), 0
    for ln in open(path):
        c = [s.strip() for s in ln.strip().strip("|").split("|")]
        if ln.startswith("| entry point"):
            sect = 1
        elif ln.startswith("| filter"):
            sect = 2
        elif ln.startswith("|") and not ln.startswith("|---"):
            if sect == 1:
                out[(c[0], c[1])] = (float(c[3]), float(c[4]), c[7])
            elif sect == 2:
                out[("filter " + c[0] + " (%s taps)" % c[1], "f32")] = (float(c[2]), float(c[3]), c[4] + " / " + c[5])
    return out


a, b = table(os.path.join(D, "matrix_exact.md")), table(os.path.join(D, "matrix_fused.md"))
print("\n## Every entry point, single calls (`tools/perf_matrix.py --arithmetic exact|fused`, median of one event pair per call)\n")
print("| entry point | T | exact fwd µs | fused fwd µs | gain | exact inv µs | fused inv µs | gain | kernels |")
print("|---|---|---|---|---|---|---|---|---|")
for k, (ef, ei, kn) in a.items():
    if k not in b:
        continue
    ff, fi, _ = b[k]
    print(f"| {k[0]} | {k[1]} | {ef:.1f} | {ff:.1f} | {100 * (ef / ff - 1):+.1f} % | {ei:.1f} | {fi:.1f} | {100 * (ei / fi - 1):+.1f} % | {kn} |")
