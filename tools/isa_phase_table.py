#!/usr/bin/env python3
"""Static instruction counts of k_fm_multi_poly12<6> per phase, from the kernel's own assembly.

Compiles csrc/kernels_poly.hip for gfx950 with -DRR_ISA_MARKS (RR_MARK("...") leaves a comment line at every phase boundary
and is otherwise a compiler barrier where the product build already has a sched_barrier / wave_fence), cuts the kernel at the
marks and counts instructions per class.  Dynamic counts per launch follow from the trip counts of BASELINE configs[3]
(2,400,000 samples, 1:6, 463 taps -> 423 tiles x 32 channels, 946 outputs per tile and channel): every region is straight-line
code in the steady state except the demodulation's pair loop (2 outputs per lane and iteration), whose body is counted
separately and multiplied by its 7 iterations (+ the single-output tail).  No GPU needed.  Output: a table on stdout
(profiles/r05_fm_multi_phase_table.txt is this script's output; SQ_INSTS_VALU of the same build is in
profiles/r05_fm_multi_stall_counters.txt)."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN2rr17k_fm_multi_poly12ILi6ENS_4VSrcINS_2cfEEEEEvT0_PfllPKS2_S7_iNS_8PolyArgsES7_PS2_NS_8PolyPartE"


def classify(op):
    if op.startswith("v_pk_"): return "valu_packed"
    if op in ("v_rcp_f32_e32", "v_rcp_f32_e64", "v_rsq_f32_e32", "v_sqrt_f32_e32"): return "valu_trans"
    if op.startswith("v_"): return "valu_scalar"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def count(lines):
    c = collections.Counter()
    for l in lines:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        c[classify(l.split()[0])] += 1
    return c


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "poly.s")
        extra = sys.argv[1:]
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize",
                        "--offload-device-only", "-S", "-DRR_ISA_MARKS", *extra, os.path.join(ROOT, "rustradio_amd", "csrc", "kernels_poly.hip"),
                        "-o", asm], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end + 1]
    meta = [l.strip() for l in text[end:end + 80] if re.search(r"; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)", l)]
    marks = [(i, l.split("RRMARK")[1].strip()) for i, l in enumerate(body) if "RRMARK" in l]
    names = [m[1] for m in marks]
    # (the marks appear in LAYOUT order — the loop latches are placed first; each region's blocks follow its mark)
    assert sorted(names) == sorted(["forward", "channels", "products", "inverse", "demod", "channel_end", "tile_end"]), names
    seg = {}
    for (i, n), (j, _) in zip(marks, marks[1:] + [(len(body), "end")]):
        seg.setdefault(n, []).extend(body[i:j])
    # the code between "demod" and "channel_end" holds both demodulator flavours; the exact-atan2 pair loop is the loop that
    # stores two dwords per iteration and uses v_fmaak (the polynomial); its single-output tail follows it
    dem = seg["demod"]
    loops = []
    labels = {l.split(":")[0]: k for k, l in enumerate(dem) if re.match(r"^\.LBB\d+_\d+:", l)}
    for k, l in enumerate(dem):
        m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            blk = dem[labels[m.group(1)]:k + 1]
            if any("v_fmaak_f32" in x or "v_fmamk_f32" in x for x in blk):
                loops.append(blk)
    nst = lambda b: sum("global_store_dword" in x for x in b)
    pair = min((b for b in loops if nst(b) == 2), key=len)
    single = min((b for b in loops if nst(b) == 1), key=len)
    # trip counts, configs[3] on one GPU
    tiles, chans, nv = 423, 32, 946
    ct = tiles * chans
    pair_iters, tail = nv // 128, (nv % 128 + 63) // 64       # 7 pair iterations; then 50 outputs: one single-output pass (50 of 64 lanes)
    tail_iters = tail
    rows = [
        ("forward (per tile run: 6 of 12 waves)", count(seg["forward"]), tiles * 6, "6 waves x 423 tile runs (+ the runs a tile is split into)"),
        ("products (per channel-tile)", count(seg["products"]), ct, "6 phases x 16 complex MACs per lane + response stream + parked spectra"),
        ("inverse + natural-order store", count(seg["inverse"]), ct, "1024-point inverse, 16 values per lane"),
        ("demodulation: pair loop body", count(pair), ct * pair_iters, f"{pair_iters} iterations per channel-tile, 2 outputs per lane each"),
        ("demodulation: single-output tail", count(single), ct * tail_iters, f"{tail_iters} pass per channel-tile"),
    ]
    classes = ["valu_packed", "valu_scalar", "valu_trans", "lds", "vmem", "salu", "s_nop", "s_waitcnt"]
    print("# k_fm_multi_poly12<6, VSrc<cf>>: static instruction counts per phase (tools/isa_phase_table.py" + (" " + " ".join(extra) if extra else "") + ")")
    for m in meta:
        print("#", m)
    print("# region".ljust(42) + "".join(c.rjust(12) for c in classes) + "   executions/launch")
    tot = collections.Counter()
    for name, c, trips, note in rows:
        print(name.ljust(42) + "".join(str(c[k]).rjust(12) for k in classes) + f"   {trips:>9}   # {note}")
        for k in classes:
            tot[k] += c[k] * trips
    valu = tot["valu_packed"] + tot["valu_scalar"] + tot["valu_trans"]
    print("# per launch (wave-instructions): " + ", ".join(f"{k} {tot[k]:.4g}" for k in classes))
    print(f"# predicted SQ_INSTS_VALU per launch {valu:.4g}; per channel-tile {valu / ct:.0f} VALU, of which packed {tot['valu_packed'] / ct:.0f}")
    shares = []
    for name, c, trips, _ in rows:
        v = (c["valu_packed"] + c["valu_scalar"] + c["valu_trans"]) * trips
        shares.append(f"{name.split(':')[-1].split('(')[0].strip()} {100 * v / valu:.1f} %")
    print("# VALU share per phase: " + "; ".join(shares))
    # executed flops: packed FMA 4, packed add / mul 2, scalar FMA 2, other scalar arithmetic 1 (moves, compares, selects 0)
    print("# (flops per VALU lane-instruction = executed flops / (SQ_INSTS_VALU x 64); bench.py's executed_flops_per_launch is the numerator)")


if __name__ == "__main__":
    main()
