#!/usr/bin/env python3
"""Generates distributedfft_amd/csrc/kernels_mixed.inc: the PassCfg of every natively supported line length that is
NOT a power of two (N = 2^a 3^b 5^c 7^d <= 2048), for both precisions.

The reference accepts any length through cufftMakePlanMany64 (src/pencil/mpicufft_pencil_opt1.cpp:165-197); here a length
with a configuration runs the same Stockham axis-pass kernel as the powers of two (mixed-radix butterflies, fft_pass.hip.h:
Dif / dft_prime), every other length <= 1024 runs the Bluestein kernel (~2.6x the time of a native pass).

A configuration is (E points per thread, radices r1..r4 with product N, G tiles per workgroup, LDS planes).  Constraints of
the kernel: every radix divides E and is <= 32; threads = (N/E) * TL * G <= 1024; LDS <= 160 KiB; the E points plus an
overhead estimate fit the VGPRs a lane gets at that workgroup size (long fp32 lines fall back to sub-tile workgroups).  Choice: fewest passes,
then E closest to the precision's sweet spot (16 at fp64; 24 at fp32 -- the power-of-two fp32 kernels like 32, but the
mixed ones carry more address arithmetic per point and 40 points per thread left one 7-wave workgroup per CU: 1000^3 fp32
35.8 -> 31.1 ms with 20 points per thread), then the largest radix first.

usage: python tools/gen_mixed_configs.py        (rewrites the .inc; the file is committed)
"""
import math
import os

SIZES = [6, 10, 12, 20, 24, 36, 40, 48, 60, 72, 80, 96, 100, 112, 120, 144, 160, 192, 200, 224, 240, 250, 288, 300, 320, 384,
         400, 448, 480, 500, 576, 600, 640, 720, 768, 800, 896, 960, 1000, 1152, 1200, 1280, 1440, 1536, 1600, 1728, 1792,
         1920, 2000]
RADICES = [r for r in range(2, 33) if all(r % p for p in (11, 13, 17, 19, 23, 29, 31))]


def factorizations(n, maxk, smallest=2):
    """multisets of radices with product n, at most maxk of them"""
    if n == 1:
        yield ()
        return
    if maxk == 0:
        return
    for r in RADICES:
        if r < smallest or n % r:
            continue
        for rest in factorizations(n // r, maxk - 1, r):
            yield (r,) + rest


def lds_bytes(N, TW, r1, planes, rsz, npass):
    if npass < 2:
        return 0
    slots = N * TW
    ps = int(math.log2(r1 * TW))
    pws = min(4, int(math.log2(TW)))
    return planes * (slots + ((slots >> ps) << pws)) * rsz


def vgpr_cap(threads):
    """VGPRs a lane may use so that the whole workgroup fits on one CU (4 SIMDs x 512 VGPRs per lane slot)"""
    waves_per_simd = -(-(-(-threads // 64)) // 4)
    return min(256, (512 // waves_per_simd) & ~7)


def choose(N, prec):
    TL, rsz, epref, emax = (8, 8, 16, 32) if prec == "f64" else (16, 4, 24, 64)
    # VGPRs besides the E points (twiddles, butterfly temporaries, LDS / global addresses), fitted to the generated kernels'
    # resource remarks: fp64 46 (E = 20 at 800 threads fits the 128 it gets exactly); fp32 26 + 2E (16-line tiles and per-point
    # LDS indices: E = 24 -> 122, E = 40 -> 186), where a spill of a few registers is accepted: measured at 2000 x 1600 x
    # 1280 fp32, the 1600-point pass with 60 B/lane of scratch runs 19.4 ms, its sub-tile form without scratch 29.6 ms
    # (64-byte runs on the tiled side); the 2000-point pass with 416 B/lane 45.7 ms, its sub-tile form 22.6 ms
    overhead = (lambda E: 46) if prec == "f64" else (lambda E: 26 + 2 * E - 24)
    best = None
    for rad in factorizations(N, 4):
        lcm = 1
        for r in rad:
            lcm = lcm * r // math.gcd(lcm, r)
        for mult in (1, 2, 3, 4):
            E = lcm * mult
            if E > emax or N % E:
                continue
            NT = N // E
            # sub-tile workgroups (half the lines of a tile per workgroup, PassCfg::SUB): only where a whole tile does not
            # fit the register file (long fp32 lines)
            for sub in ((1, 2) if prec == "f32" and N >= 1024 else (1,)):
                TLK = TL // sub
                if NT * TLK > 1024:
                    continue
                G = 1
                while sub == 1 and NT * TL * G < 256 and NT * TL * G * 2 <= 1024 and G < 32:
                    G *= 2
                threads = NT * TLK * G
                if E * (rsz // 2) + overhead(E) > vgpr_cap(threads):
                    continue
                rr = tuple(sorted(rad, reverse=True))
                npass = len(rr)
                planes = 2 if lds_bytes(N, TLK * G, rr[0], 2, rsz, npass) <= 32 * 1024 else 1
                lds = lds_bytes(N, TLK * G, rr[0], planes, rsz, npass)
                if lds > 160 * 1024:
                    continue
                # fp32: a configuration whose register estimate (2E for the points + 26 + 2E) exceeds the 256 VGPRs of a lane
                # spills; it counts like one more pass.  Measured (profiles/r2_mixed_radix_vs_bluestein.txt): 1200^3 C2C
                # 76.0 -> 60.9 ms with (30, 10, 2, 2) on 30 points per thread instead of (30, 10, 4) on 60; the same trade at
                # 40 / 56 points per thread (no spills) LOSES: 1280^3 62.7 -> 67.4 ms, 896^3 18.3 -> 23.3 ms, 640^3 6.2 -> 6.8 ms
                heavy = 1 if prec == "f32" and 4 * E + 26 > 256 else 0
                score = (sub, npass + heavy, abs(E - epref), -rr[0], rr)
                if best is None or score < best[0]:
                    best = (score, dict(N=N, E=E, TL=TL, G=G, rad=rr, planes=planes, threads=threads, lds=lds, sub=sub))
    return None if best is None else best[1]


def choose_real(N, prec, base):
    """Configuration pair for the packed real z passes (R2C / C2R) of M = N complex points with the Hermitian split / merge IN
    REGISTERS (fft_r2c_kernel ONEPLANE = 2, fft_c2r_kernel PAIRED): the pass next to the split / merge -- the last one of the R2C
    chain, the first one of the C2R chain -- assigns its butterflies in conjugate pairs, which needs an even number of butterflies
    per thread in that pass, E / radix even.  Same constraints as choose(); prefers the complex configuration's own points per
    thread and radices (only their order changes).  None: the length keeps the split through LDS."""
    TL, rsz, epref, emax = (8, 8, 16, 32) if prec == "f64" else (16, 4, 24, 64)
    overhead = (lambda E: 46) if prec == "f64" else (lambda E: 26 + 2 * E - 24)
    best = None
    for rad in factorizations(N, 4):
        if len(rad) < 2:
            continue
        lcm = 1
        for r in rad:
            lcm = lcm * r // math.gcd(lcm, r)
        for mult in (1, 2, 3, 4):
            E = lcm * mult
            if E > emax or N % E:
                continue
            NT = N // E
            if NT * TL > 1024:
                continue
            G = 1
            while NT * TL * G < 256 and NT * TL * G * 2 <= 1024 and G < 32:
                G *= 2
            threads = NT * TL * G
            if E * (rsz // 2) + overhead(E) > vgpr_cap(threads):
                continue
            if prec == "f32" and 4 * E + 26 > 256:      # would spill (see choose()): such a length keeps the LDS form
                continue
            heavy = 0
            for last in sorted(set(rad), reverse=True):
                if (E // last) % 2:
                    continue
                rest = list(rad)
                rest.remove(last)
                rest.sort(reverse=True)
                r2c, c2r = tuple(rest) + (last,), (last,) + tuple(rest)
                lds = max(lds_bytes(N, TL * G, r2c[0], 1, rsz, len(rad)), lds_bytes(N, TL * G, c2r[0], 1, rsz, len(rad)))
                if lds > 160 * 1024:
                    continue
                same = 0 if (E == base["E"] and sorted(rad) == sorted(base["rad"])) else 1
                score = (len(rad) + heavy, same, abs(E - epref), -r2c[0], r2c)
                if best is None or score < best[0]:
                    best = (score, dict(N=N, E=E, TL=TL, G=G, r2c=r2c, c2r=c2r, threads=threads, lds=lds, npass=len(rad) + heavy))
    if best is None or best[1]["npass"] > len(base["rad"]):       # one more pass than the LDS form costs more than the split saves
        return None
    return best[1]


# compile parts per precision: (mixed_*.hip, rmixed_*.hip)
PARTS = {"f64": (2, 3), "f32": (4, 6)}


def main():
    out = []
    out.append("// kernels_mixed.inc -- GENERATED by tools/gen_mixed_configs.py; do not edit.")
    out.append("// Axis-pass configurations of the natively supported lengths that are not powers of two (mixed radix 2, 3, 5, 7).")
    out.append("//                      real   N     E  TL  G  radices      planes chain")
    for prec, real_t, tag in (("f64", "double", "F64"), ("f32", "float", "F32")):
        cfgs = []
        for N in SIZES:
            c = choose(N, prec)
            if c is None:
                out.append(f"// {tag}: no configuration for N = {N} (stays on the Bluestein kernel)")
                continue
            cfgs.append(c)
        out.append(f"#ifdef DFFT_MIXED_{tag}")
        for c in cfgs:
            r = list(c["rad"]) + [1] * (4 - len(c["rad"]))
            chain = 1 if max(c["rad"]) >= 16 and len(c["rad"]) > 1 else 0
            tail = ", 0, 0, 2" if c["sub"] == 2 else ""      # NTMEM, MAP, SUB
            out.append(f"using {tag}_M{c['N']} = PassCfg<{real_t}, {c['N']}, {c['E']}, {c['TL']}, {c['G']}, {r[0]}, {r[1]}, {r[2]}, {r[3]}, {c['planes']}, {chain}{tail}>;"
                       f"   // {c['threads']} threads, {c['lds']} B LDS")
        # Compile parts (mixed_*.hip / rmixed_*.hip, -DDFFT_PART=k): the configurations are dealt to PARTS[...] lists of about
        # equal compile cost (longest first onto the lightest list; cost ~ points per thread x (passes + 1), which is what
        # the unrolled code size follows).  fp32 kernels take ~2.7x as long to compile as fp64 ones, hence more parts.
        def deal(items, nparts):
            bins = [[0.0, []] for _ in range(nparts)]
            for c in sorted(items, key=lambda c: -(c["E"] * (len(c["rad"]) + 1))):
                b = min(bins, key=lambda b: b[0])
                b[0] += c["E"] * (len(c["rad"]) + 1)
                b[1].append(c)
            return [sorted(b[1], key=lambda c: c["N"]) for b in bins]

        nm, nr = PARTS[prec]
        for k, sel in enumerate(deal(cfgs, nm)):
            xs = " ".join(f"X({c['N']}, 0, {tag}_M{c['N']})" for c in sel)
            out.append(f"#define DFFT_{tag}_LIST_MIXED{k}(X) {xs}")
        out.append(f"#define DFFT_{tag}_MIXED_FOREACH_PART(P) " + " ".join(f"P({k})" for k in range(nm)))
        out.append(f"#define DFFT_{tag}_LIST_MIXED_ALL(X) " + " ".join(f"DFFT_{tag}_LIST_MIXED{k}(X)" for k in range(nm)))
        # packed real z passes (R2C / C2R of a real line of 2M points as an M-point complex transform + Hermitian split /
        # merge): every configuration with whole-tile workgroups; Y(M, cfg, ONEPLANE)
        real = [c for c in cfgs if c["sub"] == 1 and c["N"] <= 1024]
        paired = {c["N"]: choose_real(c["N"], prec, c) for c in real}
        for c in real:
            pr = paired[c["N"]]
            if pr is None:
                continue
            chain = 1 if max(pr["r2c"]) >= 16 else 0
            for kind, rad in (("R", pr["r2c"]), ("C", pr["c2r"])):
                r = list(rad) + [1] * (4 - len(rad))
                out.append(f"using {tag}_{kind}{c['N']} = PassCfg<{real_t}, {c['N']}, {pr['E']}, {pr['TL']}, {pr['G']}, {r[0]}, {r[1]}, {r[2]}, {r[3]}, 1, {chain}>;"
                           f"   // {pr['threads']} threads, {'R2C: pairs in the last pass' if kind == 'R' else 'C2R: pairs in the first pass'}")
        for k, sel in enumerate(deal(real, nr)):
            ys = []
            for c in sel:
                if paired[c["N"]] is not None:
                    ys.append(f"Y({c['N']}, {tag}_R{c['N']}, 2, {tag}_C{c['N']}, 2)")
                    continue
                plane = lds_bytes(c["N"], c["TL"] * c["G"], c["rad"][0], 1, 8 if prec == "f64" else 4, 2)
                # two planes when two workgroups with two planes each still fit a CU, else one plane after the other (two more
                # barriers, half the LDS: the mixed configurations have 256-512 threads, one workgroup per CU starves it)
                ys.append(f"Y({c['N']}, {tag}_M{c['N']}, {0 if 2 * plane <= 80 * 1024 else 1}, {tag}_M{c['N']}, 0)")
            out.append(f"#define DFFT_{tag}_LIST_RMIXED{k}(Y) {' '.join(ys)}")
        out.append(f"#define DFFT_{tag}_RMIXED_FOREACH_PART(P) " + " ".join(f"P({k})" for k in range(nr)))
        out.append("#endif")
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "distributedfft_amd", "csrc")
    with open(os.path.join(csrc, "kernels_mixed.inc"), "w") as f:
        f.write("\n".join(out) + "\n")
    with open(os.path.join(csrc, "kernels_mixed.mk"), "w") as f:      # part counts for the Makefile
        f.write("# GENERATED by tools/gen_mixed_configs.py: compile parts of mixed_*.hip / rmixed_*.hip\n")
        for prec in ("f64", "f32"):
            f.write(f"NM_{prec} = {PARTS[prec][0]}\nNR_{prec} = {PARTS[prec][1]}\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
