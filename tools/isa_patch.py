#!/usr/bin/env python3
"""One-variable-at-a-time ISA experiments on a compiled code object (round 5: the bf16 separable producers' non-repeatability).

Reads the gfx950 assembly hipcc emitted for a .hip file (`hipcc -S --cuda-device-only`), rewrites ONE thing about the
packed-FP32 instructions and writes the assembly back; tools/isa_variant_build.sh assembles it, wraps it as the fat binary and links a
library with it, so two builds differ in exactly the edited instructions.

    python tools/isa_patch.py audit   in.s                 # list v_pk_{fma,mul,add}_f32 whose destination pair is also an op_sel-CROSSED source
    python tools/isa_patch.py VARIANT in.s out.s [--only SUBSTR]

A packed op `v_pk_fma_f32 D, S0, S1, S2 op_sel:[a,b,c] op_sel_hi:[d,e,f]` computes D.lo from S0[a], S1[b], S2[c] and D.hi from S0[d],
S1[e], S2[f] (0 = low register of the pair, 1 = high; defaults op_sel 0, op_sel_hi 1).  Source i is CROSSED when a_i = 1 or d_i = 0 is set
such that one half of the result reads the other half's register.  "crossed overlap" = D is the same pair as a crossed source: the low
result's input register is the high result's output register (or the other way round).

Variants (all keep the arithmetic bit-identical):
  ctrl     nothing changed (pipeline control: must behave like the hipcc-built library)
  fresh    every crossed-overlap op writes a FRESH pair (v[250:251]) and a v_pk_mov_b32 copies it to D      [breaks the overlap]
  pad      every crossed-overlap op keeps D and is followed by an independent v_pk_mov_b32 on v[250:253]     [same instruction count as fresh]
  nop      every crossed-overlap op is followed by `s_nop 3`                                                 [4 wait states before any consumer]
  unalias  `v_cvt_pk_bf16_f32 vN+1, vN, vN+1 ; v_cvt_pk_bf16_f32 vN, vP, vQ ; ds_write_b64 vA, v[N:N+1]` converts into v[252:253]
           and stores those                                                                                  [converts no longer write their own sources]
  cvtnop   `s_nop 3` between the last packed op and the first of the two converts of that triple
  uncross_lo  every source whose op_sel bit is 1 (the LOW result reads the HIGH register of the pair) is first copied, with two
           v_mov_b32, into a temporary pair that holds the wanted registers straight, and the bit is dropped; uncross_hi does the same
           for op_sel_hi = 0 (the HIGH result reads the LOW register); uncross both                          [no crossed operand reads]
  fmanopN  (N = 0..7) `s_nop N` in front of every v_pk_fma_f32 whose ADDEND pair (src2) was written by a packed op within the four
           instructions before it                                                                            [distance of the dependent pair]
"""
import re
import sys

PK = re.compile(r"^\s*v_pk_(fma|mul|add)_f32\s+v\[(\d+):(\d+)\],\s*(.*)$")
SRC = re.compile(r"v\[(\d+):(\d+)\]")


def parse_pk(line):
    m = PK.match(line)
    if not m:
        return None
    d = int(m.group(2))
    rest = m.group(4)
    ops = rest.split(" op_sel")[0]
    srcs = [int(x.group(1)) for x in SRC.finditer(ops)]
    n = 3 if m.group(1) == "fma" else 2
    # operands may also be literals / SGPR pairs: keep positions by splitting on commas
    parts = [p.strip() for p in ops.split(",")][:n]
    pairs = []
    for p in parts:
        mm = SRC.fullmatch(p)
        pairs.append(int(mm.group(1)) if mm else None)
    sel = [0] * n
    sel_hi = [1] * n
    ms = re.search(r"op_sel:\[([\d,]+)\]", rest)
    if ms:
        v = [int(x) for x in ms.group(1).split(",")]
        sel[:len(v)] = v
    mh = re.search(r"op_sel_hi:\[([\d,]+)\]", rest)
    if mh:
        v = [int(x) for x in mh.group(1).split(",")]
        sel_hi[:len(v)] = v
    crossed_overlap = any(pairs[i] == d and (sel[i] == 1 or sel_hi[i] == 0) for i in range(n))
    plain_overlap = any(pairs[i] == d for i in range(n))
    return {"d": d, "pairs": pairs, "sel": sel, "sel_hi": sel_hi, "crossed_overlap": crossed_overlap, "overlap": plain_overlap, "srcs": srcs}


def kernels(lines):
    """yield (name, first line index, last line index) of every function body"""
    name, start = None, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, start = m.group(1), i
        elif l.startswith(".Lfunc_end") and name is not None:
            yield name, start, i
            name = None


def audit(path):
    lines = open(path).read().split("\n")
    total = 0
    for name, a, b in kernels(lines):
        n_cross = n_pk = 0
        for l in lines[a:b]:
            p = parse_pk(l)
            if p:
                n_pk += 1
                n_cross += p["crossed_overlap"]
        if n_pk:
            print(f"{n_cross:5d} crossed-overlap of {n_pk:5d} packed-f32 ops  {name}")
        total += n_cross
    print(f"total crossed-overlap packed ops: {total}")
    return total


def bump_vgprs(lines, a_meta_names):
    """kernels we touched use v[250:253]: raise their register count to 256 (two waves per SIMD at 512 threads: always available)"""
    out = []
    cur = None
    for l in lines:
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", l)
        if m:
            cur = m.group(1)
        if cur in a_meta_names:
            if ".amdhsa_next_free_vgpr" in l:
                l = re.sub(r"\d+", "256", l, count=1)
            elif ".amdhsa_accum_offset" in l:
                l = re.sub(r"\d+", "256", l, count=1)
        if ".end_amdhsa_kernel" in l:
            cur = None
        out.append(l)
    # metadata (informational): .vgpr_count of the touched kernels
    txt = "\n".join(out)
    for n in a_meta_names:
        txt = re.sub(r"(\.name:\s+" + re.escape(n) + r"\n(?:.*\n)*?\s+\.vgpr_count:\s+)\d+", r"\g<1>256", txt, count=1)
    return txt.split("\n")


def patch(variant, src, dst, only):
    lines = open(src).read().split("\n")
    out = list(lines)
    touched = set()
    edits = 0
    # work back to front so indices stay valid
    spans = list(kernels(lines))
    for name, a, b in reversed(spans):
        if only and only not in name:
            continue
        body = out[a:b]
        new = []
        i = 0
        while i < len(body):
            l = body[i]
            p = parse_pk(l)
            if variant in ("fresh", "pad", "nop") and p and p["crossed_overlap"]:
                d = p["d"]
                if variant == "fresh":
                    new.append(re.sub(r"(v_pk_\w+_f32\s+)v\[%d:%d\]" % (d, d + 1), r"\g<1>v[250:251]", l, count=1))
                    new.append("\tv_pk_mov_b32 v[%d:%d], v[250:251], v[250:251] op_sel:[0,1]" % (d, d + 1))
                elif variant == "pad":
                    new.append(l)
                    new.append("\tv_pk_mov_b32 v[250:251], v[252:253], v[252:253] op_sel:[0,1]")
                else:
                    new.append(l)
                    new.append("\ts_nop 3")
                touched.add(name)
                edits += 1
                i += 1
                continue
            if variant.startswith("uncross") and p:
                n = len(p["pairs"])
                m = PK.match(l)
                ops = [t.strip() for t in m.group(4).split(" op_sel")[0].split(",")][:n]
                sel, sel_hi = list(p["sel"]), list(p["sel_hi"])
                pre, tmp = [], 244
                for k in range(n):
                    if p["pairs"][k] is None:
                        continue
                    fix_lo = sel[k] == 1 and variant in ("uncross", "uncross_lo")
                    fix_hi = sel_hi[k] == 0 and variant in ("uncross", "uncross_hi")
                    if fix_lo or fix_hi:
                        base = p["pairs"][k]
                        pre.append("\tv_mov_b32_e32 v%d, v%d" % (tmp, base + sel[k]))
                        pre.append("\tv_mov_b32_e32 v%d, v%d" % (tmp + 1, base + sel_hi[k]))
                        ops[k] = "v[%d:%d]" % (tmp, tmp + 1)
                        sel[k], sel_hi[k] = 0, 1
                        tmp += 2
                if pre:
                    txt = "\tv_pk_%s_f32 v[%d:%d], %s" % (m.group(1), p["d"], p["d"] + 1, ", ".join(ops))
                    if any(sel):
                        txt += " op_sel:[%s]" % ",".join(str(v) for v in sel)
                    if not all(sel_hi):
                        txt += " op_sel_hi:[%s]" % ",".join(str(v) for v in sel_hi)
                    new += pre
                    new.append(txt)
                    touched.add(name)
                    edits += 1
                    i += 1
                    continue
            if variant.startswith("fmanop") and p and len(p["pairs"]) == 3 and p["pairs"][2] is not None:
                recent = []
                for back in new[-4:]:
                    q = parse_pk(back)
                    if q:
                        recent.append(q["d"])
                if p["pairs"][2] in recent:
                    new.append("\ts_nop %d" % int(variant[6:]))
                    touched.add(name)
                    edits += 1
            if variant in ("unalias", "cvtnop"):
                m1 = re.match(r"^\s*v_cvt_pk_bf16_f32 v(\d+), v(\d+), v(\d+)\s*$", l)
                if m1 and i + 2 < len(body):
                    hi, s0, s1 = (int(x) for x in m1.groups())
                    m2 = re.match(r"^\s*v_cvt_pk_bf16_f32 v(\d+), v(\d+), v(\d+)\s*$", body[i + 1])
                    m3 = re.match(r"^(\s*ds_write_b64 v\d+, )v\[(\d+):(\d+)\](.*)$", body[i + 2])
                    if m2 and m3 and hi == s1 and s0 == hi - 1 and int(m2.group(1)) == hi - 1 and int(m3.group(2)) == hi - 1:
                        if variant == "unalias":
                            new.append("\tv_cvt_pk_bf16_f32 v253, v%d, v%d" % (s0, s1))
                            new.append("\tv_cvt_pk_bf16_f32 v252, v%s, v%s" % (m2.group(2), m2.group(3)))
                            new.append(m3.group(1) + "v[252:253]" + m3.group(4))
                        else:
                            new.append("\ts_nop 3")
                            new += body[i:i + 3]
                        touched.add(name)
                        edits += 1
                        i += 3
                        continue
            new.append(l)
            i += 1
        out[a:b] = new
    if variant in ("fresh", "pad", "unalias") or variant.startswith("uncross"):
        out = bump_vgprs(out, touched)
    open(dst, "w").write("\n".join(out))
    print(f"{variant}: {edits} edits in {len(touched)} kernels -> {dst}")


if __name__ == "__main__":
    if sys.argv[1] == "audit":
        audit(sys.argv[2])
    else:
        only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
        patch(sys.argv[1], sys.argv[2], sys.argv[3], only)
