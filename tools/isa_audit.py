#!/usr/bin/env python3
"""Audit the gfx950 code objects of libpnpadmm for the packed-FP32 operand form that profiles/r05_race.md found faulty next to bf16 MFMAs.

The fault (MI355X, ROCm 7.2): in a wave that shares its SIMD with a wave streaming v_mfma_*_bf16, a v_pk_{mul,fma,add}_f32 whose LOW
result reads the HIGH register of a source pair (op_sel bit = 1 for that source) sometimes sees ZERO for that operand in lanes 48-63.
Same instruction stream with the operand copied into a straight pair first: clean (tools/isa_patch.py `uncross_lo`).

    python tools/isa_audit.py --lib [libpnpadmm.so] # the code objects inside the BUILT library (default: the in-tree one) - seconds
    python tools/isa_audit.py [file.hip ...]        # compile to ISA first (default: every .hip under dt4image_restoration_amd/csrc) - minutes
    python tools/isa_audit.py --asm file.s

Per kernel: packed-f32 ops, those with an op_sel = 1 source ("low reads high"), and the matrix instructions of the kernel.  Exit code 1 if a
kernel with bf16 / f16 / fp8 MFMAs (the matrix pipe runs beside the vector pipe: co-execution) holds such an op - `make -C csrc audit` and
tests/test_host_logic.py run it on the bf16 kernels' sources.  Kernels whose only MFMAs are f32 (they hold the vector port)
and kernels without MFMAs are listed, not failed: 350 such ops in the FFT and F(4x4) kernels are stress-clean.
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_patch as ip  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dt4image_restoration_amd", "csrc")


def asm_of(src, extra=()):
    out = "/tmp/_isa_audit_%s.s" % os.path.basename(src)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", src, "-o", out, *extra], check=True, stderr=subprocess.DEVNULL)
    return out


def code_objects(lib):
    """the gfx950 code objects embedded in a host library: every __CLANG_OFFLOAD_BUNDLE__ in it (one per translation unit)"""
    blob = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], blob.find(magic)
    while at >= 0:
        import struct
        n = struct.unpack_from("<Q", blob, at + len(magic))[0]
        q = at + len(magic) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "amdgcn" in ident and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(magic, at + len(magic))
    return out


def disassemble(lib):
    """llvm-objdump of every embedded code object, rewritten to the `name:` / instruction / `.Lfunc_end` shape audit_asm() reads"""
    paths = []
    for i, co in enumerate(code_objects(lib)):
        f = "/tmp/_isa_audit_co%d.o" % i
        open(f, "wb").write(co)
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f], check=True, capture_output=True, text=True).stdout
        lines = []
        for l in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", l)
            if m:
                if lines:
                    lines.append(".Lfunc_end")
                lines.append(m.group(1) + ":")
            else:
                lines.append(l.split("//")[0].rstrip())
        lines.append(".Lfunc_end")
        out = "/tmp/_isa_audit_co%d.s" % i
        open(out, "w").write("\n".join(lines))
        paths.append(out)
    return paths


def audit_asm(path, verbose=True):
    lines = open(path).read().split("\n")
    bad = 0
    rows = []
    for name, a, b in ip.kernels(lines):
        n_pk = n_lohi = 0
        mf = {}
        for l in lines[a:b]:
            p = ip.parse_pk(l)
            if p:
                n_pk += 1
                if any(p["pairs"][k] is not None and p["sel"][k] == 1 for k in range(len(p["pairs"]))):
                    n_lohi += 1
            m = re.match(r"\s*(v_mfma_\w+|v_smfmac_\w+)", l)
            if m:
                mf[m.group(1)] = mf.get(m.group(1), 0) + 1
        coexec = any(not k.endswith("_f32") or "xf32" in k for k in mf) and any(re.search(r"bf16|f16|fp8|bf8|f8f6f4|i8", k) for k in mf)
        flagged = coexec and n_lohi > 0
        bad += flagged
        rows.append((name, n_pk, n_lohi, mf, flagged))
    if verbose:
        for name, n_pk, n_lohi, mf, flagged in rows:
            if n_lohi or flagged:
                print("%s %5d low-reads-high of %5d packed-f32 ops  mfma: %s  %s" % ("FAIL" if flagged else "  ok", n_lohi, n_pk,
                      ",".join(sorted(mf)) or "-", name))
        print("%s: %d kernels, %d packed ops with a low-reads-high source, %d kernels flagged" %
              (os.path.basename(path), len(rows), sum(r[2] for r in rows), bad))
    return bad, rows


def main():
    args = sys.argv[1:]
    if args and args[0] == "--asm":
        sys.exit(1 if audit_asm(args[1])[0] else 0)
    if args and args[0] == "--lib":
        lib = args[1] if len(args) > 1 else os.path.join(CSRC, "libpnpadmm.so")
        bad = sum(audit_asm(f)[0] for f in disassemble(lib))
        sys.exit(1 if bad else 0)
    srcs = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for s in srcs:
        bad += audit_asm(asm_of(s))[0]
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
