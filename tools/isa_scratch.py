"""Where does a kernel spill?  Lists scratch_load/store counts per basic block with the block's loop depth.
    python tools/isa_scratch.py <file.hip> <mangled-name-substring>"""
import re, subprocess, sys, os
src, pat = sys.argv[1], sys.argv[2]
out = "/tmp/_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.dirname(os.path.abspath(src)),
                "-S", "--cuda-device-only", src, "-o", out] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r"^(_Z\w+):.*\n", s, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    body = s[m.end():s.index(".Lfunc_end", m.end())].split("\n")
    blk, depth, rows = "entry", 0, {}
    for l in body:
        t = l.strip()
        mm = re.match(r"(\.LBB\d+_\d+):(.*)", t)
        if mm:
            blk = mm.group(1)
            d = re.search(r"Depth=(\d+)", mm.group(2))
            depth = int(d.group(1)) if d else 0
            rows.setdefault(blk, [depth, 0, 0, 0])
            continue
        r = rows.setdefault(blk, [depth, 0, 0, 0])
        if t.startswith("scratch_load"): r[1] += 1
        elif t.startswith("scratch_store"): r[2] += 1
        elif t.startswith("v_mfma"): r[3] += 1
    print(name)
    for b, (d, ld, st, mf) in rows.items():
        if ld or st or mf:
            print(f"  {b:12s} depth {d}  scratch loads {ld:3d} stores {st:3d}  mfma {mf}")
