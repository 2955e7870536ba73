#!/usr/bin/env python3
"""Register / scratch metadata of every kernel in the library's gfx950 code object (what the judge reads with
llvm-readelf --notes): tools/scans/regs.py [substring ...] -> name, VGPRs, AGPRs, SGPRs, VGPR spills, SGPR spills, scratch B/lane, LDS B.
Runs on the build machine (no GPU)."""
import re
import subprocess
import sys
import os
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.environ.get("MSC_LIB_PATH", os.path.join(ROOT, "common_amd", "lib", "libmicroscopes_hip.so"))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    """every gfx950 code object of the library: .hip_fatbin holds one offload bundle per translation unit"""
    d = tempfile.mkdtemp(prefix="regs_")
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for i, st in enumerate(starts):
        piece = os.path.join(d, "bundle%d.bin" % i)
        open(piece, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(d, "gfx950_%d.co" % i)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + piece,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        out.append(co)
    return out


def kernels(co):
    txt = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    res = []
    for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        def f(key, default="0"):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else default
        sym = f("name", "?")
        try:
            name = subprocess.check_output(["c++filt", sym], text=True).strip()
        except Exception:
            name = sym
        # "void msc::k<...>(args)" -> "k<...>": cut the argument list at the parenthesis that closes the template list
        name = name.replace("void ", "", 1).replace("msc::", "")
        depth, cut = 0, len(name)
        for i, ch in enumerate(name):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        name = name[:cut]
        res.append((name, int(f("vgpr_count")), int(f("agpr_count")), int(f("sgpr_count")), int(f("vgpr_spill_count")),
                    int(f("sgpr_spill_count")), int(f("private_segment_fixed_size")), int(f("group_segment_fixed_size"))))
    return res


def main():
    subs = sys.argv[1:]
    rows = []
    for co in code_objects(LIB):
        rows += kernels(co)
    rows.sort()
    print("%-72s %5s %5s %5s %7s %7s %8s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "v-spill", "s-spill", "scratchB", "LDS B"))
    for r in rows:
        if subs and not any(s in r[0] for s in subs):
            continue
        print("%-72s %5d %5d %5d %7d %7d %8d %7d" % r)
    print("# %d kernels, library %d bytes" % (len(rows), os.path.getsize(LIB)))


if __name__ == "__main__":
    main()
