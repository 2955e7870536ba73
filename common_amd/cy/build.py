"""Build the Cython extension(s) in-tree:  python common_amd/cy/build.py [extra.pyx ...]

cythonize + g++ (C++17) against include/ and libmicroscopes_hip.so (rpath $ORIGIN/../lib, so the built module finds
the library wherever the tree is copied).  Called by __graft_entry__.build(); tests build their downstream probe module
with build_module() the same way a downstream package would."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def build_module(pyx, out_dir=None, force=False):
    """pyx -> <out_dir>/<name><EXT_SUFFIX>; returns the path of the shared object"""
    from Cython.Build import cythonize  # noqa: F401  (fail early when Cython is missing)
    pyx = os.path.abspath(pyx)
    out_dir = out_dir or os.path.dirname(pyx)
    name = os.path.splitext(os.path.basename(pyx))[0]
    so = os.path.join(out_dir, name + sysconfig.get_config_var("EXT_SUFFIX"))
    deps = [pyx] + [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".pxd")]
    hdr = os.path.join(ROOT, "include", "microscopes_amd")
    deps += [os.path.join(hdr, f) for f in os.listdir(hdr)]
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
        return so
    cpp = os.path.join(out_dir, name + ".cpp")
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", "-I", ROOT, pyx, "-o", cpp])
    libdir = os.path.join(ROOT, "common_amd", "lib")
    rpath = os.path.relpath(libdir, out_dir)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-Wno-deprecated-declarations",
           "-I", sysconfig.get_paths()["include"], "-I", os.path.join(ROOT, "include"), cpp, "-o", so,
           "-L", libdir, "-lmicroscopes_hip", "-Wl,-rpath,$ORIGIN/" + rpath]
    subprocess.check_call(cmd)
    os.remove(cpp)
    return so


def main(argv):
    for pyx in [os.path.join(HERE, "_models.pyx")] + list(argv):
        print(build_module(pyx))


if __name__ == "__main__":
    main(sys.argv[1:])
