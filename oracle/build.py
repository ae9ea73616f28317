"""Build recipe for the ORACLE (test infrastructure only — see oracle/README.md).

  python oracle/build.py            builds oracle/liboracle.so (gcc, the repo's own C restatement)
                                    and, when /root/reference is present (build container only),
                                    oracle/_ref/cpu_nms*.so = the reference's own Cython Soft-NMS.

oracle/_ref: compiled from the reference source where it lies
(/root/reference/ext/nms/nms/cpu_nms.pyx, lines 1-120 = imports, max/min helpers and
cpu_soft_nms; the rest of that file (cpu_nms) does not cythonize under numpy 2 —
`np.int_t`/`np.float` were removed — and is not on the hot path).  Intermediates
(.pyx slice, generated .c, build/) are deleted; only the .so stays, git-ignored.
No reference source is copied into the repository.
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PYX = "/root/reference/ext/nms/nms/cpu_nms.pyx"


def build_oracle(force=False):
    out = os.path.join(HERE, "liboracle.so")
    srcs = [os.path.join(HERE, "soft_nms.c")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-o", out] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    return out


def build_ref(force=False):
    """Compile the reference's cpu_soft_nms in place -> oracle/_ref/.  No-op off the build container."""
    refdir = os.path.join(HERE, "_ref")
    existing = glob.glob(os.path.join(refdir, "cpu_nms*.so"))
    if not os.path.exists(REF_PYX):
        return existing[0] if existing else None
    if existing and not force:
        return existing[0]
    os.makedirs(refdir, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="rr_ref_")
    try:
        with open(REF_PYX) as f:
            head = f.readlines()[:120]
        with open(os.path.join(tmp, "cpu_nms.pyx"), "w") as f:
            f.writelines(head)
        setup = (
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "import numpy\n"
            "setup(ext_modules=cythonize([Extension('cpu_nms', ['cpu_nms.pyx'],"
            " include_dirs=[numpy.get_include()])], language_level=3),"
            " script_args=['build_ext', '--inplace'])\n")
        with open(os.path.join(tmp, "setup_ref.py"), "w") as f:
            f.write(setup)
        subprocess.check_call([sys.executable, "setup_ref.py"], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
        so = glob.glob(os.path.join(tmp, "cpu_nms*.so"))[0]
        dst = os.path.join(refdir, os.path.basename(so))
        shutil.copy(so, dst)
        return dst
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print("oracle:", build_oracle(force="--force" in sys.argv))
    print("oracle/_ref:", build_ref(force="--force" in sys.argv))
