#!/usr/bin/env python3
"""Build the REFERENCE's own Cython bitboard into oracle/_ref/ (test infrastructure only).

The sources stay where they lie under /root/reference (src/cython/bitboard.pyx + .pxd); only
build outputs land in oracle/_ref/, which is git-ignored AND listed in .gpurunignore: the compiled reference
module never leaves this container (the reference cannot travel in any form).  The generated C file is deleted
again once the extension module is compiled, so that only machine code is kept.  Compile directives follow the reference's own recipe (/root/reference/setup.py:22-30:
-O3, language_level 3, boundscheck/wraparound off, cdivision on).

Nothing in the product path (othello_reinforcement_learning_test_amd/) imports this.  Used by
  * tests/golden/make_golden.py  -- to generate the committed golden vectors, and
  * tests (CPU suite)            -- optional live cross-check of the C restatement.
On the GPU box neither /root/reference nor oracle/_ref exists; this script then does nothing and the tests that
cross-check against the live reference skip themselves (the committed golden vectors carry its answers).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("OTHELLO_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
PKG = os.path.join(OUT, "src", "cython")


def built_path():
    return os.path.join(PKG, "bitboard" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False):
    pyx = os.path.join(REF, "src", "cython", "bitboard.pyx")
    if not os.path.exists(pyx):
        return None  # reference not present (GPU box): nothing to build
    so = built_path()
    if (not force and os.path.exists(so)
            and os.path.getmtime(so) >= os.path.getmtime(pyx)):
        return so
    import numpy as np
    os.makedirs(PKG, exist_ok=True)
    c_file = os.path.join(PKG, "bitboard.c")
    # cythonize from the reference location, output C into oracle/_ref only
    subprocess.check_call([
        sys.executable, "-m", "cython", "-3",
        "-X", "boundscheck=False", "-X", "wraparound=False", "-X", "cdivision=True",
        "-I", os.path.join(REF, "src", "cython"),
        "--module-name", "src.cython.bitboard",
        "-o", c_file, pyx,
    ])
    inc = sysconfig.get_paths()["include"]
    subprocess.check_call([
        "gcc", "-O3", "-fPIC", "-shared", "-w",
        "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
        "-I", inc, "-I", np.get_include(), c_file, "-o", so,
    ])
    os.remove(c_file)   # keep only the compiled module: the generated C is a transcript of the reference source
    return so


def import_reference():
    """Make `import src...` resolve to /root/reference with the freshly built bitboard.

    tensorboard is absent from this image and src/train/__init__.py imports the trainer eagerly
    (SURVEY.md section 5), so a no-op SummaryWriter is injected on the oracle side only.
    """
    import types
    so = build()
    if so is None:
        raise RuntimeError("reference tree not available at %s" % REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "torch.utils.tensorboard" not in sys.modules:
        try:
            import torch.utils.tensorboard  # noqa: F401
        except Exception:
            tb = types.ModuleType("torch.utils.tensorboard")

            class SummaryWriter:  # no-op stand-in for logging only
                def __init__(self, *a, **k):
                    pass

                def add_scalar(self, *a, **k):
                    pass

                def close(self):
                    pass

            tb.SummaryWriter = SummaryWriter
            sys.modules["torch.utils.tensorboard"] = tb
    import src.cython as refcy
    if PKG not in list(refcy.__path__):
        refcy.__path__.append(PKG)
    import src.cython.bitboard as bb
    return bb


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print("oracle/_ref:", p if p else "reference not present; skipped")
