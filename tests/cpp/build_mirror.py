"""Build tests/cpp/test_mirror (test infrastructure: the only host program that links the oracle library)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TEST_BIN = os.path.join(HERE, "_build", "test_mirror")


def build(force=False, verbose=False):
    from kmerutils_amd import build as kbuild
    kbuild.build()
    odir = os.path.join(ROOT, "oracle", "_build")
    oracle_so = os.path.join(odir, "libkmu_oracle.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    extra = ["-L" + odir, "-lkmu_oracle", "-Wl,-rpath,$ORIGIN/" + os.path.relpath(odir, os.path.dirname(TEST_BIN))]
    return kbuild.compile_host(os.path.join(HERE, "test_mirror.cpp"), TEST_BIN, extra=extra, force=force, verbose=verbose,
                               deps=[oracle_so, os.path.join(ROOT, "oracle", "kmu_oracle.h")])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
