"""Builds tests/fakerccl/libfakerccl.so (hipcc; cross-compiles without a GPU).  Test infrastructure."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libfakerccl.so")


def build(force=False):
    src = os.path.join(HERE, "fakerccl.cpp")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", src, "-o", LIB, "-lrt", "-lpthread"],
                   check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
