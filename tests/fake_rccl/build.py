"""Builds tests/fake_rccl/librccl_fake.so — the test-only stand-in for librccl (see fake_rccl.cpp). Host code only: g++ against
the HIP runtime. The .so is git-ignored and travels to the GPU box with the snapshot, like the product library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.cpp")
LIB = os.path.join(HERE, "librccl_fake.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", SRC, "-o", LIB,
                           "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
