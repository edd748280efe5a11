"""Build the test double of librccl (host code only; g++ against the HIP runtime and RCCL's header) in-tree, so that it
travels to the GPU box with the snapshot.  No pytest import: __graft_entry__.build() calls this too."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
DOUBLE_SRC = os.path.join(HERE, "rccl_double.cpp")
DOUBLE_LIB = os.path.join(HERE, "librccl_double.so")


def build_double(force=False):
    if force or not os.path.exists(DOUBLE_LIB) or os.path.getmtime(DOUBLE_LIB) < os.path.getmtime(DOUBLE_SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__",
                               "-I/opt/rocm/include", DOUBLE_SRC, "-o", DOUBLE_LIB,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return DOUBLE_LIB
