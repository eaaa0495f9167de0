"""Builds tests/_build/libssc_host_harness.so (host code only; hipcc is used so that the
__host__ __device__ functions compile unchanged)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "tests", "host_harness", "harness.hip")
OUT = os.path.join(ROOT, "tests", "_build", "libssc_host_harness.so")
DEPS = [SRC, os.path.join(ROOT, "smartstartcontinuous_amd", "csrc", "ssc_device.h"),
        os.path.join(ROOT, "include", "ssc.h")]


def build():
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
                           "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "smartstartcontinuous_amd", "csrc"), SRC, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build())
