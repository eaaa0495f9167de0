"""Build libssc.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m smartstartcontinuous_amd.build
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


def build(verbose=False, jobs=4):
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stdout.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libssc.so failed (see output above)")
    out = os.path.join(HERE, "libssc.so")
    if not os.path.exists(out):
        raise RuntimeError(f"{out} was not produced")
    return out


if __name__ == "__main__":
    print(build(verbose=True))
