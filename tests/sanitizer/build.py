"""Builds libssc's HOST side with AddressSanitizer + UndefinedBehaviorSanitizer (device code untouched:
-fno-gpu-sanitize; GPU ASan is not available on this pool) and the C driver tests/sanitizer/abi_args.c against it.
Returns the path of the driver executable.  Objects are cached under tests/_build/asan/."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "smartstartcontinuous_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "_build", "asan")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SAN = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build(jobs=8):
    os.makedirs(OUT, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "ssc.h")]
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    procs, objs = [], []
    for f in srcs:
        o = os.path.join(OUT, os.path.splitext(f)[0] + ".o")
        objs.append(o)
        if _newer(o, [os.path.join(CSRC, f)] + hdrs):
            continue
        cmd = [HIPCC, "-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
               "-munsafe-fp-atomics", "-x", "hip", "-c", os.path.join(CSRC, f), "-o", o] + SAN
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) >= jobs:
            _drain(procs)
    _drain(procs)
    lib = os.path.join(OUT, "libssc_asan.so")
    if not _newer(lib, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-o", lib] + objs + SAN)
    exe = os.path.join(OUT, "abi_args")
    src = os.path.join(HERE, "abi_args.c")
    if not _newer(exe, [src, lib] + hdrs):
        subprocess.check_call([HIPCC, "-O1", "-g", "-x", "c", src, "-I" + os.path.join(ROOT, "include"), "-o", exe,
                               "-L" + OUT, "-lssc_asan", "-Wl,-rpath," + OUT] + SAN)
    return exe


def _drain(procs):
    while procs:
        f, p = procs.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("sanitizer build of %s failed:\n%s" % (f, out[-3000:]))


if __name__ == "__main__":
    print(build())
