"""Builds ``libmi_restore.so`` in-tree with hipcc for gfx950 (no JIT cache, no pip install)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi_restore.so")


def build(verbose: bool = False, jobs: int = 8) -> str:
    """make -C csrc; returns the library path.  Raises CalledProcessError with the compiler output on failure."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libmi_restore.so failed:\n" + res.stdout[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
