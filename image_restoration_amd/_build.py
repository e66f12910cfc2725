"""Builds ``libmi_restore.so`` in-tree with hipcc for gfx950 (no JIT cache, no pip install)."""
import glob
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi_restore.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                  [os.path.join(CSRC, "Makefile"), os.path.join(HERE, "..", "include", "mi_restore.h")])


def fresh() -> bool:
    """True when the library exists and is newer than every source it is built from (decided in Python: no child process)."""
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(s) <= t for s in sources() if os.path.exists(s))


def _profiled() -> bool:
    """Under rocprofv3 the preloaded tool library initialises the GPU in every child too: a `make` child that goes on to exec
    hipcc is the exec-after-GPU-init this pool forbids."""
    env = os.environ
    return any(k in env for k in ("ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or \
        "rocprof" in env.get("LD_PRELOAD", "")


def build(verbose: bool = False, jobs: int = 8) -> str:
    """Returns the library path, running ``make -C csrc`` only when a source is newer than the library.  Call it BEFORE the
    process touches the GPU (bench.py and the test session do); a stale library inside a profiled process is refused."""
    if fresh():
        return LIB
    if _profiled():
        raise RuntimeError("libmi_restore.so is older than its sources and this process runs under rocprofv3: build first "
                           "(python -m image_restoration_amd._build), then profile")
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libmi_restore.so failed:\n" + res.stdout[-4000:])
    os.utime(LIB, None)             # an up-to-date make leaves the old mtime: stamp it so that fresh() holds from now on
    return LIB


def wait_fresh(timeout_s: float = 900.0) -> str:
    """Ranks other than 0 of a multi-process launch: wait for rank 0's build instead of racing it on the object files."""
    t0 = time.time()
    while not fresh():
        if time.time() - t0 > timeout_s:
            raise RuntimeError("timed out waiting for libmi_restore.so to be built by rank 0")
        time.sleep(1.0)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
