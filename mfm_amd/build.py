"""Build libmfm_hip.so for gfx950 with hipcc (in-tree: mfm_amd/lib/)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmfm_hip.so")
SOURCES = ["api.hip"]          # unity build: api.hip includes the kernel translation units
SCHED_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-kernarg-preload-count=12"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mfm.h"), os.path.abspath(__file__)]      # (this file: the flags)
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-sched-strategy=max-ilp: the machine scheduler orders for instruction-level parallelism instead of for occupancy.  Every
    # hot kernel here runs at a FIXED occupancy (8 waves per CU with up to 256 registers, set by its LDS tile), so there is nothing for
    # the default strategy's register-pressure heuristics to win; measured on the same chain states (tools/flow_ab.py): flow step
    # 48.68 -> 46.04 ms, bench 7.56 -> 7.90 M chain-steps/s (0.710 -> 0.748 of the MFMA peak); the other workloads within +-0.7 %
    # (tools/dbg/ab_workloads.sh).  Same arithmetic: attempt counts and results are unchanged.
    # -amdgpu-kernarg-preload-count: scalar kernel arguments at the head of the list arrive in SGPRs with the wave instead of through a read
    # of the argument segment; wgrad_sk and the training kernel repeat the pointers of their first loads there (tools/dbg/ab_iter.sh on one
    # box: iteration 78.4 -> 77.8 us).  Same fallback as the scheduling strategy.
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]
    tail = ["-o", LIB] + SOURCES
    cmd = base + SCHED_FLAGS + tail
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0 and ("amdgpu-sched-strategy" in r.stderr or "Unknown command line argument" in r.stderr):
        # -amdgpu-sched-strategy is an internal LLVM option: a ROCm update may drop or rename it.  The library is the same
        # arithmetic without it (the default strategy: ~5 % slower flow step), so build without rather than not at all.
        print("mfm_amd.build: hipcc rejected %s -- rebuilding with the default scheduling strategy\n%s" % (" ".join(SCHED_FLAGS), r.stderr.strip()[-400:]),
              file=sys.stderr)
        cmd = base + tail
        r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    if r.stderr and (verbose or r.returncode != 0):
        sys.stderr.write(r.stderr)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd, stderr=r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
