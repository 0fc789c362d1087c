"""In-tree build of libgaviko_hip.so (hipcc, gfx950 only).  `python -m gaviko_amd.build [--force]`.

The .so is git-ignored but travels with the gpurun snapshot; nothing is JIT-compiled at run time.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libgaviko_hip.so")
LIB_DIAG = os.path.join(HERE, "libgaviko_hip_diag.so")
# sources compiled into the diag library only (none at present: the experiment kernels of rounds 2-3 were measured slower and deleted,
# their numbers are in DESIGN.md 7b.1 / 7b.5 / 7c.5b)
DIAG_ONLY = set()
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]
# Per-file additions.  gemm_bf16: keep the MFMA accumulators in the VGPR half of gfx950's unified register file -- with the
# default AGPR form the software-pipelined main loop came out with ~100 v_accvgpr_read/write/mov copies per k-tile.
# attention_*: the score accumulator is re-used in place as the next MFMA's B operand, which the AGPR form can only do through
# v_accvgpr_read copies (352 / 688 of them in the forward / backward kernels).
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
FILE_FLAGS = {"gemm_bf16.hip": _VGPR_FORM, "gemm8p_bf16.hip": _VGPR_FORM, "attention_fwd.hip": _VGPR_FORM, "attention_bwd.hip": _VGPR_FORM, "skinny.hip": _VGPR_FORM}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _newer(src_list, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    """diag=False: the product library (no A/B switches, no experiment kernels).  diag=True: libgaviko_hip_diag.so = every source with
    -DGVK_DIAG, for tools/ (GAVIKO_HIP_DIAG=1)."""
    srcs = sorted(s for s in glob.glob(os.path.join(CSRC, "*.hip")) if diag or os.path.basename(s) not in DIAG_ONLY)
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(HERE, "..", "include", "gaviko_hip.h"),
                                                              os.path.join(HERE, "..", "include", "gaviko_hip_diag.h")]
    obj_dir = OBJ + ("_diag" if diag else "")
    lib_path = LIB_DIAG if diag else LIB
    extra = ["-DGVK_DIAG=1"] if diag else []
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()
    jobs = []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o")
        if force or _newer([s] + hdrs + [os.path.abspath(__file__)], o):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [cc, *FLAGS, *extra, *FILE_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s, rc, out in ex.map(compile_one, jobs):
                if verbose or rc:
                    print(f"[hipcc] {os.path.basename(s)} rc={rc}", file=sys.stderr)
                if out.strip() and (verbose or rc):
                    print(out, file=sys.stderr)
                if rc:
                    raise RuntimeError(f"hipcc failed on {s}")
    objs = [os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _newer(objs, lib_path):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError(f"link of {os.path.basename(lib_path)} failed")
        # a kernel template whose host stub was silently dropped by the compiler shows up only as an undefined symbol at load time:
        # resolve every symbol now (fresh interpreter, RTLD_NOW) so that such a build fails HERE and not on the GPU box
        r = subprocess.run([sys.executable, "-c", f"import ctypes, os; ctypes.CDLL({lib_path!r}, mode=os.RTLD_NOW)"], capture_output=True, text=True)
        if r.returncode:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError(f"{os.path.basename(lib_path)} does not load (undefined symbol?)")
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv))
