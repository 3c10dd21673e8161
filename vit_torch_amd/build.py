"""Build libvitmi.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

`python -m vit_torch_amd.build` or `__graft_entry__.build()`.  hipcc
cross-compiles without a GPU; the resulting .so travels with the repo snapshot
to the GPU box.  Objects are cached under vit_torch_amd/csrc/_obj by source
mtime so rebuilds only touch what changed.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = CSRC / "_obj"
LIB = HERE / "libvitmi.so"
COMM_LIB = HERE / "libvitmi_comm.so"      # the RCCL gradient exchange (include/vitmi_comm.h): host code only, RCCL bound at run time
COMM_SRC = "comm.cpp"
ARCH = "gfx950"

SOURCES = [
    "core.cpp",
    "gemm.hip",
    "gemm_fast.hip",
    "gemm_fast2.hip",
    "gemm_small.hip",
    "layernorm.hip",
    "elementwise.hip",
    "split3.hip",
    "ingest.hip",
    "posembed.hip",
    "optim.hip",
    "attention.hip",
    "attention_f32.hip",
    "cait_ops.hip",
    "cait_fused.hip",
    "swin_ops.hip",
]
HEADERS = ["common.h", "epilogue.h", "gemm_tile.h", "../../include/vitmi.h"]

FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-gpu-rdc",
    "-Wno-unused-command-line-argument",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP kernels cannot be built")
    return exe


def _newest_header() -> float:
    return max((CSRC / h).resolve().stat().st_mtime for h in HEADERS)


def _compile_one(src: str, force: bool) -> Path:
    s = CSRC / src
    o = OBJ / (src.rsplit(".", 1)[0] + ".o")
    stamp = max(s.stat().st_mtime, _newest_header(), Path(__file__).stat().st_mtime)
    if not force and o.exists() and o.stat().st_mtime >= stamp:
        return o
    extra = os.environ.get("VITMI_EXTRA_FLAGS", "").split()      # e.g. -DVITMI_GEMM_PHASE_STAMPS for tools/gemm_phases.py
    remarks = ["-Rpass-analysis=kernel-resource-usage"] if src.endswith(".hip") else []
    cmd = [_hipcc(), *FLAGS, *extra, *remarks, "-x", "hip", "-c", str(s), "-o", str(o)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    if remarks:
        _save_resources(src, r.stderr)
    return o


def _save_resources(src: str, stderr: str) -> None:
    """Per-kernel register / scratch / LDS figures of one source, from hipcc's resource-usage remarks ->
    csrc/_obj/<src>.resources.json (tests/test_abi_cpu.py bounds the scratch of the GEMM kernels: a dynamically indexed
    accumulator array shows up there, not in any numerical test)."""
    import json
    import re
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m or ":" not in m.group(1):
            continue
        k, v = (t.strip() for t in m.group(1).split(":", 1))
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None:
            cur[k] = v
    (OBJ / (src + ".resources.json")).write_text(json.dumps(out, indent=0))


def kernel_resources(src: str) -> dict:
    """{mangled kernel name: {"VGPRs": "..", "ScratchSize [bytes/lane]": "..", ...}} of the last build of `src`."""
    import json
    return json.loads((OBJ / (src + ".resources.json")).read_text())


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, force), SOURCES))
    newest = max(o.stat().st_mtime for o in objs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest:
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc",
               "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    build_comm(force)
    if verbose:
        print(f"built {LIB} ({LIB.stat().st_size} bytes), {COMM_LIB} ({COMM_LIB.stat().st_size} bytes)")
    return LIB


def build_comm(force: bool = False) -> Path:
    """libvitmi_comm.so: one host source; links libamdhip64 (through hipcc) and libdl, NOT librccl — the RCCL entry points
    are bound with dlopen / dlsym from the copy the process has already loaded (vitmi_comm_load)."""
    src = CSRC / COMM_SRC
    hdr = (CSRC / "../../include/vitmi_comm.h").resolve()
    stamp = max(src.stat().st_mtime, hdr.stat().st_mtime, Path(__file__).stat().st_mtime)
    if not force and COMM_LIB.exists() and COMM_LIB.stat().st_mtime >= stamp:
        return COMM_LIB
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O2", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
           "-Wno-unused-command-line-argument", "-x", "hip", str(src), "-o", str(COMM_LIB), "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {COMM_SRC}:\n{r.stderr}")
    return COMM_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
