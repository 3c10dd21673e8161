"""ctypes binding of libvitmi_comm.so (include/vitmi_comm.h): the gradient exchange on the library's OWN RCCL
communicator — one `ncclComm_t` per process, one HIP stream beside the compute stream, two events, no helper thread.

torch.distributed is used for ONE thing here: handing rank 0's 128-byte `ncclUniqueId` to the other ranks
(`broadcast_object_list`, any backend).  After that the buckets, the parameter broadcast and the epoch's metric
all-reduce go through `ncclAllReduce` / `ncclBroadcast` of the RCCL copy the process has already loaded (PyTorch's:
`torch/lib/librccl.so`), bound with dlopen — no second RCCL in the process.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch

from ._lib import VitmiError

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libvitmi_comm.so"
UNIQUE_ID_BYTES = 128

c_vp, c_i64 = C.c_void_p, C.c_int64
SIGNATURES = {
    "vitmi_comm_version": (C.c_int, []),
    "vitmi_comm_last_error": (C.c_char_p, []),
    "vitmi_comm_load": (C.c_int, [C.c_char_p]),
    "vitmi_comm_rccl_version": (C.c_int, [C.POINTER(C.c_int)]),
    "vitmi_comm_unique_id": (C.c_int, [c_vp]),
    "vitmi_comm_init": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.POINTER(c_vp)]),
    "vitmi_comm_info": (C.c_int, [c_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vitmi_comm_allreduce_sum_f32_async": (C.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "vitmi_comm_join": (C.c_int, [c_vp, c_vp]),
    "vitmi_comm_broadcast_f32": (C.c_int, [c_vp, c_vp, c_i64, C.c_int, c_vp]),
    "vitmi_comm_allreduce_sum_f32": (C.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "vitmi_comm_destroy": (C.c_int, [c_vp]),
}

_lib = None


def rccl_path() -> Optional[str]:
    """The librccl the process uses: PyTorch's bundled copy if there is one (it is what `backend="nccl"` has loaded),
    else None (= by soname, the ROCm installation's)."""
    p = Path(torch.__file__).resolve().parent / "lib" / "librccl.so"
    return str(p) if p.exists() else None


def load() -> C.CDLL:
    """libvitmi_comm.so with RCCL bound; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise VitmiError(f"{LIB_PATH} is missing: build the extension first (python -m vit_torch_amd.build)")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    p = os.environ.get("VITMI_RCCL_PATH") or rccl_path()
    rc = lib.vitmi_comm_load(p.encode() if p else None)
    if rc != 0:
        raise VitmiError(f"vitmi_comm_load failed (rc={rc}): {lib.vitmi_comm_last_error().decode()}")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().vitmi_comm_last_error()
        raise VitmiError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def rccl_version() -> int:
    v = C.c_int(0)
    _check(load().vitmi_comm_rccl_version(C.byref(v)), "vitmi_comm_rccl_version")
    return int(v.value)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class RcclComm:
    """One RCCL communicator over the ranks of `group` (default: the world), on `device`."""

    def __init__(self, group=None, device: Optional[int] = None):
        import torch.distributed as dist
        lib = load()
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            world, rank = 1, 0
        self.device = torch.cuda.current_device() if device is None else int(device)
        uid = C.create_string_buffer(UNIQUE_ID_BYTES)
        if rank == 0:
            _check(lib.vitmi_comm_unique_id(uid), "vitmi_comm_unique_id")
        if world > 1:
            box = [uid.raw if rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            uid = C.create_string_buffer(box[0], UNIQUE_ID_BYTES)
        h = c_vp()
        _check(lib.vitmi_comm_init(uid, world, rank, self.device, C.byref(h)), "vitmi_comm_init")
        self._h = h
        self.world, self.rank = world, rank

    def info(self) -> dict:
        """What RCCL itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice, ncclGetVersion)."""
        w, r, d = C.c_int(), C.c_int(), C.c_int()
        _check(load().vitmi_comm_info(self._h, C.byref(w), C.byref(r), C.byref(d)), "vitmi_comm_info")
        return {"rccl_version": rccl_version(), "comm_ranks": int(w.value), "comm_rank": int(r.value),
                "comm_device": int(d.value)}

    def _buf(self, t: torch.Tensor):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device.index == self.device):
            raise VitmiError("RcclComm: buffers must be contiguous fp32 tensors on the communicator's device")
        return t.data_ptr(), t.numel()

    def allreduce_async(self, t: torch.Tensor) -> None:
        p, n = self._buf(t)
        _check(load().vitmi_comm_allreduce_sum_f32_async(self._h, p, n, _stream()), "vitmi_comm_allreduce_sum_f32_async")

    def join(self) -> None:
        _check(load().vitmi_comm_join(self._h, _stream()), "vitmi_comm_join")

    def broadcast(self, t: torch.Tensor, root: int = 0) -> None:
        p, n = self._buf(t)
        _check(load().vitmi_comm_broadcast_f32(self._h, p, n, root, _stream()), "vitmi_comm_broadcast_f32")

    def allreduce(self, t: torch.Tensor) -> None:
        p, n = self._buf(t)
        _check(load().vitmi_comm_allreduce_sum_f32(self._h, p, n, _stream()), "vitmi_comm_allreduce_sum_f32")

    def destroy(self) -> None:
        if self._h is not None:
            h, self._h = self._h, None
            _check(load().vitmi_comm_destroy(h), "vitmi_comm_destroy")


_default = {}


def default_comm(group=None) -> RcclComm:
    """THE communicator of this process for `group` (created on first use, every rank at the same point: creation is a
    collective), SURVEY §8(b): "ncclComm_t created once per process"."""
    key = id(group) if group is not None else None
    c = _default.get(key)
    if c is None or c._h is None:
        c = _default[key] = RcclComm(group)
    return c


def shutdown() -> None:
    """Destroy the process's communicators (call before torch.distributed.destroy_process_group / at the end of a job)."""
    for c in list(_default.values()):
        c.destroy()
    _default.clear()
