"""Toolchain and topology probe: ``python -m distributed_tensorflow_amd.utils.envinfo [--json]``.

Answers, before any training is started, the questions a cluster launch depends on: which
compiler and ROCm stack is installed, whether the in-tree libraries are built for the current
sources, how many GPUs this process sees and what they are, which pairs can reach each other's
memory directly (the one-sided HIP-IPC transport and RCCL's peer-to-peer paths need it), and
whether the cross-process IPC mode the host driver supports is selected.

Nothing here initialises a GPU context unless a GPU is visible; on a CPU-only machine the device
section is empty and everything else is still reported.
"""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys
from typing import Any, Dict, List


def _first_line(cmd: List[str]) -> str:
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return ""
    for line in out.splitlines():
        if line.strip():
            return line.strip()
    return ""


def _libraries() -> Dict[str, Any]:
    from .. import build

    info = {}
    for lib, srcs, flags in ((build.HIP_LIB, sorted(build.CSRC.glob("*.hip")) + sorted(build.CSRC.glob("*.h")),
                              build.HIP_FLAGS),
                             (build.RT_LIB, sorted(build.RTSRC.glob("*.cpp")) + sorted(build.RTSRC.glob("*.h")),
                              build.CXX_FLAGS)):
        stamp = build._digest(srcs, " ".join(flags))
        info[lib.name] = {"present": lib.exists(), "matches_sources": build._up_to_date(lib, stamp),
                          "bytes": lib.stat().st_size if lib.exists() else 0}
    return info


def collect() -> Dict[str, Any]:
    import torch

    info: Dict[str, Any] = {
        "python": sys.version.split()[0],
        "torch": torch.__version__,
        "torch_hip": getattr(torch.version, "hip", None),
        "hipcc": _first_line([shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "--version"]),
        "rocprofv3": shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None),
        "offload_arch": "gfx950",
        "distributed_backends": {"nccl (RCCL)": bool(torch.distributed.is_available() and torch.distributed.is_nccl_available()),
                                 "gloo": bool(torch.distributed.is_available() and torch.distributed.is_gloo_available())},
        "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                "GPU_MAX_HW_QUEUES", "NCCL_DEBUG", "DTF_TRACE", "DTF_HIP_LIB")},
        "libraries": _libraries(),
        "devices": [],
        "peer_access": [],
    }
    try:
        info["rccl"] = ".".join(map(str, torch.cuda.nccl.version())) if torch.cuda.is_available() else None
    except Exception:  # noqa: BLE001 — builds without the binding
        info["rccl"] = None
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    for i in range(n):
        p = torch.cuda.get_device_properties(i)
        info["devices"].append({"index": i, "name": p.name, "arch": getattr(p, "gcnArchName", ""),
                                "compute_units": p.multi_processor_count,
                                "memory_gib": round(p.total_memory / 2 ** 30, 1),
                                "wavefront": getattr(p, "warp_size", 64)})
    for i in range(n):
        row = []
        for j in range(n):
            row.append(True if i == j else bool(torch.cuda.can_device_access_peer(i, j)))
        info["peer_access"].append(row)
    return info


def warnings_for(info: Dict[str, Any]) -> List[str]:
    out = []
    for name, lib in info["libraries"].items():
        if not lib["present"]:
            out.append(f"{name} is not built: run python -m distributed_tensorflow_amd.build")
        elif not lib["matches_sources"]:
            out.append(f"{name} is older than its sources: run python -m distributed_tensorflow_amd.build")
    if info["devices"]:
        if info["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] != "0":
            out.append("HSA_ENABLE_IPC_MODE_LEGACY is not 0: cross-process GPU memory sharing (transport=ipc, RCCL "
                       "between processes) needs the dmabuf IPC mode on hosts whose driver only supports that one")
        for d in info["devices"]:
            if d["arch"] and not d["arch"].startswith("gfx950"):
                out.append(f"device {d['index']} is {d['arch']}: the kernels are built for gfx950 only")
        for i, row in enumerate(info["peer_access"]):
            for j, ok in enumerate(row):
                if not ok:
                    out.append(f"device {i} cannot access device {j} directly: shards on {j} are reached through the host")
    return out


def format_report(info: Dict[str, Any]) -> str:
    lines = [f"python {info['python']}  torch {info['torch']}  (HIP {info['torch_hip']})  RCCL {info['rccl']}",
             f"hipcc: {info['hipcc'] or 'not found'}", f"rocprofv3: {info['rocprofv3'] or 'not found'}",
             "backends: " + ", ".join(f"{k}={'yes' if v else 'no'}" for k, v in info["distributed_backends"].items())]
    for name, lib in info["libraries"].items():
        state = "missing" if not lib["present"] else ("up to date" if lib["matches_sources"] else "STALE")
        lines.append(f"{name}: {state} ({lib['bytes']} bytes)")
    lines.append("env: " + ", ".join(f"{k}={v}" for k, v in info["env"].items() if v is not None))
    if not info["devices"]:
        lines.append("devices: none visible (CPU execution; kernels cross-compile for gfx950)")
    for d in info["devices"]:
        lines.append(f"gpu:{d['index']} {d['name']} {d['arch']}  {d['compute_units']} CUs  {d['memory_gib']} GiB  "
                     f"wave{d['wavefront']}")
    if len(info["devices"]) > 1:
        lines.append("peer access (row can read column):")
        for i, row in enumerate(info["peer_access"]):
            lines.append(f"  gpu:{i}  " + " ".join("x" if ok else "." for ok in row))
    for w in warnings_for(info):
        lines.append("WARNING: " + w)
    return "\n".join(lines)


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    info = collect()
    if "--json" in argv:
        info["warnings"] = warnings_for(info)
        print(json.dumps(info))
    else:
        print(format_report(info))
    return 0


if __name__ == "__main__":
    sys.exit(main())
