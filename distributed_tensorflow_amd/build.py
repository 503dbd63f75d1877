"""In-tree build of the native libraries.

Two shared objects are produced under ``distributed_tensorflow_amd/_lib``:

* ``libdtf_hip.so`` — the gfx950 HIP kernels (``csrc/*.hip``), compiled with
  ``hipcc --offload-arch=gfx950``.  hipcc cross-compiles, so this works on a
  machine without a GPU.
* ``libdtf_rt.so`` — the host-side C++ runtime (``runtime/*.cpp``): control
  store / rendezvous server, TFRecord event writer, tensor-bundle checkpoint
  writer.  Plain g++, no GPU dependency, usable in CPU-only tests.

The libraries are loaded with ctypes (``ops/hip_backend.py``,
``runtime/native.py``); nothing is installed outside the repository.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
RTSRC = PKG / "runtime"
LIB = PKG / "_lib"

HIP_LIB = LIB / "libdtf_hip.so"
RT_LIB = LIB / "libdtf_rt.so"

HIP_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-fno-gpu-rdc",
    "-Wall",
    "-Wno-unused-function",
]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _digest(paths, extra: str) -> str:
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _up_to_date(lib: Path, stamp: str) -> bool:
    s = lib.with_suffix(".stamp")
    return lib.exists() and s.exists() and s.read_text() == stamp


def _run(cmd) -> None:
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        raise RuntimeError("build failed: " + " ".join(map(str, cmd)))
    if proc.stdout.strip() and os.environ.get("DTF_BUILD_VERBOSE"):
        print(proc.stdout)


def build_hip(force: bool = False) -> Path:
    LIB.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = sorted(CSRC.glob("*.h"))
    stamp = _digest(srcs + hdrs, " ".join(HIP_FLAGS))
    if not force and _up_to_date(HIP_LIB, stamp):
        return HIP_LIB
    tmp = HIP_LIB.with_suffix(".so.tmp")
    _run([_hipcc(), *HIP_FLAGS, f"-I{CSRC}", *map(str, srcs), "-o", str(tmp)])
    os.replace(tmp, HIP_LIB)
    HIP_LIB.with_suffix(".stamp").write_text(stamp)
    return HIP_LIB


def build_runtime(force: bool = False) -> Path:
    LIB.mkdir(exist_ok=True)
    srcs = sorted(RTSRC.glob("*.cpp"))
    hdrs = sorted(RTSRC.glob("*.h"))
    if not srcs:
        return None
    stamp = _digest(srcs + hdrs, " ".join(CXX_FLAGS))
    if not force and _up_to_date(RT_LIB, stamp):
        return RT_LIB
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    tmp = RT_LIB.with_suffix(".so.tmp")
    _run([cxx, *CXX_FLAGS, f"-I{RTSRC}", *map(str, srcs), "-o", str(tmp)])
    os.replace(tmp, RT_LIB)
    RT_LIB.with_suffix(".stamp").write_text(stamp)
    return RT_LIB


SANITIZERS = {
    "thread": ["-fsanitize=thread"],
    "address": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
}
STRESS_SOURCES = {
    "server_stress": ["stress/server_stress.cpp", "ps_server.cpp"],
    "records_stress": ["stress/records_stress.cpp", "records.cpp"],
    "board_stress": ["stress/board_stress.cpp", "shm_board.cpp"],
}


def _sanitizer_cxx() -> str:
    """clang++ of the ROCm toolchain: its sanitizer runtimes intercept the clock-based condition
    variable waits libstdc++ uses; g++ 11's ThreadSanitizer does not and reports locks it never saw
    released."""
    for cand in (os.environ.get("DTF_SANITIZER_CXX"), "/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("no clang++ found for the sanitizer builds (set DTF_SANITIZER_CXX)")


def build_stress(name: str, sanitizer: str, out_dir: Path) -> Path:
    """Host-only sanitizer build of one stress program of the native runtime (``runtime/stress``)."""
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    exe = out_dir / f"{name}_{sanitizer}"
    srcs = [str(RTSRC / rel) for rel in STRESS_SOURCES[name]]
    _run([_sanitizer_cxx(), "-std=c++17", "-O1", "-g", "-pthread", *SANITIZERS[sanitizer], f"-I{RTSRC}", *srcs,
          "-o", str(exe)])
    return exe


def build_all(force: bool = False):
    return [lib for lib in (build_hip(force), build_runtime(force)) if lib is not None]


if __name__ == "__main__":
    force = "--force" in sys.argv
    for lib in build_all(force):
        print("built", lib)
