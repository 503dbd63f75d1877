"""No kernel of the library may use scratch (private) memory.

A kernel whose code object asks for a private segment starts about a microsecond later than one that does not
(the wave's scratch has to be set up), which is 10 % of a whole training step here; it creeps in silently, e.g.
through a small local array indexed by a loop variable.  The test compiles every HIP source to gfx950 assembly
(no GPU needed) and reads the kernel descriptors.
"""
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parent.parent / "distributed_tensorflow_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _descriptors(src: Path, out: Path):
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", f"-I{CSRC}",
                    "--cuda-device-only", "-S", str(src), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    text = out.read_text()
    return re.findall(r"\.amdhsa_kernel (\S+).*?\.amdhsa_private_segment_fixed_size (\d+)", text, flags=re.S)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
@pytest.mark.timeout(900)
def test_no_kernel_asks_for_a_private_segment(tmp_path):
    sources = sorted(CSRC.glob("*.hip"))
    assert sources
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(lambda s: _descriptors(s, tmp_path / (s.stem + ".s")), sources))
    kernels = [k for r in results for k in r]
    assert len(kernels) >= 10
    offenders = {name: int(size) for name, size in kernels if int(size) != 0}
    assert not offenders, offenders
