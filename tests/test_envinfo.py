"""Toolchain / topology probe."""
import json
import subprocess
import sys

import pytest
import torch

from distributed_tensorflow_amd.utils import envinfo


def test_probe_reports_toolchain_and_library_state_on_any_machine():
    info = envinfo.collect()
    assert info["torch"] == torch.__version__ and info["offload_arch"] == "gfx950"
    assert set(info["libraries"]) == {"libdtf_hip.so", "libdtf_rt.so"}
    for lib in info["libraries"].values():
        assert lib["present"] and lib["matches_sources"], "run python -m distributed_tensorflow_amd.build"
    assert info["distributed_backends"]["gloo"] is True
    text = envinfo.format_report(info)
    assert "libdtf_hip.so: up to date" in text
    assert len(info["peer_access"]) == len(info["devices"])


def test_probe_warns_about_what_breaks_a_cluster_launch():
    info = {"libraries": {"libdtf_hip.so": {"present": True, "matches_sources": False, "bytes": 1},
                          "libdtf_rt.so": {"present": False, "matches_sources": False, "bytes": 0}},
            "devices": [{"index": 0, "arch": "gfx950:sramecc+:xnack-"}, {"index": 1, "arch": "gfx942"}],
            "peer_access": [[True, False], [True, True]],
            "env": {"HSA_ENABLE_IPC_MODE_LEGACY": None}}
    w = "\n".join(envinfo.warnings_for(info))
    assert "libdtf_hip.so is older than its sources" in w and "libdtf_rt.so is not built" in w
    assert "HSA_ENABLE_IPC_MODE_LEGACY is not 0" in w
    assert "device 1 is gfx942" in w and "device 0 is" not in w
    assert "device 0 cannot access device 1" in w and "device 1 cannot access device 0" not in w


def test_probe_cli_emits_json():
    r = subprocess.run([sys.executable, "-m", "distributed_tensorflow_amd.utils.envinfo", "--json"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    doc = json.loads(r.stdout.strip().splitlines()[-1])
    assert "warnings" in doc and "hipcc" in doc


@pytest.mark.gpu
def test_probe_sees_the_gfx950_device():
    info = envinfo.collect()
    assert info["devices"], "no GPU visible"
    d = info["devices"][0]
    assert d["arch"].startswith("gfx950") and d["compute_units"] >= 200 and d["memory_gib"] > 200
    assert info["peer_access"][0][0] is True
    assert not any("built for gfx950 only" in w for w in envinfo.warnings_for(info))
