import pytest

import importlib

F = importlib.import_module("distributed_tensorflow_amd.flags")  # the package attribute `flags` is the tf.app.flags facade
from distributed_tensorflow_amd.cluster import ClusterSpec, ConfigProto, Server


def _fresh():
    fv = F._FlagValues()
    fv._define("job_name", str, "", "")
    fv._define("task_index", int, 0, "")
    fv._define("resident", F._parse_bool, False, "")
    fv._define("lr", float, 0.001, "")
    return fv


def test_flag_forms_and_defaults():
    fv = _fresh()
    fv._parse(["--job_name=worker", "--task_index", "2", "--resident", "--lr=0.5", "extra", "--unknown=1"])
    assert (fv.job_name, fv.task_index, fv.resident, fv.lr) == ("worker", 2, True, 0.5)
    assert fv.unparsed == ["extra", "--unknown=1"]
    fv2 = _fresh()
    fv2._parse([])
    assert (fv2.job_name, fv2.task_index, fv2.resident) == ("", 0, False)
    fv3 = _fresh()
    fv3._parse(["--noresident", "--task-index=4"])
    assert fv3.resident is False and fv3.task_index == 4
    with pytest.raises(AttributeError):
        fv3.nope


def test_lazy_parse_from_argv(monkeypatch):
    fv = _fresh()
    monkeypatch.setattr("sys.argv", ["prog", "--job_name=ps"])
    assert fv.job_name == "ps"      # first attribute access parses sys.argv


def test_help_lists_every_flag_and_exits(capsys):
    fv = _fresh()
    fv._define("job_name", str, "", "either 'ps' or 'worker'")
    with pytest.raises(SystemExit) as e:
        fv._parse(["--job_name=ps", "--help"])
    assert e.value.code == 0
    out = capsys.readouterr().out
    assert "--job_name=<string>: either 'ps' or 'worker'" in out and "--[no]resident" in out
    assert "--task_index=<integer>" in out and "(default: 0.001)" in out


def test_enum_flag_rejects_other_values(monkeypatch):
    monkeypatch.setattr(F, "FLAGS", F._FlagValues())
    F.DEFINE_enum("transport", "tcp", ["tcp", "ipc"], "parameter-server transport")
    F.FLAGS._parse(["--transport=ipc"])
    assert F.FLAGS.transport == "ipc"
    with pytest.raises(ValueError, match="one of"):
        F.FLAGS._parse(["--transport", "carrier-pigeon"])


def test_cluster_spec():
    c = ClusterSpec({"ps": ["a:1"], "worker": ["a:2", "b:3"]})
    assert c.num_tasks("worker") == 2 and c.task_address("worker", 1) == "b:3"
    assert c.as_dict() == {"ps": ["a:1"], "worker": ["a:2", "b:3"]} and ClusterSpec(c) == c
    with pytest.raises(ValueError):
        c.task_address("worker", 5)
    assert ClusterSpec({"worker": {1: "x:1", 0: "y:2"}}).job_tasks("worker") == ["y:2", "x:1"]


def test_server_roles_and_bad_job():
    from cluster_utils import free_ports

    port, wport = free_ports(2)           # two ports known to be free (port + 1 may belong to somebody else)
    cluster = {"ps": [f"localhost:{port}"], "worker": [f"localhost:{wport}"]}
    with pytest.raises(ValueError):
        Server(cluster, job_name="", task_index=0)
    srv = Server(cluster, job_name="ps", task_index=0, config=ConfigProto(allow_soft_placement=True))
    try:
        assert srv.target == f"dtf://localhost:{port}" and srv.num_ps == 1 and srv.num_workers == 1
        assert not srv.is_chief and srv.join(timeout=0.05) is False
        w = Server(cluster, job_name="worker", task_index=0)
        try:
            assert w.is_chief
            w.report_worker_done()            # the only worker -> PS is released from join()
            assert srv.join(timeout=2.0) is True
        finally:
            w.stop()
    finally:
        srv.stop()
