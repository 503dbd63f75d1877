"""Host tracing: spans, collectors, the RunOptions/RunMetadata/Timeline surface, profiler markers."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.client import timeline
from distributed_tensorflow_amd.examples.mnist_mlp import build_model
from distributed_tensorflow_amd.utils import trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _tracing_off():
    trace.disable()
    yield
    trace.disable()


def test_span_is_a_shared_noop_when_tracing_is_off():
    a, b = trace.span("pull"), trace.span("push", bytes=4)
    assert a is b and not trace.enabled()
    with a as s:
        s.set(step=3)       # accepted and ignored


def test_collector_records_nested_spans_and_marks_per_thread():
    with trace.Collector(process_name="worker 0") as col:
        with trace.span("outer", step=7):
            with trace.span("inner") as s:
                s.set(bytes=318040)
            trace.mark("stale_gradient", local_step=3)
        seen = []
        t = threading.Thread(target=lambda: seen.append(trace.span("elsewhere")))
        t.start()
        t.join()
    assert seen[0] is trace.span("x")                       # the collector is scoped to its own thread
    names = [e[0] for e in col.events]
    assert names == ["inner", "stale_gradient", "outer"]     # spans land when they close
    inner, _, outer = col.events
    assert outer[1] <= inner[1] and inner[1] + inner[2] <= outer[1] + outer[2]
    tot = col.totals()
    assert tot["outer"]["count"] == 1 and "stale_gradient" not in tot
    assert "outer" in col.format_totals()
    ct = col.chrome_trace(pid=1)
    evs = {e["name"]: e for e in ct["traceEvents"]}
    assert evs["process_name"]["args"]["name"] == "worker 0"
    assert evs["inner"]["ph"] == "X" and evs["inner"]["args"] == {"bytes": 318040}
    assert evs["stale_gradient"]["ph"] == "i" and evs["outer"]["args"] == {"step": 7}
    json.dumps(ct)


def test_collector_capacity_counts_what_it_drops():
    with trace.Collector(capacity=2) as col:
        for _ in range(5):
            with trace.span("s"):
                pass
    assert len(col.events) == 2 and col.dropped == 3
    assert col.chrome_trace()["otherData"]["dropped"] == 3


def test_session_run_fills_run_metadata_like_tf1(tmp_path):
    rng = np.random.RandomState(0)
    x = rng.rand(20, 784).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 20)]
    with tf.Graph().as_default():
        m = build_model(0.01)
        with tf.Session() as sess:
            sess.run(m.init_op)
            plain = sess.run([m.train_op, m.cross_entropy], feed_dict={m.x: x, m.y_: y})[1]
            md = tf.RunMetadata()
            opts = tf.RunOptions(trace_level=tf.RunOptions.FULL_TRACE)
            traced = sess.run([m.train_op, m.cross_entropy], feed_dict={m.x: x, m.y_: y}, options=opts, run_metadata=md)[1]
            assert traced < plain                                  # the traced call trained as usual
            names = [e[0] for e in md.step_stats.events]
            # (a local session applies the update inside "step"; "pull" / "push_apply" appear with parameter servers)
            assert names[-1] == "session.run" and {"stage", "step", "fetch_stats"} <= set(names)
            # a second traced run appends to the same metadata object
            sess.run(m.train_op, feed_dict={m.x: x, m.y_: y}, options=opts, run_metadata=md)
            assert [e[0] for e in md.step_stats.events].count("session.run") == 2
            assert not trace.enabled()                             # tracing ended with the call
            # no options -> nothing recorded
            before = len(md.step_stats.events)
            sess.run(m.train_op, feed_dict={m.x: x, m.y_: y})
            assert len(md.step_stats.events) == before
    text = timeline.Timeline(md.step_stats).generate_chrome_trace_format()
    path = tmp_path / "timeline.json"
    path.write_text(text)
    doc = json.loads(path.read_text())
    assert sum(1 for e in doc["traceEvents"] if e["name"] == "step") == 2


def test_saver_and_trainers_emit_spans(tmp_path):
    with tf.Graph().as_default():
        m = build_model(0.01)
        with tf.Session() as sess:
            sess.run(m.init_op)
            saver = tf.train.Saver()
            with trace.Collector() as col:
                prefix = saver.save(sess, str(tmp_path / "model.ckpt"), global_step=m.global_step)
                saver.restore(sess, prefix)
    assert [e[0] for e in col.events] == ["checkpoint_save", "checkpoint_restore"]


def test_profiler_markers_can_be_switched_on_without_a_profiler_attached():
    try:
        trace.enable(roctx=True)
    except RuntimeError:
        pytest.skip("marker library not present in this image")
    assert trace.enabled()
    with trace.span("pull"):
        with trace.span("step"):
            trace.mark("here")
    trace.disable()
    assert trace.span("x") is trace.span("y")


def test_env_switch_writes_a_timeline_at_exit(tmp_path):
    out = tmp_path / "trace_%p.json"
    code = ("from distributed_tensorflow_amd.utils import trace\n"
            "import os\n"
            "with trace.span('pull', shard=0):\n    pass\n"
            "print(os.getpid())\n")
    env = dict(os.environ, DTF_TRACE="timeline", DTF_TRACE_FILE=str(out), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    pid = int(r.stdout.strip().splitlines()[-1])
    doc = json.loads((tmp_path / f"trace_{pid}.json").read_text())
    assert any(e["name"] == "pull" and e.get("args") == {"shard": 0} for e in doc["traceEvents"])
