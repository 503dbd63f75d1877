"""Native task server: control plane, parameter store, accumulators (fake cluster = localhost ports)."""
import struct
import threading
import time

import numpy as np
import pytest

from distributed_tensorflow_amd.runtime import native as rt


@pytest.fixture()
def server():
    s = rt.TaskServer(port=0)
    yield s
    s.stop()


def _client(s):
    return rt.StoreClient("127.0.0.1", s.port, timeout=5)


def test_kv_counters_queues(server):
    a, b = _client(server), _client(server)
    a.set("k", b"v1")
    assert b.get("k") == b"v1"
    big = bytes(range(256)) * 1024            # larger than the default receive buffer
    a.set("big", big)
    assert b.get("big") == big
    assert a.add("c", 3) == 3 and b.add("c", 4) == 7
    got = []
    t = threading.Thread(target=lambda: got.append(b.wait_ge("c", 10, timeout=5)))
    t.start()
    time.sleep(0.05)
    a.add("c", 3)
    t.join()
    assert got == [10]
    with pytest.raises(TimeoutError):
        b.wait_ge("c", 99, timeout=0.1)
    a.qpush("q", struct.pack("<q", 7), copies=3)
    assert a.qlen("q") == 3 and struct.unpack("<q", b.qpop("q"))[0] == 7 and a.qlen("q") == 2
    with pytest.raises(TimeoutError):
        a.qpop("empty", timeout=0.05)
    # one request feeds a queue per worker (the chief's token release); a waiting consumer is woken
    woken = []
    t = threading.Thread(target=lambda: woken.append(b.qpop("tok/2", timeout=5)))
    t.start()
    time.sleep(0.05)
    assert a.qpush_fanout("tok", struct.pack("<q", 41), fanout=3, copies=2) == 3
    t.join()
    assert woken == [struct.pack("<q", 41)]
    assert [a.qlen(f"tok/{w}") for w in range(4)] == [2, 2, 1, 0]
    # blocking get is released by a later set (event-driven wait-for-init)
    t0 = time.time()
    res = []
    t = threading.Thread(target=lambda: res.append(b.get("late", timeout=5)))
    t.start()
    time.sleep(0.1)
    a.set("late", b"x")
    t.join()
    assert res == [b"x"] and time.time() - t0 < 2
    a.close(), b.close()


def test_buffers_and_server_side_sgd(server):
    c = _client(server)
    assert c.buf_create("shard", 1000) is True and c.buf_create("shard", 1000) is False
    w = np.random.RandomState(0).randn(1000).astype(np.float32)
    c.buf_write("shard", w)
    g = np.ones(10, dtype=np.float32)
    c.buf_axpy("shard", -0.5, g, offset=5, locked=True)
    want = w.copy()
    want[5:15] -= 0.5
    np.testing.assert_allclose(c.buf_read("shard", 1000), want)
    assert c.buf_add_scalar("shard", 0, 1.0) == int(want[0] + 1)
    with pytest.raises(RuntimeError):
        c.buf_read("shard", 10, offset=995)
    with pytest.raises(RuntimeError):
        c.buf_axpy("missing", 1.0, g)
    c.close()


def test_async_pushes_all_apply(server):
    """N unlocked pushes from concurrent connections -> N applies (integer-valued, so exact)."""
    c0 = _client(server)
    c0.buf_create("p", 4096)
    workers, pushes = 4, 25

    def work():
        c = _client(server)
        g = np.ones(4096, dtype=np.float32)
        for _ in range(pushes):
            c.buf_axpy("p", -1.0, g, locked=True)
        c.close()

    ts = [threading.Thread(target=work) for _ in range(workers)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    np.testing.assert_array_equal(c0.buf_read("p", 4096), np.full(4096, -workers * pushes, dtype=np.float32))
    c0.close()


def test_accumulator_mean_stale_drop_and_step(server):
    c = _client(server)
    c.buf_create("var", 8)
    assert c.acc_apply("acc", np.full(8, 2.0, np.float32), local_step=0)
    with pytest.raises(TimeoutError):       # one of two gradients: the barrier does not open
        c.acc_take_apply("acc", "var", 2, lr=1.0, timeout=0.1)
    assert c.acc_apply("acc", np.full(8, 4.0, np.float32), local_step=0)
    assert c.acc_take_apply("acc", "var", 2, lr=0.5, timeout=1) == 2
    np.testing.assert_allclose(c.buf_read("var", 8), np.full(8, -1.5))     # -lr * mean(2, 4)
    count, step, dropped = c.acc_info("acc")
    assert (count, step, dropped) == (0, 1, 0)
    assert c.acc_apply("acc", np.ones(8, np.float32), local_step=0) is False   # stale: local_step < 1
    assert c.acc_info("acc")[2] == 1
    assert c.acc_apply("acc", np.ones(8, np.float32), local_step=1) is True
    c.close()


def test_crc32c_known_vectors():
    assert rt.crc32c(b"123456789") == 0xE3069283
    assert rt.crc32c(b"") == 0
    assert rt.crc32c(bytes(32)) == 0x8A9136AA


def test_store_group_gathers_objects_and_synchronises_tasks(server):
    """The rendezvous used by worker tasks that share no torch.distributed group (handles of peer-mapped buffers)."""
    from distributed_tensorflow_amd.parallel.xgmi import StoreGroup

    world, results, order = 3, {}, []

    def task(rank):
        g = StoreGroup(_client(server), "test", rank, world, timeout=10)
        for round_no in range(2):                       # rounds do not mix
            out = [None] * world
            g.all_gather_object(out, {"rank": rank, "round": round_no, "blob": bytes([rank]) * 64})
            results[(rank, round_no)] = out
        if rank == 0:
            time.sleep(0.3)
        order.append(("before", rank))
        g.barrier()
        order.append(("after", rank))
        g.barrier()

    threads = [threading.Thread(target=task, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(20)
        assert not t.is_alive()
    for rank in range(world):
        for round_no in range(2):
            got = results[(rank, round_no)]
            assert [o["rank"] for o in got] == [0, 1, 2] and all(o["round"] == round_no for o in got)
            assert got[2]["blob"] == b"\x02" * 64
    assert [kind for kind, _ in order[:world]] == ["before"] * world      # nobody passed before the slowest arrived
