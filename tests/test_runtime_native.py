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
    # an update that is counted in the same request: the buffer's step counter (element 999 here) goes up by one per
    # call and its new value comes back (an asynchronous worker's push over tcp: one round trip, not two)
    c.buf_write("shard", np.array([0, 1, 2, 3], dtype=np.float32), offset=996)
    assert c.buf_axpy("shard", 2.0, np.ones(3, np.float32), offset=996, bump=999) == 4
    assert c.buf_axpy("shard", 2.0, np.ones(3, np.float32), offset=996, locked=True, bump=999) == 5
    assert list(c.buf_read("shard", 4, offset=996)) == [4, 5, 6, 5]
    with pytest.raises(RuntimeError):
        c.buf_axpy("shard", 1.0, g, bump=1000)                   # a counter outside the buffer
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


def test_push_and_wait_in_one_round_trip_and_bulk_pop(server):
    """QPUSH_POP (a replica's "gradient is there, wake me with my token") and QPOP_N (the aggregator's arrivals)."""
    worker, chief = _client(server), _client(server)
    got = []
    t = threading.Thread(target=lambda: got.append(worker.qpush_pop("arrivals", b"w0-step7", "tokens/0", timeout=5)))
    t.start()
    n, recs = chief.qpop_n("arrivals", 1, timeout=5)
    assert (n, recs) == (1, b"w0-step7") and not got          # the worker is still blocked on its token
    chief.qpush("tokens/0", struct.pack("<q", 8))
    t.join()
    assert struct.unpack("<q", got[0])[0] == 8
    with pytest.raises(TimeoutError):
        worker.qpush_pop("arrivals", b"x", "tokens/0", timeout=0.1)
    for i in range(3):
        chief.qpush("arrivals", bytes([i]) * 4)
    n, recs = chief.qpop_n("arrivals", 8, timeout=0.2)        # fewer than asked for when the time is up
    assert n == 4 and recs == b"x" + b"\x00" * 4 + b"\x01" * 4 + b"\x02" * 4
    with pytest.raises(TimeoutError):
        chief.qpop_n("arrivals", 1, timeout=0.05)


def test_chief_round_in_one_request_and_push_that_waits_for_its_token(server):
    """ACC_TAKE_APPLY with the release flag = take_grad + apply + step bump + token deal; ACC_APPLY with the
    then-pop flag = a synchronous replica's push and token wait."""
    chief = _client(server)
    workers = [_client(server) for _ in range(2)]
    chief.buf_create("var", 5)
    chief.buf_write("var", np.array([10.0, 10.0, 10.0, 10.0, 3.0], dtype=np.float32))     # element 4: the step
    chief.acc_set_step("acc", 3)
    results = [None, None]

    def push(i):
        g = np.array([1.0, 2.0, 3.0, 4.0, 0.0], dtype=np.float32) * (i + 1)
        results[i] = workers[i].acc_apply_then_pop("acc", g, 3, f"tok/{i}", timeout=5)

    threads = [threading.Thread(target=push, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    step = chief.acc_take_apply_release("acc", "var", 2, 0.5, 4, "tok", 2, copies=1, timeout=5)
    for t in threads:
        t.join()
    assert step == 4
    assert all(ok and struct.unpack("<q", tok)[0] == 4 for ok, tok in results)
    got = chief.buf_read("var", 5)
    np.testing.assert_allclose(got, [10 - 0.5 * 1.5, 10 - 0.5 * 3.0, 10 - 0.5 * 4.5, 10 - 0.5 * 6.0, 4.0])
    # a stale push is dropped but still waits for (and gets) a token
    chief.qpush("tok/0", struct.pack("<q", 9))
    ok, tok = workers[0].acc_apply_then_pop("acc", np.zeros(5, dtype=np.float32), 2, "tok/0", timeout=5)
    assert ok is False and struct.unpack("<q", tok)[0] == 9


def test_barrier_board_in_shared_memory_passes_arrivals_and_tokens_between_mappings():
    """runtime/shm_board.cpp through its binding: the chief's mapping and a second mapping of the same board."""
    import threading

    from distributed_tensorflow_amd.runtime import native as rt

    name = rt.Board.unique_name("test")
    chief = rt.Board(name, 3, create=True)
    with pytest.raises(RuntimeError):
        rt.Board(name, 3, create=True)                  # exclusive create
    with pytest.raises(RuntimeError):
        rt.Board(name, 4, create=False)                 # the worker count is part of the contract
    other = rt.Board(name, 3, create=False)
    assert chief.attached() == 2
    with pytest.raises(TimeoutError):
        chief.take(3, 0.01)
    with pytest.raises(TimeoutError):
        other.wait_token(1, 0.01)
    got = {}

    def replica(w):
        b = other if w else chief
        b.arrive(w, 0, w & 1)
        got[w] = b.wait_token(w, 10.0)

    threads = [threading.Thread(target=replica, args=(w,)) for w in range(3)]
    for t in threads:
        t.start()
    arrivals = []
    while len(arrivals) < 3:
        arrivals += chief.take(3 - len(arrivals), 10.0)
    assert sorted(arrivals) == [(0, 0, 0), (1, 0, 1), (2, 0, 0)]
    chief.release(-1, 1, 1)                             # one token each, carrying the new step
    for t in threads:
        t.join(10)
    assert got == {0: 1, 1: 1, 2: 1}
    chief.release(2, 7, 2)                              # two tokens for one worker only
    assert other.wait_token(2, 1.0) == 7 and other.wait_token(2, 1.0) == 7
    with pytest.raises(TimeoutError):
        other.wait_token(0, 0.01)
    other.close()
    chief.close()                                       # the creator removes the name
    with pytest.raises(RuntimeError):
        rt.Board(name, 3, create=False)
