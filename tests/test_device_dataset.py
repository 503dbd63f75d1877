"""DeviceDataset on the CPU: epoch reshuffles, staged permutations, Gram blocks of consecutive mini-batches."""
import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.data import DeviceDataset


def _data(n=1000, k=32, c=10, seed=0):
    rng = np.random.RandomState(seed)
    return rng.rand(n, k).astype(np.float32), np.eye(c, dtype=np.float32)[rng.randint(0, c, n)]


def test_every_epoch_is_a_permutation_of_the_split_and_rows_keep_their_labels():
    x, y = _data()
    ds = DeviceDataset(x, y, "cpu", 100, seed=5)
    key = {tuple(np.round(r, 6)): tuple(l) for r, l in zip(x, y)}
    seen = []
    for _ in range(3):
        xe, ye = ds.x_epoch.numpy(), ds.y_epoch.numpy()
        assert sorted(map(tuple, np.round(xe, 6))) == sorted(key)           # same rows, other order
        assert all(key[tuple(np.round(r, 6))] == tuple(l) for r, l in zip(xe, ye))
        seen.append(xe.copy())
        ds.shuffle_epoch()
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])
    assert ds.epoch == 3 and ds.batches_per_epoch == 10


def test_same_seed_same_orders_and_workers_differ():
    x, y = _data()
    a, b, c = (DeviceDataset(x, y, "cpu", 100, seed=s) for s in (7, 7, 8))
    for _ in range(2):
        assert torch.equal(a.x_epoch, b.x_epoch) and not torch.equal(a.x_epoch, c.x_epoch)
        for d in (a, b, c):
            d.shuffle_epoch()


def test_pair_gram_is_one_plus_the_dot_products_of_consecutive_batches_and_follows_reshuffles():
    x, y = _data(n=530, k=48)          # 5 batches of 100 (30 rows never used), ld = 112
    ds = DeviceDataset(x, y, "cpu", 100, seed=1)
    g = ds.enable_pair_gram()
    assert g.shape == (4, 100, 112)
    for _ in range(2):
        xb = ds.x_epoch[:500].view(5, 100, -1)
        for j in range(4):
            torch.testing.assert_close(ds.gram[j, :, :100], 1.0 + xb[j + 1] @ xb[j].t(), rtol=1e-5, atol=1e-5)
        assert float(ds.gram[:, :, 100:].abs().max()) == 0.0
        ds.shuffle_epoch()
    assert ds.gram.data_ptr() == g.data_ptr()          # refreshed in place: captured graphs stay valid


def test_double_buffer_is_ignored_on_the_cpu():
    x, y = _data()
    ds = DeviceDataset(x, y, "cpu", 100, seed=1, double_buffer=True)
    assert not ds.double_buffer and ds.active == 0 and ds.set_tensors(0)[0] is ds.x_epoch


def test_launches_of_a_set_go_to_its_own_device_whatever_device_is_current(monkeypatch):
    """A single client that drives towers on several GPUs calls `shuffle_epoch()` with some other device current: the
    gathers and the permutation upload must still be issued with the set's device current (their stream is resolved from
    the current device).  Checked here by recording which device the set makes current around its work."""
    import contextlib

    import numpy as np
    import torch

    from distributed_tensorflow_amd.data import DeviceDataset

    rng = np.random.RandomState(0)
    ds = DeviceDataset(rng.rand(40, 8).astype(np.float32), np.eye(4, dtype=np.float32)[rng.randint(0, 4, 40)], "cpu", 10)
    assert isinstance(ds._on_device(), contextlib.nullcontext)
    seen = []

    class FakeDeviceGuard:
        def __init__(self, device):
            seen.append(torch.device(device))

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    monkeypatch.setattr(torch.cuda, "device", FakeDeviceGuard)
    ds.device = torch.device("cuda", 3)           # (pretend: only the guard is exercised, no launch is made)
    with ds._on_device():
        pass
    assert seen == [torch.device("cuda", 3)]


def test_next_batch_orders_are_those_of_a_fresh_images_perm_per_epoch():
    """The reshuffle is drawn one epoch ahead and gathered on a background thread into a second pair of arrays
    (data/mnist.py::DataSet._shuffle); the batches are those of ``images = images[perm]`` at every epoch boundary with
    the same generator (mnist.train.next_batch, tfdist_between.py:92), also when a batch straddles the boundary."""
    import numpy as np

    from distributed_tensorflow_amd.data.mnist import DataSet

    rng = np.random.RandomState(0)
    x = rng.rand(5000, 24).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 5000)]
    for n, batch in ((5000, 100), (1050, 100)):
        a = DataSet(x[:n].copy(), y[:n].copy(), seed=3)
        r = np.random.RandomState(3)
        xi, yi = x[:n].copy(), y[:n].copy()
        perm = r.permutation(n)
        xi, yi = xi[perm], yi[perm]
        pos = 0
        for _ in range(3 * n // batch + 5):
            bx, by = a.next_batch(batch)
            if pos + batch > n:
                rest_x, rest_y = xi[pos:], yi[pos:]
                perm = r.permutation(n)
                xi, yi = xi[perm], yi[perm]
                pos = batch - len(rest_x)
                want_x, want_y = np.concatenate((rest_x, xi[:pos])), np.concatenate((rest_y, yi[:pos]))
            else:
                want_x, want_y = xi[pos:pos + batch], yi[pos:pos + batch]
                pos += batch
            assert np.array_equal(bx, want_x) and np.array_equal(by, want_y)


def _write_idx(path, array):
    import gzip
    import struct

    array = np.ascontiguousarray(array, dtype=np.uint8)
    blob = bytes([0, 0, 0x08, array.ndim]) + struct.pack(">" + "I" * array.ndim, *array.shape) + array.tobytes()
    if str(path).endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(blob)
    else:
        with open(path, "wb") as f:
            f.write(blob)


def test_read_data_sets_reads_the_mnist_idx_files_the_reference_would_download(tmp_path):
    """``input_data.read_data_sets("MNIST_data", one_hot=True)`` (tfsingle.py:6-7): with the four idx files of the MNIST
    distribution in the directory (gzip-compressed or not) they ARE the data -- 55,000 training rows after the first 5,000
    of the training file became the validation split, 10,000 test rows, pixels scaled to [0, 1], one-hot labels.
    (Files of the distribution's format and sizes written here: there is no network to fetch the real ones.)"""
    from distributed_tensorflow_amd.data import mnist as M

    rng = np.random.RandomState(3)
    tr_l = rng.randint(0, 10, 60000)
    te_l = rng.randint(0, 10, 10000)
    tr_i = np.zeros((60000, 28, 28), np.uint8)
    te_i = np.zeros((10000, 28, 28), np.uint8)
    tr_i[np.arange(60000), tr_l, tr_l] = 255                      # one bright pixel that tells the label
    tr_i[:, 27, 27] = (np.arange(60000) % 256).astype(np.uint8)   # ... and one that tells the row
    te_i[np.arange(10000), te_l, te_l] = 128
    _write_idx(tmp_path / "train-images-idx3-ubyte.gz", tr_i)
    _write_idx(tmp_path / "train-labels-idx1-ubyte.gz", tr_l)
    _write_idx(tmp_path / "t10k-images-idx3-ubyte", te_i)         # (uncompressed files are found as well)
    _write_idx(tmp_path / "t10k-labels-idx1-ubyte", te_l)
    d = M.read_data_sets(str(tmp_path), one_hot=True)
    assert d.source == "mnist-idx"
    assert d.train.images.shape == (55000, 784) and d.validation.images.shape == (5000, 784)
    assert d.test.images.shape == (10000, 784) and d.train.labels.shape == (55000, 10)
    assert d.train.images.dtype == np.float32 and float(d.train.images.max()) == 1.0
    np.testing.assert_array_equal(d.validation.labels.argmax(1), tr_l[:5000])
    np.testing.assert_array_equal(d.train.labels.argmax(1), tr_l[5000:])
    np.testing.assert_array_equal(d.test.labels.argmax(1), te_l)
    row = 5000 + 123                                              # training row 123 is row 5123 of the file
    np.testing.assert_allclose(d.train.images[123, 27 * 28 + 27], (row % 256) / 255.0, rtol=1e-6)
    np.testing.assert_allclose(d.test.images[7, te_l[7] * 28 + te_l[7]], 128 / 255.0, rtol=1e-6)
    xb, yb = d.train.next_batch(100)
    assert xb.shape == (100, 784) and yb.shape == (100, 10) and np.all(yb.sum(1) == 1)
    # a file that is not an idx file of bytes is refused, not reshaped
    (tmp_path / "t10k-labels-idx1-ubyte").write_bytes(b"\\x00\\x00\\x0d\\x01" + b"\\x00" * 100)
    with pytest.raises(ValueError, match="idx"):
        M.read_data_sets(str(tmp_path), one_hot=True)
