"""Placement and variable layout (round-robin order reproduces the reference's mapping)."""
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.placement import (DeviceSpec, GreedyLoadBalancingStrategy, ReplicaDeviceSetter,
                                                  replica_device_setter)
from distributed_tensorflow_amd.variables import ShardLayout, assign_shards, mlp_var_specs

ORDER = ["global_step", "weights/Variable", "weights/Variable_1", "biases/Variable", "biases/Variable_1"]


def _owners(n):
    specs = mlp_var_specs()
    assign_shards(specs, n)
    return {s.name: s.shard for s in specs}


def test_round_robin_one_shard():
    assert set(_owners(1).values()) == {0}


def test_round_robin_two_shards_matches_reference():
    # creation order global_step, W1, W2, b1, b2 -> ps0: global_step, W2, b2 ; ps1: W1, b1
    o = _owners(2)
    assert [o[n] for n in ORDER] == [0, 1, 0, 1, 0]
    lay = ShardLayout(_specs(2), 2)
    assert lay.shard_numel(1) == 78400 + 100 and lay.shard_numel(0) == 4 + 1000 + 12


def _specs(n, strategy=None):
    specs = mlp_var_specs()
    assign_shards(specs, n, strategy)
    return specs


@pytest.mark.parametrize("n", [3, 8])
def test_round_robin_many_shards(n):
    o = _owners(n)
    assert [o[name] for name in ORDER] == [i % n for i in range(5)]
    lay = ShardLayout(_specs(n), n)
    assert sum(lay.shard_numel(k) for k in range(n)) == lay.total
    if n == 8:
        assert [lay.shard_numel(k) for k in (5, 6, 7)] == [0, 0, 0]


def test_greedy_strategy_balances_bytes():
    specs = _specs(2, GreedyLoadBalancingStrategy(2))
    big = [s for s in specs if s.name == "weights/Variable"][0]
    others = [s.shard for s in specs if s.name not in ("weights/Variable", "global_step")]
    assert all(sh != big.shard for sh in others)


def test_flat_layout_equal_shards_and_contiguous_views():
    lay = ShardLayout(_specs(8), 8, mode="flat")
    sizes = {lay.shard_numel(k) for k in range(8)}
    assert len(sizes) == 1 and lay.total % 8 == 0
    slab = lay.new_slab("cpu")
    v = lay.mlp_views(slab)
    assert v["W1"].shape == (784, 100) and v["W1"].is_contiguous() and v["b2"].shape == (10,)
    assert all(o % 4 == 0 for o, _ in lay.var_range.values())


def test_initialisation_is_deterministic_and_shard_consistent():
    lay = ShardLayout(_specs(2), 2)
    a, b = lay.new_slab("cpu"), lay.new_slab("cpu")
    lay.initialize(a, seed=1)
    lay.initialize(b, seed=1)
    assert torch.equal(a, b)
    W1 = lay.view(a, "weights/Variable")
    assert abs(float(W1.mean())) < 0.02 and abs(float(W1.std()) - 1) < 0.02
    assert float(lay.view(a, "biases/Variable").abs().max()) == 0 and float(lay.view(a, "global_step")) == 0
    # initialising one shard's storage alone reproduces the same values
    s0, e0 = lay.shard_range[1]
    shard = torch.zeros(e0 - s0)
    lay.initialize(shard, seed=1, shard=1, slab_offset=s0)
    assert torch.equal(shard, a[s0:e0])
    c = lay.new_slab("cpu")
    lay.initialize(c, seed=2)
    assert not torch.equal(a, c)


def test_device_spec_and_setter_through_graph_api():
    with tf.Graph().as_default() as g:
        cluster = tf.train.ClusterSpec({"ps": ["h:1", "h:2"], "worker": ["h:3"]})
        setter = replica_device_setter(worker_device="/job:worker/task:0/gpu:0", cluster=cluster)
        assert isinstance(setter, ReplicaDeviceSetter)
        with tf.device(setter):
            gs = tf.get_variable("global_step", [], initializer=tf.constant_initializer(0), trainable=False)
            x = tf.placeholder(tf.float32, [None, 4])
            with tf.name_scope("weights"):
                W1 = tf.Variable(tf.random_normal([4, 3]))
                W2 = tf.Variable(tf.random_normal([3, 2]))
            y = tf.matmul(x, W1)
        assert [gs.device, W1.device, W2.device] == ["/job:ps/task:0", "/job:ps/task:1", "/job:ps/task:0"]
        assert y.device == "/job:worker/task:0/gpu:0" and x.device == "/job:worker/task:0/gpu:0"
        assert [v.op_name for v in g.variables] == ["global_step", "weights/Variable", "weights/Variable_1"]
        assert [v.spec.shard for v in g.variables] == [0, 1, 0]
    d = DeviceSpec.from_string("/job:worker/task:3/gpu:1")
    assert (d.job, d.task, d.device_type, d.device_index) == ("worker", 3, "gpu", 1)
    assert DeviceSpec.from_string("/gpu:0").merged_over(DeviceSpec.from_string("/job:ps/task:1")).to_string() == \
        "/job:ps/task:1/gpu:0"


def test_name_scopes_uniquify_like_tensorflow():
    with tf.Graph().as_default():
        with tf.name_scope("biases"):
            a = tf.Variable(tf.zeros([3]))
            b = tf.Variable(tf.zeros([3]))
        with tf.name_scope("biases"):
            c = tf.Variable(tf.zeros([3]))
        assert [a.op_name, b.op_name, c.op_name] == ["biases/Variable", "biases/Variable_1", "biases_1/Variable"]
        assert a.name == "biases/Variable:0"
