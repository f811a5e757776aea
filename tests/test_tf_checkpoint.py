"""Row f-2: TensorFlow checkpoint reader.  The two `.index` files under tests/golden/tf_ckpt are DATA files
shipped by the reference (model/2000pickle_base, 1.8 KB and 8.7 KB); the 1.2 MB weight shards stay in the
reference checkout, so the tests that need them run only where /root/reference is mounted."""
import os

import numpy as np
import pytest
import torch

import tf_checkpoint as T

FX = os.path.join(os.path.dirname(__file__), "golden", "tf_ckpt")
REF = "/root/reference/model/2000pickle_base"


def test_index_of_published_actor():
    idx = T.read_index(os.path.join(FX, "Agent1_Actor_pickle.index"))
    kern = {k: v for k, v in idx.items() if k.endswith("kernel/.ATTRIBUTES/VARIABLE_VALUE")}
    assert len(kern) == 13                                             # 13 GCN layers (truss2D_RL.py:49-120)
    assert idx["gcn_l1_1/kernel/.ATTRIBUTES/VARIABLE_VALUE"]["shape"] == (13, 200)
    assert idx["gcn_l1_4/kernel/.ATTRIBUTES/VARIABLE_VALUE"]["shape"] == (4, 200)
    assert idx["gcn_l4_1/kernel/.ATTRIBUTES/VARIABLE_VALUE"]["shape"] == (200, 2)
    assert idx["gcn_l4_2/kernel/.ATTRIBUTES/VARIABLE_VALUE"]["shape"] == (200, 3)
    n = sum(int(np.prod(v["shape"])) for k, v in idx.items() if k.startswith("gcn_"))
    assert n == 291805                                                 # 3x(13x200+200) + (4x200+200) + 7x(200x200+200) + (200x2+2) + (200x3+3)
    # offsets tile the data shard without overlap
    spans = sorted((v["offset"], v["offset"] + v["size"]) for v in idx.values() if v["size"])
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    assert "_CHECKPOINTABLE_OBJECT_GRAPH" in idx


def test_index_of_published_critic_lists_all_layers():
    idx = T.read_index(os.path.join(FX, "Agent1_Critic_pickle.index"))
    gcn = {k.split("/")[0] for k in idx if k.startswith("gcn_")}
    assert len(gcn) == 21 and {"dense_1", "dense_2"} <= {k.split("/")[0] for k in idx}      # 21 GCN + MLP


def test_bad_file_is_refused(tmp_path):
    p = tmp_path / "x.index"
    p.write_bytes(b"not a table" * 10)
    with pytest.raises(ValueError):
        T.read_index(str(p))


def test_crc32c_known_answers():
    assert T.crc32c(b"123456789") == 0xE3069283                       # CRC-32C check value
    assert T.crc32c(b"") == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Agent1_Actor_pickle.data-00000-of-00001")),
                    reason="weight shards live in the reference checkout only")
def test_published_actors_load_and_run():
    import truss2D_RL as RL
    for agent in (1, 2, 3):
        actor = RL.multimodes_actor(200, 2, 3)
        n = T.load_gcn_actor(actor, os.path.join(REF, f"Agent{agent}_Actor_pickle"))     # CRC-checked
        assert n == 291805
        N, Pn = 16, 20
        g = torch.Generator().manual_seed(agent)
        A = torch.eye(N).unsqueeze(0)
        ins = (torch.rand(1, N, 13, generator=g), A, A, A, A, torch.rand(1, Pn, 4, generator=g), torch.eye(Pn).unsqueeze(0))
        with torch.no_grad():
            geo, topo = actor(ins)
        assert geo.shape == (1, N, 2) and topo.shape == (1, N, 3)
        assert torch.isfinite(geo).all() and (geo > 0).all() and (geo < 1).all() and (topo > 0).all() and (topo < 1).all()
        assert sum(p.numel() for p in actor.parameters()) == 291805
    # the critic shards are not shipped (model/.../.MISSING_LARGE_BLOBS): a clear error, not garbage
    with pytest.raises((ValueError, FileNotFoundError)):
        T.load_variables(os.path.join(REF, "Agent1_Critic_pickle"))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Agent1_Actor_pickle.data-00000-of-00001")),
                    reason="weight shards live in the reference checkout only")
def test_maddpg_restores_published_actors():
    import truss2D_RL as RL
    import master_DDPG_truss2D_MO as M
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device="cpu")
    rl.load_weights(REF + "/")                      # the reference's own directory layout (master…:710-729)
    v = T.load_variables(os.path.join(REF, "Agent2_Actor_pickle"), verify=False)
    w = rl.agents[1].actor_model.gcn_l2_3.lin.weight.detach().numpy()
    assert np.array_equal(w, v["gcn_l2_3/kernel"].T)
    assert np.array_equal(rl.agents[1].target_actor_model.gcn_l4_2.bias.detach().numpy(), v["gcn_l4_2/bias"])
