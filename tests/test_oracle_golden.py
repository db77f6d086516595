"""Pin the CPU oracle against vectors produced by the real reference (tests/golden/make_golden.py).

CPU-only.  Tolerances: these compare two fp32 CPU evaluations of the same op sequence, so they are tight.
"""
import torch

from oracle import image_ops as io
from oracle import pointcloud_ops as po


def test_gram_schmidt_reference_kat(golden):
    g = golden("gram_schmidt.pt")
    assert g["provenance"] == "reference"
    out = po.gram_schmidt(g["kat_in"])
    # the reference's own known-answer test: tests/common/test_utils.py:6-12
    assert torch.allclose(out[0][0][0], torch.tensor(0.5740), atol=1e-4)
    assert torch.equal(out, g["kat_out"])
    assert torch.allclose(po.gram_schmidt(g["batch_in"]), g["batch_out"], atol=1e-6, rtol=0)


def test_discrete_onehot_ste_and_losses(golden):
    g = golden("discrete_group.pt")
    acts = g["acts"]
    for mode in ("eval", "train"):
        a = acts.clone().requires_grad_(True)
        oh = io.onehot_from_activations(a, 8, g["beta"], training=(mode == "train"))
        assert torch.equal(oh.detach(), g["cases"][mode]["onehot"]), mode
        if mode == "train":
            (oh * torch.arange(8.0)).sum().backward()
            assert torch.allclose(a.grad, g["cases"][mode]["grad"], atol=1e-7, rtol=0)
    # the engineered tie at row 3 (columns 2 and 5) resolves to the first index
    assert io.onehot_from_activations(acts, 8, 1.0, False)[3].argmax().item() == 2
    assert torch.equal(io.prior_regularization_loss(acts), g["prior_loss"])
    assert torch.equal(io.identity_metric(acts), g["identity_metric"])


def test_continuous_losses(golden):
    g = golden("continuous_group.pt")
    assert torch.equal(po.continuous_prior_loss(g["rep"]), g["prior_loss"])
    assert torch.equal(po.continuous_identity_metric(g["rep"]), g["identity_metric"])


def test_vn_layers(golden):
    g = golden("vn_layers.pt")
    p = {"l." + k: v for k, v in g["lin_state"].items()}
    out = po.vn_linear_leaky_relu(g["lin_in"], p, "l.", 5, training=False)
    assert torch.allclose(out, g["lin_out_eval"], atol=1e-6, rtol=1e-6)
    p_tr = {k: v.clone() for k, v in p.items()}
    out = po.vn_linear_leaky_relu(g["lin_in"], p_tr, "l.", 5, training=True)
    assert torch.allclose(out, g["lin_out_train"], atol=1e-5, rtol=1e-5)
    p = {"b." + k: v for k, v in g["bn_state"].items()}
    assert torch.allclose(po.vn_batchnorm(g["bn_in"], p, "b.", 4, False), g["bn_out_eval"], atol=1e-6, rtol=1e-6)
    p = {"p." + k: v for k, v in g["pool_state"].items()}
    assert torch.equal(po.vn_max_pool(g["pool_in"], p, "p."), g["pool_out"])
    assert torch.equal(g["pool_in"].mean(-1), g["mean_pool_out"])


def test_vnsmall_and_pointcloud_canonicalize(golden):
    g = golden("pointcloud.pt")
    assert "omegaconf stand-in" in g["provenance"]
    for pooling in ("mean", "max"):
        c = g[pooling]
        idx = po.knn(c["x"], 20)
        assert torch.equal(idx, c["knn_idx"])
        assert torch.equal(po.graph_feature_cross(c["x"].unsqueeze(1), 20), c["graph_feature"])
        vec = po.vnsmall_forward(c["x"], c["state"], 20, pooling, training=False)
        assert torch.allclose(vec, c["vnsmall_out"], atol=1e-6, rtol=1e-5), pooling
        R = po.gram_schmidt(vec)
        assert torch.allclose(R, c["rotation"], atol=1e-5, rtol=0)
        assert torch.allclose(po.canonicalize_pointcloud(c["x"], c["rotation"]), c["x_canonicalized"], atol=1e-6, rtol=0)
        assert torch.allclose(po.continuous_prior_loss(c["rotation"]), c["prior_loss"])
        assert torch.allclose(po.continuous_identity_metric(c["rotation"]), c["identity_metric"])
    t = g["mean_train"]
    st = {k: v.clone() for k, v in t["state"].items()}
    vec = po.vnsmall_forward(t["x"], st, 20, "mean", training=True)
    assert torch.allclose(vec, t["vnsmall_out"], atol=1e-5, rtol=1e-4)
    # running statistics were updated in place exactly like the reference modules do
    for k in ("conv_pos.batchnorm.bn2d.running_mean", "bn1.bn1d.running_var"):
        assert torch.allclose(st[k], t["state_after"][k], atol=1e-6, rtol=1e-5), k


def test_image_restatement_fixture_is_labelled_and_reproducible(golden):
    g = golden("images_restatement.pt")
    assert "parity unpinned" in g["provenance"]
    x = g["x"]
    ang = io.group_angles(8)[g["c8"]["gidx"]]
    assert torch.equal(io.canonicalize_images(x, ang, None, (3, 32, 32)), g["c8"]["canon"])


def test_nbody_e3_canonicalizer(golden):
    g = golden("nbody.pt")
    assert g["provenance"] == "reference"
    R = po.modified_gram_schmidt(g["rot_vec"])
    assert torch.equal(R, g["rotation"])
    cl, cv = po.nbody_canonicalize(g["loc"], g["vel"], R, g["trans"])
    assert torch.equal(cl, g["canonical_loc"]) and torch.equal(cv, g["canonical_vel"])
    assert torch.equal(po.nbody_invert(g["pred"], R, g["trans"]), g["inverted"])
