"""Generate golden vectors from the REAL reference, in the build container only.

Run:  cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

What runs the reference and how:
  * ``equiadapt/common/utils.py``, ``equiadapt/common/basecanonicalization.py`` and
    ``equiadapt/pointcloud/canonicalization_networks/vector_neuron_layers.py`` are torch-only and are
    imported UNMODIFIED by file path (``importlib.util.spec_from_file_location``).
    provenance = "reference".
  * ``equiadapt/pointcloud/canonicalization_networks/equivariant_networks.py`` and
    ``equiadapt/pointcloud/canonicalization/continuous_group.py`` fail only on
    ``from omegaconf import DictConfig`` (a type annotation).  They are imported unmodified with a
    throw-away ``sys.modules["omegaconf"]`` whose ``DictConfig`` is never called; no arithmetic passes
    through it.  provenance = "reference source + annotation-only omegaconf stand-in".
  * nothing under ``equiadapt/images`` can run here (kornia / torchvision / e2cnn absent); NO stand-ins
    are made for those.  The image fixtures written by this script come from this repo's own oracle
    restatement and are labelled provenance = "restatement (parity unpinned)".

The fixtures are data only (inputs, parameters, expected outputs, provenance strings) saved with
``torch.save`` as plain dicts of tensors.  No reference source or bytecode is stored.
"""
import importlib.util
import os
import sys
import types

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def load_by_path(mod_name: str, rel: str):
    spec = importlib.util.spec_from_file_location(mod_name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[mod_name] = mod
    spec.loader.exec_module(mod)
    return mod


def save(name: str, payload: dict) -> None:
    path = os.path.join(HERE, name)
    torch.save(payload, path)
    print(f"wrote {name}: {os.path.getsize(path)} B")


def main() -> None:
    # ---- package skeleton so the reference's absolute imports resolve without running its __init__ ----
    for pkg in ("equiadapt", "equiadapt.common", "equiadapt.pointcloud",
                "equiadapt.pointcloud.canonicalization_networks", "equiadapt.pointcloud.canonicalization"):
        m = types.ModuleType(pkg)
        m.__path__ = []  # mark as package
        sys.modules[pkg] = m
    utils = load_by_path("equiadapt.common.utils", "equiadapt/common/utils.py")
    base = load_by_path("equiadapt.common.basecanonicalization", "equiadapt/common/basecanonicalization.py")
    vnl = load_by_path("equiadapt.pointcloud.canonicalization_networks.vector_neuron_layers",
                       "equiadapt/pointcloud/canonicalization_networks/vector_neuron_layers.py")

    # ---------------- gram_schmidt: the reference's one numeric known-answer test ----------------
    torch.manual_seed(0)
    v = torch.randn(1, 3, 3)  # exactly tests/common/test_utils.py:6-12
    out = utils.gram_schmidt(v)
    assert abs(out[0, 0, 0].item() - 0.5740) < 1e-4
    torch.manual_seed(11)
    vb = torch.randn(64, 3, 3)
    save("gram_schmidt.pt", {"provenance": "reference", "kat_in": v, "kat_out": out,
                             "batch_in": vb, "batch_out": utils.gram_schmidt(vb)})

    # ---------------- discrete group: one-hot / STE / losses ----------------
    class _Disc(base.DiscreteGroupCanonicalization):
        pass

    torch.manual_seed(12)
    acts = torch.randn(32, 8)
    acts[3, 2] = acts[3, 5] = acts[3].max() + 1.0  # an exact tie: argmax must take the first
    cases = {}
    for training in (False, True):
        d = _Disc(torch.nn.Identity(), beta=0.7)
        d.num_group = 8
        d.train(training)
        a = acts.clone().requires_grad_(True)
        oh = d.groupactivations_to_groupelementonehot(a)
        w = torch.arange(8.0)
        grad = None
        if training:  # eval mode returns the bare one-hot: no graph, as in the reference
            (oh * w).sum().backward()
            grad = a.grad.clone()
        cases["train" if training else "eval"] = {"onehot": oh.detach(), "grad": grad}
    d = _Disc(torch.nn.Identity())
    d.device = torch.device("cpu")
    d.canonicalization_info_dict = {"group_activations": acts}
    save("discrete_group.pt", {"provenance": "reference", "beta": 0.7, "acts": acts, "cases": cases,
                               "prior_loss": d.get_prior_regularization_loss(),
                               "identity_metric": d.get_identity_metric()})

    # ---------------- continuous group losses ----------------
    c = base.ContinuousGroupCanonicalization(torch.nn.Identity())
    c.device = torch.device("cpu")
    rep = utils.gram_schmidt(vb)
    c.canonicalization_info_dict = {"group_element_matrix_representation": rep}
    save("continuous_group.pt", {"provenance": "reference", "rep": rep,
                                 "prior_loss": c.get_prior_regularization_loss(),
                                 "identity_metric": c.get_identity_metric()})

    # ---------------- VN layers (direct) ----------------
    torch.manual_seed(13)
    lay = vnl.VNLinearLeakyReLU(3, 21, dim=5, negative_slope=0.0)
    bn = vnl.VNBatchNorm(21, dim=4)
    mp = vnl.VNMaxPool(21)
    for m in (lay, bn):
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.running_mean.normal_(0.5, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.normal_(0, 0.2)
    x5 = torch.randn(2, 3, 3, 16, 5)
    x4 = torch.randn(2, 21, 3, 16)
    lay.eval(); bn.eval()
    vn_payload = {"provenance": "reference",
                  "lin_state": {k: v.clone() for k, v in lay.state_dict().items()}, "lin_in": x5,
                  "lin_out_eval": lay(x5).detach(),
                  "bn_state": {k: v.clone() for k, v in bn.state_dict().items()}, "bn_in": x4,
                  "bn_out_eval": bn(x4).detach(),
                  "pool_state": {k: v.clone() for k, v in mp.state_dict().items()},
                  "pool_in": torch.randn(2, 21, 3, 16, 5)}
    vn_payload["pool_out"] = mp(vn_payload["pool_in"]).detach()
    vn_payload["mean_pool_out"] = vnl.mean_pool(vn_payload["pool_in"])
    lay.train()
    vn_payload["lin_out_train"] = lay(x5).detach()  # batch statistics
    save("vn_layers.pt", vn_payload)

    # ---------------- VNSmall + point-cloud canonicalizer (omegaconf stand-in, annotation only) ----------------
    oc = types.ModuleType("omegaconf")

    class DictConfig:  # never instantiated or called by the code under test
        pass

    oc.DictConfig = DictConfig
    sys.modules["omegaconf"] = oc
    eqn = load_by_path("equiadapt.pointcloud.canonicalization_networks.equivariant_networks",
                       "equiadapt/pointcloud/canonicalization_networks/equivariant_networks.py")
    pcc = load_by_path("equiadapt.pointcloud.canonicalization.continuous_group",
                       "equiadapt/pointcloud/canonicalization/continuous_group.py")
    prov = "reference source + annotation-only omegaconf stand-in"
    payload = {"provenance": prov}
    for pooling in ("mean", "max"):
        torch.manual_seed(2)
        hp = types.SimpleNamespace(n_knn=20, pooling=pooling)
        net = eqn.VNSmall(hp)
        torch.manual_seed(14)
        for mod in net.modules():  # non-trivial running stats so eval-mode BN is exercised
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.running_mean.normal_(0.5, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.normal_(0, 0.2)
        net.eval()
        torch.manual_seed(0)
        x = torch.randn(4, 3, 256)
        can = pcc.EquivariantPointcloudCanonicalization(net, hp)
        can.eval()
        with torch.no_grad():
            idx = eqn.knn(x, 20)
            feat = eqn.get_graph_feature_cross(x.unsqueeze(1), k=20)
            vec = net(x)
            xc = can(x)
        payload[pooling] = {
            "state": {k: v.clone() for k, v in net.state_dict().items()},
            "x": x, "knn_idx": idx, "graph_feature": feat, "vnsmall_out": vec,
            "rotation": can.canonicalization_info_dict["group_element_matrix_representation"].clone(),
            "x_canonicalized": xc,
            "prior_loss": can.get_prior_regularization_loss(), "identity_metric": can.get_identity_metric(),
        }
    # train-mode forward of the network (batch statistics; dropout disabled by p=0 to stay deterministic)
    torch.manual_seed(2)
    net = eqn.VNSmall(types.SimpleNamespace(n_knn=20, pooling="mean"))
    net.dropout.p = 0.0
    net.train()
    torch.manual_seed(0)
    x = torch.randn(4, 3, 256)
    st0 = {k: v.clone() for k, v in net.state_dict().items()}
    payload["mean_train"] = {"state": st0, "x": x, "vnsmall_out": net(x).detach(),
                             "state_after": {k: v.clone() for k, v in net.state_dict().items()}}
    save("pointcloud.pt", payload)

    # ---------------- n-body E(3) canonicalizer (torch-only reference module, fake network) ----------------
    sys.modules.setdefault("equiadapt.nbody", types.ModuleType("equiadapt.nbody")).__path__ = []
    sys.modules.setdefault("equiadapt.nbody.canonicalization", types.ModuleType("equiadapt.nbody.canonicalization")).__path__ = []
    nb = load_by_path("equiadapt.nbody.canonicalization.euclidean_group", "equiadapt/nbody/canonicalization/euclidean_group.py")
    torch.manual_seed(21)
    M = 40
    rot_vec, trans = torch.randn(M, 3, 3), torch.randn(M, 3)

    class FakeNet(torch.nn.Module):
        def forward(self, nodes, loc, edges, vel, edge_attr, charges):
            return rot_vec, trans

    can = nb.EuclideanGroupNBody(FakeNet())
    nodes, loc, vel = torch.randn(M, 1), torch.randn(M, 3), torch.randn(M, 3)
    cl, cv = can(nodes, loc=loc, edges=None, vel=vel, edge_attr=None, charges=None)
    pred = torch.randn(M, 3)
    save("nbody.pt", {"provenance": "reference", "rot_vec": rot_vec, "trans": trans, "loc": loc, "vel": vel,
                      "rotation": can.canonicalization_info_dict["group_element"]["rotation_matrix"].clone(),
                      "canonical_loc": cl, "canonical_vel": cv, "pred": pred,
                      "inverted": can.invert_canonicalization(pred)})

    # ---------------- image path: restatement-generated, LABELLED ----------------
    from oracle import image_ops as io

    prov = "restatement (parity unpinned): oracle/image_ops.py, not the reference"
    torch.manual_seed(0)
    x = torch.randn(4, 3, 32, 32)
    gid8 = torch.tensor([0, 3, 5, 6])
    ang8 = io.group_angles(8)[gid8]
    f8 = torch.randn(4, 16, 32, 32)
    gidd4 = torch.tensor([1, 4, 6, 3])  # D4: index >= 4 carries a reflection
    angd4 = torch.cat([io.group_angles(4)] * 2)[gidd4]
    refd4 = (gidd4 >= 4).float()
    img = {
        "provenance": prov, "x": x,
        "c8": {"gidx": gid8, "canon": io.canonicalize_images(x, ang8, None, (3, 32, 32)),
               "f": f8, "invert_regular": io.invert_action(f8, ang8, None, 8, 8, "regular"),
               "invert_scalar": io.invert_action(f8[:, :3], ang8, None, 8, 8, "scalar")},
        "d4": {"gidx": gidd4, "canon": io.canonicalize_images(x, angd4, refd4, (3, 32, 32)),
               "f": f8, "invert_regular": io.invert_action(f8, angd4, refd4, 4, 8, "regular"),
               "invert_scalar": io.invert_action(f8[:, :3], angd4, refd4, 4, 8, "scalar")},
        "pre": {"crop_ratio": 0.8, "resize": 16,
                "out": io.pre_canonicalization_transform(x, (3, 32, 32), 0.8, 16)},
        "orbit_d4": io.orbit_expand(x[:2], 4, "roto-reflection", 32),
        "gray_c4": io.canonicalize_images(x[:, :1], io.group_angles(4)[torch.tensor([0, 1, 2, 3])], None, (1, 32, 32)),
    }
    masks = (torch.rand(3, 32, 32) > 0.5).to(torch.uint8)
    img["masks"] = {"in": masks, "rot_m45": io.rotate_masks(masks, -45.0), "rot_90": io.rotate_masks(masks, 90.0)}
    save("images_restatement.pt", img)


if __name__ == "__main__":
    main()
