"""Reference-generated point-cloud vectors at BASELINE config 4's own size (1024 points; the reference's ModelNet40 loader:
examples/pointcloud/classification/configs/dataset/default.yaml:4-5), and for neighbourhood sizes other than 20.

Run (build container only):  cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_pointcloud1024.py

Same procedure and provenance as tests/golden/make_golden.py: the reference's
``equiadapt/pointcloud/canonicalization_networks/{vector_neuron_layers,equivariant_networks}.py``,
``equiadapt/pointcloud/canonicalization/continuous_group.py`` and ``equiadapt/common/*.py`` are imported UNMODIFIED by file
path; ``omegaconf`` (absent here, used by those files only as a type annotation) is a throw-away module object whose
``DictConfig`` is never called.  The fixture holds data only: inputs, parameters, expected outputs, provenance strings.

Cases (all seeded):
  n1024/{mean,max}    B=8, N=1024, k=20: eval-mode VNSmall output, kNN index (int16), the canonicalizer's rotation and cloud
  n1024/mean_train    the training-mode forward (batch statistics, dropout p=0), running statistics after the step, and the
                      parameter gradients of sum(out * w) for a fixed w
  k16/{mean,max,mean_train}   B=4, N=512, k=16
  k8, k32             B=3, N=300: eval-mode mean pooling (k=8) and max pooling (k=32)
"""
import importlib.util
import os
import sys
import types

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def load_by_path(mod_name: str, rel: str):
    spec = importlib.util.spec_from_file_location(mod_name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[mod_name] = mod
    spec.loader.exec_module(mod)
    return mod


def main() -> None:
    for pkg in ("equiadapt", "equiadapt.common", "equiadapt.pointcloud",
                "equiadapt.pointcloud.canonicalization_networks", "equiadapt.pointcloud.canonicalization"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    load_by_path("equiadapt.common.utils", "equiadapt/common/utils.py")
    load_by_path("equiadapt.common.basecanonicalization", "equiadapt/common/basecanonicalization.py")
    load_by_path("equiadapt.pointcloud.canonicalization_networks.vector_neuron_layers",
                 "equiadapt/pointcloud/canonicalization_networks/vector_neuron_layers.py")
    oc = types.ModuleType("omegaconf")

    class DictConfig:  # never instantiated or called by the code under test
        pass

    oc.DictConfig = DictConfig
    sys.modules["omegaconf"] = oc
    eqn = load_by_path("equiadapt.pointcloud.canonicalization_networks.equivariant_networks",
                       "equiadapt/pointcloud/canonicalization_networks/equivariant_networks.py")
    pcc = load_by_path("equiadapt.pointcloud.canonicalization.continuous_group",
                       "equiadapt/pointcloud/canonicalization/continuous_group.py")

    def randomise_norms(net, seed):
        torch.manual_seed(seed)
        for mod in net.modules():  # non-trivial running statistics / affine maps so eval-mode BN is exercised
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.running_mean.normal_(0.5, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.normal_(0, 0.2)

    def eval_case(B, N, k, pooling, seed):
        torch.manual_seed(2 + seed)
        hp = types.SimpleNamespace(n_knn=k, pooling=pooling)
        net = eqn.VNSmall(hp)
        randomise_norms(net, 14 + seed)
        net.eval()
        torch.manual_seed(100 + seed)
        x = torch.randn(B, 3, N)
        can = pcc.EquivariantPointcloudCanonicalization(net, hp)
        can.eval()
        with torch.no_grad():
            idx = eqn.knn(x, k)
            vec = net(x)
            xc = can(x)
        assert idx.max().item() < 2 ** 15
        return {"B": B, "N": N, "k": k, "pooling": pooling,
                "state": {n: v.clone() for n, v in net.state_dict().items()},
                "x": x, "knn_idx": idx.to(torch.int16), "vnsmall_out": vec,
                "rotation": can.canonicalization_info_dict["group_element_matrix_representation"].clone(),
                "x_canonicalized": xc,
                "prior_loss": can.get_prior_regularization_loss(), "identity_metric": can.get_identity_metric()}

    def train_case(B, N, k, seed):
        torch.manual_seed(2 + seed)
        net = eqn.VNSmall(types.SimpleNamespace(n_knn=k, pooling="mean"))
        randomise_norms(net, 14 + seed)
        net.dropout.p = 0.0            # deterministic; the dropout mask is torch's own RNG stream either way
        net.train()
        torch.manual_seed(100 + seed)
        x = torch.randn(B, 3, N)
        w = torch.randn(B, 3, 3)
        st0 = {n: v.clone() for n, v in net.state_dict().items()}
        out = net(x)
        (out * w).sum().backward()
        return {"B": B, "N": N, "k": k, "state": st0, "x": x, "w": w, "vnsmall_out": out.detach(),
                "state_after": {n: v.clone() for n, v in net.state_dict().items()},
                "grads": {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}}

    payload = {"provenance": "reference source + annotation-only omegaconf stand-in",
               "n1024": {"mean": eval_case(8, 1024, 20, "mean", 0), "max": eval_case(8, 1024, 20, "max", 1),
                         "mean_train": train_case(8, 1024, 20, 2)},
               "k16": {"mean": eval_case(4, 512, 16, "mean", 3), "max": eval_case(4, 512, 16, "max", 4),
                       "mean_train": train_case(4, 512, 16, 5)},
               "k8": {"mean": eval_case(3, 300, 8, "mean", 6)},
               "k32": {"max": eval_case(3, 300, 32, "max", 7)}}
    path = os.path.join(HERE, "pointcloud_n1024.pt")
    torch.save(payload, path)
    print(f"wrote {path}: {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
