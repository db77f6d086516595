"""Golden vectors for ConvNetwork from the reference's own class, in the build container only.

Run:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_convnet.py

``equiadapt/images/canonicalization_networks/custom_nonequivariant_networks.py`` is imported UNMODIFIED by file path.  Its
line 4 is ``import torchvision`` (absent in this image), used only inside the constructors of the pretrained ResNet wrappers
further down the file (:83-230, out of scope).  ``sys.modules["torchvision"]`` is therefore set to an EMPTY module object --
no attribute at all, so any use of it would raise AttributeError -- for the duration of the import.  ``ConvNetwork`` (:8-80)
touches torch only; no arithmetic passes through the stand-in.
provenance = "reference source + empty (attribute-less) torchvision module object; ConvNetwork is torch-only".

Data only is saved (inputs, state dicts, outputs, provenance string).
"""
import importlib.util
import os
import sys
import types

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def main() -> None:
    sys.modules["torchvision"] = types.ModuleType("torchvision")          # empty: nothing can be computed through it
    spec = importlib.util.spec_from_file_location(
        "ref_custom_nonequivariant_networks",
        os.path.join(REF, "equiadapt/images/canonicalization_networks/custom_nonequivariant_networks.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    del sys.modules["torchvision"]
    assert not [a for a in vars(sys.modules.get("torchvision", types.ModuleType("x"))) if not a.startswith("__")]

    payload = {"provenance": "reference source + empty (attribute-less) torchvision module object; ConvNetwork is torch-only", "cases": []}
    # (in_shape, out_channels, kernel_size, num_layers, out_vector_size, batch): the tutorial config (k5, 16ch, 3 layers, 64x64),
    # the segmentation config of BASELINE cfg5 (k7, 16ch, 3 layers on 128x128) and a 6-layer net that widens twice; the output
    # vectors are narrower than the configs' 128 to keep the fixture small (the Linear layer is most of the parameters)
    for ci, (in_shape, oc, k, L, V, B) in enumerate([((3, 64, 64), 16, 5, 3, 32, 6), ((3, 128, 128), 16, 7, 3, 16, 3),
                                                     ((1, 140, 140), 4, 3, 6, 32, 4)]):
        torch.manual_seed(30 + ci)
        net = mod.ConvNetwork(in_shape, oc, k, L, V)
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.normal_(0.1, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
        state = {k_: v.clone() for k_, v in net.state_dict().items()}
        x = torch.randn(B, *in_shape)
        net.eval()
        with torch.no_grad():
            out_eval = net(x)
        net.train()
        net.final_fc[1].p = 0.0          # Dropout1d is stochastic: disabled for the deterministic train-mode vector
        out_train = net(x)
        w = torch.randn(B, V)
        (out_train * w).sum().backward()
        payload["cases"].append({
            "args": (in_shape, oc, k, L, V), "x": x, "state": state, "out_eval": out_eval,
            "out_train": out_train.detach(), "upstream": w,
            "state_after_train": {k_: v.clone() for k_, v in net.state_dict().items() if "running" in k_ or "num_batches" in k_},
            "grads": {n: p.grad.clone() for n, p in net.named_parameters()},
        })
    # the constructor's dry run happens in training mode and so touches the batch-norm buffers (:55-58): pin that too
    torch.manual_seed(40)
    net = mod.ConvNetwork((3, 32, 32), 8, 3, 2, 16)
    payload["fresh_state_seed40"] = {k_: v.clone() for k_, v in net.state_dict().items()}
    path = os.path.join(HERE, "conv_network.pt")
    torch.save(payload, path)
    print(f"wrote conv_network.pt: {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
