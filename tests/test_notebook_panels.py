"""The first image-path facts pinned by something the REFERENCE ITSELF holds: the PNG outputs frozen in
``tutorials/images/understanding_discrete_canonicalization.ipynb`` (cells 5, 12, 19), decoded to 64 x 64 panels by
tests/golden/make_golden_notebook_panels.py -> tests/golden/notebook_panels.pt ("reference-held (notebook output)").

What the panels pin (display resolution, so direction / identity facts, not pixel tolerances):
  * cell 12: view i = crop(torchvision.transforms.functional.rotate(pad(x, 10, edge), +90 i)) is the image turned i quarter turns
    COUNTER-CLOCKWISE on screen, i.e. torch.rot90(x, +i) -- the sense of a positive torchvision angle (reference call sites:
    images/utils.py:125-136, examples/images/classification/inference_utils.py:100-123);
  * cell 19: ``canonicalizer.canonicalize`` (images/canonicalization/discrete_group.py:190-238: pad edge -> kornia rotate(-angle)
    -> crop) of the four views gives ONE image for the whole orbit, and that image is an exact quarter turn of the input --
    no blur, no border artefact: for C4 the pad -> rotate -> crop chain is a pure rot90, and the kornia rotation that undoes a
    torchvision (+theta) view is consistent over the orbit.
The oracle (CPU) and the product (GPU) are then run on the reference's own panel 0 and must turn the same way.
"""
import types

import pytest
import torch

from oracle import image_ops as io


@pytest.fixture(scope="module")
def panels(golden):
    d = golden("notebook_panels.pt")
    assert d["provenance"].startswith("reference-held (notebook output)")
    return {k: (v.float() / 255.0 if torch.is_tensor(v) else v) for k, v in d.items()}


def _mad(a, b):
    return (a - b).abs().mean().item()


def test_reference_views_turn_counter_clockwise(panels):
    v = panels["views"]
    assert _mad(panels["sample"][0], v[0]) < 0.06          # cell 5 shows the same image (rendered at another size, with axes)
    for i in range(4):
        errs = [_mad(v[i], torch.rot90(v[0], k, (1, 2))) for k in range(4)]
        assert errs[i] < 0.02, (i, errs)                     # panel i = rot90(panel 0, +i) ...
        assert all(e > 0.15 for k, e in enumerate(errs) if k != i), (i, errs)     # ... and clearly no other quarter turn


def test_reference_canonical_form_is_one_exact_quarter_turn_for_the_whole_orbit(panels):
    v, c = panels["views"], panels["canonicalized"]
    for i in range(1, 4):
        assert _mad(c[i], c[0]) < 1e-3                       # identical panels (measured 1e-6: the same pixels were rendered)
    errs = [_mad(c[0], torch.rot90(v[0], k, (1, 2))) for k in range(4)]
    k0 = min(range(4), key=lambda k: errs[k])
    assert errs[k0] < 0.02 and sorted(errs)[1] > 0.15, errs  # a pure quarter turn of the input: nothing interpolated, no border
    assert _mad(c[0], v[k0]) < 1e-3                          # in fact pixel-identical to one of the rendered views
    assert k0 == 3                                           # (this untrained e2cnn network happened to pick the 270-degree view)


def test_oracle_turns_the_way_the_reference_panels_do(panels):
    x = panels["views"][0][None]                             # the reference's own image, (1, 3, 64, 64)
    padded = io.tv_pad_edge(x, 10)                           # notebook cell 10: Pad(10, edge) -> F.rotate(angle) -> CenterCrop
    for i, ang in enumerate(panels["angles_deg"]):
        view = io.tv_center_crop(io.tv_rotate_nearest(padded[0], ang), (64, 64))
        assert torch.equal(view, torch.rot90(x[0], i, (1, 2)))                    # exact for quarter turns
        assert _mad(view, panels["views"][i]) < 0.02, i                          # = what the reference rendered
    # GroupInference's orbit (pad 0.4 H instead of 10) is the same call chain
    orbit = io.group_inference_orbit(x, 4, "rotation")
    for i in range(4):
        assert _mad(orbit[i, 0], panels["views"][i]) < 0.02, i
    # kornia(-theta) o torchvision(+theta) = identity over the orbit, as cell 19 shows: canonicalizing view i with the element
    # "rotation = r0 + 90 i" gives ONE image, the reference's canonical panel, for r0 = 90 (K.rotate(x, -90) = rot90(x, 3)) ...
    views = torch.stack([torch.rot90(x[0], i, (1, 2)) for i in range(4)])
    canon = io.canonicalize_images(views, torch.tensor([90.0, 180.0, 270.0, 0.0]), None, (3, 64, 64))
    for i in range(4):
        assert (canon[i] - canon[0]).abs().max().item() < 1e-5   # bilinear at cos(90 deg) ~ 6e-8: ~2e-6 of interpolation noise
        assert _mad(canon[i], panels["canonicalized"][i]) < 0.02, i
    # ... and pairing the views with the elements in the opposite sense does not
    wrong = io.canonicalize_images(views, torch.tensor([90.0, 0.0, 270.0, 180.0]), None, (3, 64, 64))
    assert _mad(wrong[1], panels["canonicalized"][1]) > 0.15


@pytest.mark.gpu
def test_product_turns_the_way_the_reference_panels_do(panels):
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    import equiadapt_amd as ea
    from equiadapt_amd.inference import GroupInference

    dev = torch.device("cuda:0")
    x = panels["views"][0][None].to(dev)
    # the nearest-neighbour action kernel (eqa_image_action_nearest) through GroupInference.group_orbit
    gi = GroupInference(torch.nn.Identity(), torch.nn.Identity(), 10, types.SimpleNamespace(group_type="rotation", num_rotations=4), (3, 64, 64))
    orbit = gi.group_orbit(x)
    for i in range(4):
        assert torch.equal(orbit[i, 0].cpu(), torch.rot90(x[0].cpu(), i, (1, 2)))
        assert _mad(orbit[i, 0].cpu(), panels["views"][i]) < 0.02, i
    # the canonicalizer class on the reference's four views: one canonical image for the whole orbit, an exact quarter turn
    # of the input (cell 19's two facts) -- with this repository's equivariant network in place of e2cnn's, so WHICH quarter turn
    # is the network's business
    torch.manual_seed(4)
    net = ea.ESCNNEquivariantNetwork((3, 64, 64), out_channels=16, kernel_size=9, group_type="rotation", num_rotations=4, num_layers=3)
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.9, resize_shape=64)       # notebook cell 17
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 64, 64)).to(dev).eval()
    with torch.no_grad():
        out = can(orbit[:, 0].contiguous()).cpu()
    gidx = can.canonicalization_info_dict["group_index"].cpu().long()
    assert sorted(((gidx - gidx[0]) % 4).tolist()) == [0, 1, 2, 3]                     # the element follows the view around the group
    for i in range(1, 4):
        assert (out[i] - out[0]).abs().max().item() < 1e-5, i
    errs = [(out[0] - torch.rot90(x[0].cpu(), k, (1, 2))).abs().max().item() for k in range(4)]
    assert min(errs) < 1e-5, errs
    # and through the oracle's restatement of the same chain with the product's elements: identical pixels
    ref = io.canonicalize_images(orbit[:, 0].cpu(), io.group_angles(4)[gidx], None, (3, 64, 64))
    assert (out - ref).abs().max().item() < 1e-5
