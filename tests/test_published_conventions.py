"""Known answers from the PUBLISHED documentation / source of the third-party libraries the reference's image path calls
(kornia 0.7.0, torchvision 0.17.0), against the oracle's restatements of them (oracle/image_ops.py ``kornia_*`` / ``tv_*``) and
against the product's host-built tables (equiadapt_amd/images/geometry.py).

Provenance label of every vector in this file: **published-docs** -- numbers printed in the libraries' own docstrings, or the
libraries' documented formulas evaluated by hand in the comment next to the vector.  They are not outputs of the reference run
here (kornia / torchvision are not installed in the build container); what they pin is the sign, centre and
rounding CONVENTIONS the restatement rests on, by something other than the builder's memory of the implementation:

* kornia ``get_rotation_matrix2d`` docstring example (center 0, 45 degrees, unit scale)
      tensor([[[ 0.7071,  0.7071,  0.0000], [-0.7071,  0.7071,  0.0000]]])
  and its parameter doc "Positive values mean counter-clockwise rotation (the coordinate origin is assumed to be the top-left
  corner)";
* kornia ``hflip`` docstring example;
* torchvision ``functional.rotate`` parameter doc "angle (number): rotation angle value in degrees, counter-clockwise", and the
  documented inverse-affine construction of ``_get_inverse_affine_matrix`` (the comment block above it in
  torchvision/transforms/functional.py:  M^-1 = C * RSS^-1 * C^-1 * T^-1,  RSS^-1 = [d, -b, 0, -c, a, 0]);
* torchvision ``CenterCrop``: ``crop_top = int(round((image_height - crop_height) / 2.0))``;
* torchvision ``Pad`` doc: "edge: pads with the last value at the edge of the image";
* torchvision ``Resize`` doc: "If size is an int, smaller edge of the image will be matched to this number. i.e, if
  height > width, then image will be rescaled to (size * height / width, size)".
"""
import math

import torch

from equiadapt_amd.images import geometry
from oracle import image_ops as io

PROVENANCE = "published-docs"


def test_kornia_get_rotation_matrix2d_docstring_example():
    want = torch.tensor([[[0.7071, 0.7071, 0.0000], [-0.7071, 0.7071, 0.0000]]])     # docstring of get_rotation_matrix2d
    got = io.kornia_rotation_matrix2d(torch.zeros(1, 2), 45.0 * torch.ones(1))
    assert torch.allclose(got, want, atol=5e-5)


def _pixel_matrix_from_theta(theta6: torch.Tensor, hw):
    """Undo kornia's normalisation: theta = (N M N^-1)^-1[:2]  ->  M (3x3, pixel coordinates of the (H, W) frame)."""
    H, W = hw
    N = io.kornia_normal_transform_pixel(H, W)[0].double()
    T = torch.eye(3, dtype=torch.float64)
    T[:2] = theta6.view(2, 3).double()
    return torch.linalg.inv(N) @ torch.linalg.inv(T) @ N


def test_product_tables_hold_the_documented_rotation_matrix_about_the_frame_centre():
    """The product's E x 6 tables (what the HIP kernels sample with) are the documented matrix: rotation block
    [[cos, sin], [-sin, cos]] for ``rotate(img, angle)`` and the centre ((W-1)/2, (H-1)/2) a fixed point."""
    for hw in ((64, 64), (448, 448), (40, 72)):
        th = geometry.rotation_theta(torch.tensor([45.0, 90.0, -45.0]), hw)
        for row, deg in zip(th, (45.0, 90.0, -45.0)):
            M = _pixel_matrix_from_theta(row, hw)
            c, s_ = math.cos(math.radians(deg)), math.sin(math.radians(deg))
            want = torch.tensor([[c, s_], [-s_, c]], dtype=torch.float64)               # docstring layout, general angle
            assert torch.allclose(M[:2, :2], want, atol=1e-5), (hw, deg)
            centre = torch.tensor([(hw[1] - 1) / 2, (hw[0] - 1) / 2, 1.0], dtype=torch.float64)
            assert torch.allclose(M @ centre, centre, atol=1e-4), (hw, deg)
    th45 = geometry.rotation_theta(torch.tensor([45.0]), (64, 64))[0]
    assert torch.allclose(_pixel_matrix_from_theta(th45, (64, 64))[:2, :2].float(),
                          torch.tensor([[0.7071, 0.7071], [-0.7071, 0.7071]]), atol=5e-5)   # the docstring's digits


def test_kornia_hflip_docstring_example():
    x = torch.tensor([[[[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 1.0, 1.0]]]])
    want = torch.tensor([[[[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 0.0]]]])                 # docstring of hflip
    assert torch.equal(io.kornia_hflip(x), want)


def test_positive_angle_is_counter_clockwise_in_both_libraries():
    """Both parameter docs say counter-clockwise.  On a screen (row 0 on top) torch.rot90(k=+1) over (H, W) IS the
    counter-clockwise quarter turn (numpy.rot90's documented direction), so a bright pixel right of the centre must end
    up above it -- for kornia's bilinear rotate and for torchvision's nearest rotate alike."""
    img = torch.zeros(1, 1, 9, 9)
    img[0, 0, 4, 7] = 1.0                                             # right of the centre (4, 4)
    out = io.kornia_rotate(img, torch.tensor([90.0]))
    assert out[0, 0, 1, 4] > 0.999 and out.sum() < 1.001               # now above the centre
    x = torch.randn(2, 3, 16, 16)
    assert (io.kornia_rotate(x, torch.tensor([90.0, 90.0])) - torch.rot90(x, 1, (-2, -1))).abs().max() < 2e-5
    m = (torch.rand(3, 16, 16) > 0.5).to(torch.uint8)
    assert torch.equal(io.tv_rotate_nearest(m, 90.0), torch.rot90(m, 1, (-2, -1)))
    mark = torch.zeros(1, 9, 9, dtype=torch.uint8)
    mark[0, 4, 7] = 1
    assert io.tv_rotate_nearest(mark, 90.0)[0, 1, 4] == 1


def _tv_inverse_affine_matrix_published(angle_deg: float):
    """``_get_inverse_affine_matrix(center=[0, 0], angle, translate=[0, 0], scale=1, shear=[0, 0])`` by the documented
    formulas: a = cos(rot), b = -sin(rot), c = sin(rot), d = cos(rot) (shear 0) and  RSS^-1 = [d, -b, 0, -c, a, 0]."""
    rot = math.radians(angle_deg)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    return [d, -b, 0.0, -c, a, 0.0]


def test_torchvision_inverse_affine_matrix_hand_cases_and_the_product_mask_table():
    """functional.rotate(img, angle) on a tensor calls _get_inverse_affine_matrix([0, 0], -angle, ...) ("due to current
    incoherence of rotation angle direction between affine and rotate implementations we need to set -angle").
    By hand: angle 0 -> [1, 0, 0, 0, 1, 0];  rotate(+90): rot = -90 deg, a = 0, b = 1, c = -1, d = 0 -> [0, -1, 0, 1, 0, 0]."""
    assert _tv_inverse_affine_matrix_published(0.0) == [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    m90 = _tv_inverse_affine_matrix_published(-90.0)
    assert all(abs(g - w) < 1e-15 for g, w in zip(m90, [0.0, -1.0, 0.0, 1.0, 0.0, 0.0]))
    # the product's table = that matrix, transposed and rescaled by (0.5 w, 0.5 h) as _gen_affine_grid does
    h, w = 48, 80
    for ang in (0.0, 90.0, 180.0, 270.0, 37.0):
        m = _tv_inverse_affine_matrix_published(-ang)
        want = torch.tensor([m[0] / (0.5 * w), m[1] / (0.5 * w), m[2] / (0.5 * w), m[3] / (0.5 * h), m[4] / (0.5 * h), m[5] / (0.5 * h)])
        got = geometry.mask_rotation_table([ang], (h, w))[0]        # order r00, r10, r20, r01, r11, r21
        assert torch.allclose(got, want, atol=1e-7), ang


def test_torchvision_center_crop_offset_rule():
    # crop_top = int(round((image_height - crop_height) / 2.0)): Python's round-half-to-even
    for full, crop, want in ((5, 2, 2), (7, 2, 2), (9, 2, 4), (224, 180, 22), (448, 224, 112), (33, 32, 0), (35, 32, 2)):
        assert geometry.center_crop_offset(full, crop) == want == int(round((full - crop) / 2.0))
    x = torch.arange(7.0).view(1, 1, 1, 7).expand(1, 1, 7, 7)
    assert io.tv_center_crop(x, (2, 2))[0, 0, 0].tolist() == [2.0, 3.0]


def test_torchvision_pad_edge_and_resize_size_rules():
    row = torch.tensor([1.0, 2.0, 3.0, 4.0]).view(1, 1, 1, 4)
    assert io.tv_pad_edge(row, 2)[0, 0, 2].tolist() == [1.0, 1.0, 1.0, 2.0, 3.0, 4.0, 4.0, 4.0]   # "the last value at the edge"
    assert io.tv_resize_output_size((200, 100), 50) == (100, 50)      # height > width -> (size * height / width, size)
    assert io.tv_resize_output_size((100, 200), 50) == (50, 100)
    assert io.tv_resize_output_size((180, 180), 96) == (96, 96)


def test_reference_internal_cross_check_of_the_two_libraries_sign():
    """The reference itself ties the two libraries' directions together: canonicalize rotates the image with kornia by -theta
    (discrete_group.py:213) and moves its masks with torchvision by -theta too (:232-234, rotate_masks(masks, -angle)), so on a
    binary image the two must agree wherever bilinear sampling is exact -- the quarter turns."""
    m = (torch.rand(1, 1, 32, 32) > 0.5)
    for theta in (0.0, 90.0, 180.0, 270.0):
        a = io.kornia_rotate(m.float(), torch.tensor([-theta]))
        b = io.tv_rotate_nearest(m[0].to(torch.uint8), -theta)
        assert torch.equal((a[0] > 0.5).to(torch.uint8), b), theta
