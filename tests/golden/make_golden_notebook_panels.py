"""Reference-held image outputs: the PNGs frozen in the reference's tutorial notebook, decoded to small tensors.

Run (build container only; /root/reference does not exist on the GPU box):
    python -B tests/golden/make_golden_notebook_panels.py

``tutorials/images/understanding_discrete_canonicalization.ipynb`` holds, as cell outputs, renderings the reference's authors
produced with the real kornia / torchvision / e2cnn stack -- the only image-path results the reference repository itself holds:

  cell 5   the sample CIFAR-10 test image (64 x 64 after Resize), one imshow panel;
  cell 12  ``plot_image_tensors`` of the four views  crop(F.rotate(pad(x, 10, edge), 360/4 * i)), i = 0..3   (cell 10:
           ``F`` = torchvision.transforms.functional, default nearest interpolation) -- panels in row-major order;
  cell 19  ``get_and_plot_canonicalized_images``: ``canonicalizer.canonicalize`` (GroupEquivariantImageCanonicalization around an
           e2cnn ESCNNEquivariantNetwork, C4, cell 17) of each of those four views, same layout.

Nothing of the notebook's source is kept: each figure is cut into its panels (the figure background is black) and every panel is
box-resampled to the 64 x 64 grid of the image it shows.  The panels went through matplotlib's display interpolation, so they
are the images at display resolution, not bit-exact tensors: good for "which way does a positive angle turn", "is the canonical
form the same for the whole orbit", not for pixel tolerances.
provenance = "reference-held (notebook output)".
"""
import base64
import io
import json
import os

import numpy as np
import torch
from PIL import Image

REF_NB = "/root/reference/tutorials/images/understanding_discrete_canonicalization.ipynb"
HERE = os.path.dirname(os.path.abspath(__file__))


def _png(nb, cell):
    for o in nb["cells"][cell]["outputs"]:
        if "data" in o and "image/png" in o["data"]:
            return Image.open(io.BytesIO(base64.b64decode(o["data"]["image/png"]))).convert("RGB")
    raise KeyError(cell)


def _runs(mask):
    out, start = [], None
    for i, b in enumerate(mask):
        if b and start is None:
            start = i
        if not b and start is not None:
            out.append((start, i))
            start = None
    if start is not None:
        out.append((start, len(mask)))
    return out


def _panels(img):
    """Row-major list of the figure's image panels as (3, 64, 64) uint8 tensors."""
    a = np.asarray(img).astype(np.float32)
    lit = a.sum(-1) > 30                                  # not the black figure background
    cols, rows = _runs(lit.mean(0) > 0.3), _runs(lit.mean(1) > 0.3)
    out = []
    for r0, r1 in rows:
        for c0, c1 in cols:
            p = img.crop((c0, r0, c1, r1)).resize((64, 64), Image.BOX)
            out.append(torch.from_numpy(np.asarray(p).copy()).permute(2, 0, 1).contiguous())
    return out, {"rows": rows, "cols": cols}


def main():
    nb = json.load(open(REF_NB))
    payload = {"provenance": "reference-held (notebook output): tutorials/images/understanding_discrete_canonicalization.ipynb cells 5, 12, 19, "
                             "PNG outputs decoded with PIL, panels cut at the black gutters and box-resampled to 64 x 64; uint8 RGB",
               "angles_deg": [0.0, 90.0, 180.0, 270.0]}
    for name, cell, want in (("sample", 5, 1), ("views", 12, 4), ("canonicalized", 19, 4)):
        panels, boxes = _panels(_png(nb, cell))
        assert len(panels) == want, (name, len(panels))
        payload[name] = torch.stack(panels)
        payload[name + "_boxes"] = boxes
    torch.save(payload, os.path.join(HERE, "notebook_panels.pt"))
    print({k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in payload.items()})


if __name__ == "__main__":
    main()
