"""Test-time evaluation over the group orbit ("group accuracy").

Reference: examples/images/classification/inference_utils.py:8-165 (get_inference_method, VanillaInference,
GroupInference).  The reference builds each orbit element with Pad -> [hflip] -> torchvision rotate -> CenterCrop in a
Python loop; here the WHOLE orbit is written by one nearest-neighbour kernel launch (``eqa_image_action_nearest``:
torchvision's rotate defaults to NEAREST on tensors) and the per-element forward passes then slice it.
"""
import math
from typing import Any, Dict

import torch

from equiadapt_amd import ops
from equiadapt_amd.images import geometry


def get_inference_method(canonicalizer: torch.nn.Module, prediction_network: torch.nn.Module, num_classes: int,
                         inference_hyperparams: Any, in_shape: tuple = (3, 32, 32)):
    if inference_hyperparams.method == "vanilla":
        return VanillaInference(canonicalizer, prediction_network, num_classes)
    if inference_hyperparams.method == "group":
        return GroupInference(canonicalizer, prediction_network, num_classes, inference_hyperparams, in_shape)
    raise ValueError(f"{inference_hyperparams.method} is not implemented for now.")


class VanillaInference:
    def __init__(self, canonicalizer: torch.nn.Module, prediction_network: torch.nn.Module, num_classes: int) -> None:
        self.canonicalizer = canonicalizer
        self.prediction_network = prediction_network
        self.num_classes = num_classes

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.prediction_network(self.canonicalizer(x))

    def _class_metrics(self, preds: torch.Tensor, y: torch.Tensor, clamp: bool) -> Dict[str, torch.Tensor]:
        out = {}
        for i in range(self.num_classes):
            sel = y == i
            acc = (preds[sel] == y[sel]).float().mean()
            acc = torch.tensor(0.0) if math.isnan(acc) else acc
            out[f"test/acc_class_{i}"] = max(acc, 0.0) if clamp else acc
        return out

    def get_inference_metrics(self, x: torch.Tensor, y: torch.Tensor) -> Dict[str, torch.Tensor]:
        preds = self.forward(x).argmax(dim=-1)
        metrics = {"test/acc": (preds == y).float().mean()}
        metrics.update(self._class_metrics(preds, y, clamp=True))
        return metrics


class GroupInference(VanillaInference):
    def __init__(self, canonicalizer: torch.nn.Module, prediction_network: torch.nn.Module, num_classes: int,
                 inference_hyperparams: Any, in_shape: tuple = (3, 32, 32)):
        super().__init__(canonicalizer, prediction_network, num_classes)
        self.group_type = inference_hyperparams.group_type
        self.num_rotations = inference_hyperparams.num_rotations
        self.num_group_elements = self.num_rotations if self.group_type == "rotation" else 2 * self.num_rotations
        self.in_shape = tuple(in_shape)
        self.pad_size = math.ceil(in_shape[-2] * 0.4)
        self._tables: Dict[str, tuple] = {}

    def group_orbit(self, x: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) -> (E,B,C,H,W): element e = rotate(+deg_e) of the padded image [after an h-flip for e >= N]."""
        B, C, H, W = x.shape
        N, E, pad = self.num_rotations, self.num_group_elements, self.pad_size
        Hp, Wp = H + 2 * pad, W + 2 * pad
        key = f"{x.device}:{H}x{W}"
        tabs = self._tables.get(key)
        if tabs is None:
            deg = torch.linspace(0, 360, N + 1)[:-1]
            rtheta = geometry.mask_rotation_table(deg.tolist(), (Hp, Wp))
            flags = torch.zeros(N, dtype=torch.int32)
            if E > N:
                rtheta = torch.cat([rtheta, rtheta], dim=0)
                flags = torch.cat([flags, torch.full((N,), geometry.FLIP_SRC, dtype=torch.int32)])
            tabs = (rtheta.to(x.device), flags.to(x.device))
            self._tables[key] = tabs
        rtheta, flags = tabs
        eidx = torch.arange(E, device=x.device, dtype=torch.int32).repeat_interleave(B * C)
        top, left = geometry.center_crop_offset(Hp, H), geometry.center_crop_offset(Wp, W)
        out = ops.image_action_nearest(x.reshape(B * C, H, W), eidx, rtheta, flags, pad, (H, W), (top, left), E * B * C, B * C)
        return out.view(E, B, C, H, W)

    def get_group_element_wise_logits(self, x: torch.Tensor) -> Dict[int, torch.Tensor]:
        orbit = self.group_orbit(x)
        return {e: self.forward(orbit[e]) for e in range(self.num_group_elements)}

    def get_inference_metrics(self, x: torch.Tensor, y: torch.Tensor) -> Dict[str, torch.Tensor]:
        logits = self.get_group_element_wise_logits(x)
        acc_e = torch.tensor([(lg.argmax(dim=-1) == y).float().mean() for lg in logits.values()])
        metrics = {"test/group_acc": torch.mean(acc_e)}
        metrics.update({f"test/acc_group_element_{i}": acc_e[i] for i in range(self.num_group_elements)})
        preds = logits[0].argmax(dim=-1)
        metrics["test/acc"] = (preds == y).float().mean()
        metrics.update(self._class_metrics(preds, y, clamp=False))
        return metrics


def invert_detection_outputs(canonicalizer: torch.nn.Module, outputs, image_width: int):
    """Map a detection / segmentation network's per-image outputs (dicts with "boxes", "labels", "scores", "masks") from the
    canonical frame back to the input's, as the reference's segmentation ``GroupInference.forward`` does
    (examples/images/segmentation/inference_utils.py:86-117): boxes rotated by the image's +rotation angle about
    (W/2, W/2) and re-sorted (images/utils.py:161-187), then h-flipped where the image's element carries a reflection
    (:97-109); labels, scores and masks are passed through unchanged (the reference does not transform the masks there).
    All boxes of the batch go through ONE launch (``eqa_boxes_action``) and the reflection indicator stays on the device: the
    reference loops over the samples and reads ``if reflection`` on the host once per image."""
    info = getattr(canonicalizer, "canonicalization_info_dict", None)
    if not info:
        return outputs                                  # identity canonicalization: outputs are already in the input's frame
    element = info["group_element"]
    rotation = element["rotation"].detach()
    reflection = element.get("reflection")
    box_list = [o["boxes"] for o in outputs]
    counts = [int(b.shape[0]) for b in box_list]
    if sum(counts) == 0:
        return [dict(boxes=o["boxes"], labels=o["labels"], scores=o["scores"], masks=o["masks"]) for o in outputs]
    from equiadapt_amd.images.utils import owner_table

    all_boxes = torch.cat(list(box_list), dim=0).contiguous().float()
    owner = owner_table(counts, all_boxes.device)
    rotated, _ = ops.boxes_action(all_boxes, owner, rotation.float().contiguous(), image_width, False)
    if reflection is not None:
        flipped = rotated.clone()
        flipped[:, 0] = image_width - rotated[:, 2]
        flipped[:, 2] = image_width - rotated[:, 0]
        rotated = torch.where((reflection.detach()[owner.long()] != 0)[:, None], flipped, rotated)
    return [dict(boxes=b, labels=o["labels"], scores=o["scores"], masks=o["masks"]) for b, o in zip(rotated.split(counts), outputs)]
