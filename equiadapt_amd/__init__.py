"""equiadapt_amd -- MI355X-native canonicalization hot path with equiadapt's API surface.

Public names mirror ``equiadapt/__init__.py`` of the reference for the classes on the hot path
(SURVEY.md section 8); the arithmetic underneath is libeqa_hip.so (hand-written gfx950 kernels, C ABI in
``include/eqa_hip.h``).  There is no CPU fallback: ops raise if the library is missing or a tensor is
not on a ROCm device.
"""
import os as _os

# MIOpen's one-time algorithm search (first convolution of each shape in a process) also times its *naive reference*
# solver; for the channels-last 256->256 5x5 layer of the canonicalization network that is 16 runs of 6.7 s (measured,
# profiles/r01).  Excluding that debug solver keeps the real search (1.6 s) and the same winner.  Overridable.
_os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")

from equiadapt_amd.common.basecanonicalization import (  # noqa: E402,F401
    BaseCanonicalization,
    ContinuousGroupCanonicalization,
    DiscreteGroupCanonicalization,
    IdentityCanonicalization,
)
from equiadapt_amd.common.utils import gram_schmidt  # noqa: F401
from equiadapt_amd.images.canonicalization.continuous_group import (  # noqa: F401
    ContinuousGroupImageCanonicalization,
    OptimizedSteerableImageCanonicalization,
    SteerableImageCanonicalization,
)
from equiadapt_amd.images.canonicalization.discrete_group import (  # noqa: F401
    DiscreteGroupImageCanonicalization,
    GroupEquivariantImageCanonicalization,
    OptimizedGroupEquivariantImageCanonicalization,
)
from equiadapt_amd.images.canonicalization_networks import (  # noqa: F401
    ConvNetwork,
    CustomEquivariantNetwork,
    ESCNNEquivariantNetwork,
)
from equiadapt_amd.images.utils import (  # noqa: F401
    flip_boxes,
    flip_masks,
    get_action_on_image_features,
    roll_by_gather,
    rotate_boxes,
    rotate_masks,
    rotate_points,
)
from equiadapt_amd.nbody.canonicalization.euclidean_group import EuclideanGroupNBody  # noqa: F401
from equiadapt_amd.pointcloud.canonicalization.continuous_group import (  # noqa: F401
    ContinuousGroupPointcloudCanonicalization,
    EquivariantPointcloudCanonicalization,
)
from equiadapt_amd.pointcloud.canonicalization_networks import VNSmall  # noqa: F401

__version__ = "0.1.0"
