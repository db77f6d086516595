from equiadapt_amd.images.canonicalization_networks.custom_equivariant_networks import CustomEquivariantNetwork  # noqa: F401
from equiadapt_amd.images.canonicalization_networks.custom_group_equivariant_layers import (  # noqa: F401
    RotationEquivariantConv,
    RotationEquivariantConvLift,
    RotoReflectionEquivariantConv,
    RotoReflectionEquivariantConvLift,
)
from equiadapt_amd.images.canonicalization_networks.custom_nonequivariant_networks import ConvNetwork  # noqa: F401
from equiadapt_amd.images.canonicalization_networks.escnn_networks import ESCNNEquivariantNetwork  # noqa: F401
from equiadapt_amd.images.canonicalization_networks import (  # noqa: E402,F401  (submodule aliases, equiadapt/images/__init__.py)
    custom_equivariant_networks,
    custom_group_equivariant_layers,
    custom_nonequivariant_networks,
    escnn_networks,
)
