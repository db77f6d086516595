from equiadapt_amd.images.canonicalization_networks.custom_equivariant_networks import CustomEquivariantNetwork  # noqa: F401
from equiadapt_amd.images.canonicalization_networks.custom_group_equivariant_layers import (  # noqa: F401
    RotationEquivariantConv,
    RotationEquivariantConvLift,
    RotoReflectionEquivariantConv,
    RotoReflectionEquivariantConvLift,
)
from equiadapt_amd.images.canonicalization_networks.custom_nonequivariant_networks import ConvNetwork  # noqa: F401
from equiadapt_amd.images.canonicalization_networks.escnn_networks import ESCNNEquivariantNetwork  # noqa: F401
