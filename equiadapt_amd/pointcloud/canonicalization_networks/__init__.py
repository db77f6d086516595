from equiadapt_amd.pointcloud.canonicalization_networks.equivariant_networks import (  # noqa: F401
    VNSmall,
    get_graph_feature_cross,
    knn,
)
from equiadapt_amd.pointcloud.canonicalization_networks.vector_neuron_layers import (  # noqa: F401
    VNBatchNorm,
    VNLinearLeakyReLU,
    VNMaxPool,
    mean_pool,
)
from equiadapt_amd.pointcloud.canonicalization_networks import equivariant_networks  # noqa: E402,F401
