"""MI355X (gfx950)-native multimodal-MIL attention / fusion hot path.

Drop-in mirror of the attention/fusion path of helenypzhang/Subspace-Multimodal-Learning: the classes
below keep the reference's constructors, forward signatures and parameter names, and run their arithmetic
in hand-written HIP kernels reached through the C-ABI of include/smml.h (lib/libsmml_hip.so).

Import with ``importlib.import_module("subspace-multimodal-learning_amd")`` (the directory name is not a
Python identifier).  No CPU fallback: the kernels need a gfx950 device."""
from . import synth  # noqa: F401  (no GPU dependency)
from ._capi import LIB_PATH, SIGNATURES, lib  # noqa: F401
from . import functional  # noqa: F401
from .deform_attention import CPB, DeformCrossAttention1D, DeformCrossAttention2D, Scale  # noqa: F401
from .deform_cross_trans_mil import DeformCrossTransLayer, DeformCrossTransMIL, FusionNet, Pooler  # noqa: F401
from .nystrom_attention import NystromAttention, PPEG, TransLayer, TransMIL, moore_penrose_iter_pinv  # noqa: F401
from .coattention import MultiheadAttention  # noqa: F401
from .cmta import CMTA, SNN_Block, Transformer_G, Transformer_P  # noqa: F401
from .fusion import BilinearFusion, define_bifusion  # noqa: F401
from .pathomic import DeformPathomicNet, MaxNet, define_net  # noqa: F401
from .losses import BatchLoss, GatherLayer, OrthogonalLoss  # noqa: F401
from .data_parallel import BagDataParallel  # noqa: F401
from .bag_store import BagStore, BagStoreDataset, BagStoreWriter, fixdim_gather, fixdim_indices  # noqa: F401
from .train_step import PinnedBagStager, allreduce_scores_then_modulate, gradient_modulate, gradient_modulate_survival  # noqa: F401

__all__ = [
    "CPB", "Scale", "DeformCrossAttention1D", "DeformCrossAttention2D", "FusionNet", "DeformCrossTransLayer",
    "DeformCrossTransMIL", "Pooler", "NystromAttention", "TransLayer", "PPEG", "TransMIL", "moore_penrose_iter_pinv", "MultiheadAttention", "CMTA", "Transformer_P", "Transformer_G", "SNN_Block", "BilinearFusion", "define_bifusion", "MaxNet", "DeformPathomicNet", "define_net", "BatchLoss", "GatherLayer",
    "OrthogonalLoss", "BagDataParallel", "gradient_modulate", "gradient_modulate_survival", "PinnedBagStager", "BagStore", "BagStoreWriter", "BagStoreDataset", "fixdim_gather", "fixdim_indices", "functional", "synth", "lib",
]
