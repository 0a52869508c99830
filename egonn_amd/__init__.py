"""egonn_amd — MI355X-native descriptor-extraction path of EgoNN (reference: jac99/Egonn).

Public surface mirrors the reference: ModelParams (misc/utils.py), model_factory (models/model_factory.py),
CartesianQuantizer / PolarQuantizer (datasets/quantization.py), plus DescriptorExtractor (the
compute_embedding slice of eval/evaluate.py).  All arithmetic runs in libegonn_hip.so (hand-written HIP for
gfx950, C ABI in include/egonn_hip.h); importing the package does not need a GPU, using it does.
"""
from .params import ModelParams
from .quantization import CartesianQuantizer, PolarQuantizer, Quantizer
from .model import MinkGL, MinkHead, MinkTrunk, model_factory, create_egonn_model
from .minkloc import MinkFPN, MinkLoc, MinkLoc3D
from .evaluator import DescriptorExtractor, GraphExtractor
from .stream import StreamingExtractor
from .local_loss import KeypointLoss, CorrespondenceLoss, KeypointCorrLoss, make_local_loss

__all__ = ["ModelParams", "model_factory", "create_egonn_model", "MinkGL", "MinkHead", "MinkTrunk",
           "CartesianQuantizer", "PolarQuantizer", "Quantizer", "DescriptorExtractor", "GraphExtractor", "StreamingExtractor", "MinkFPN", "MinkLoc", "MinkLoc3D",
           "KeypointLoss", "CorrespondenceLoss", "KeypointCorrLoss", "make_local_loss"]
