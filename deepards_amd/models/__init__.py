"""MI355X-native mirror of the reference's ``deepards.models`` operator surface for the cnn_linear
hot path: same constructors, attributes and ``state_dict`` keys."""
from .resnet import ResNet, BasicBlock, resnet18, resnet34  # noqa: F401
from .densenet import DenseNet, densenet18, densenet121, densenet169, densenet201       # noqa: F401
from .torch_cnn_linear_network import (CNNLinearNetwork, CNNLinearToMean, CNNLinearComprToRF,      # noqa: F401
                                       CNNSingleBreathLinearNetwork, CNNDoubleLinearNetwork, CNNLSTMNetwork,
                                       BreathBlockLinear)

# the 1-D BasicBlock / growth-32 entries of the reference's base_networks (train_ards_detector.py:45-69); resnet34 is in
# its models/resnet.py (:178) though not in that dict; densenet161 (growth 48) and the Bottleneck / SE / VGG nets are not built
base_networks = {'resnet18': resnet18, 'resnet34': resnet34, 'densenet18': densenet18, 'densenet121': densenet121,
                 'densenet169': densenet169, 'densenet201': densenet201}
