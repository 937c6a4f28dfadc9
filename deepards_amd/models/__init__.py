"""MI355X-native mirror of the reference's ``deepards.models`` operator surface for the cnn_linear
hot path: same constructors, attributes and ``state_dict`` keys."""
from .resnet import ResNet, BasicBlock, resnet18            # noqa: F401
from .densenet import DenseNet, densenet18                  # noqa: F401
from .torch_cnn_linear_network import (CNNLinearNetwork, CNNLinearToMean, CNNLinearComprToRF,      # noqa: F401
                                       CNNSingleBreathLinearNetwork, CNNDoubleLinearNetwork, CNNLSTMNetwork)

base_networks = {'resnet18': resnet18, 'densenet18': densenet18}
