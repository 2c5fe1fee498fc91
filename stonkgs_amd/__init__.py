"""stonkgs_amd: MI355X-native implementation of the STonKGs pre-training hot path."""
__version__ = "0.1.0"
