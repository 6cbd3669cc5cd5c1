"""nexoclom_amd: MI355X-native implementation of nexoclom's particle-tracking + image hot path."""
__version__ = '0.1.0'
