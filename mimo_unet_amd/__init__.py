"""mimo_unet_amd — MI355X-native execution path for the MIMO U-Net (host-side mirror of the
reference's `mimo` package on top of libmimo_hip.so)."""
__version__ = "0.1.0"
