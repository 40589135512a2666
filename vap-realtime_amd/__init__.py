"""vap-realtime_amd — MI355X-native many-stream engine for the Realtime-VAP streaming forward pass.

Import as ``vap_realtime_amd`` (see the shim ``vap_realtime_amd.py`` at the repo root).
Heavy submodules (the HIP engine binding) are imported lazily so that the pure-Python parts
(weights, synthetic audio, wire codec) work on machines without a GPU.
"""
__version__ = "0.1.0"
