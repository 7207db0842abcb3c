"""MI355X-native 3-D U-Net / CAE training path behind the stroke-prediction API.

Sub-packages mirror the reference layout (``common.model``, ``common.dto``,
``common.inference``, ``common.metrics``, ``learner``, ``tester``); ``runtime``
holds the ctypes binding of ``libstroke_amd.so`` (hand-written gfx950 kernels,
``csrc/``), the convolution planner and the network engines.
"""
__version__ = "0.1.0"
