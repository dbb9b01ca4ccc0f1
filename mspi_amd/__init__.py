"""mspi_amd -- MI355X-native (gfx950) implementation of MSPI's saliency-inference hot path.

Host side: Python modules that mirror the reference's operator surface (config.cfg,
model.get_video_backbones.video_motion_extractor, model.model_utils.AudioVisualSaliencyModel,
inference.*).  Device side: hand-written HIP kernels behind the C ABI in include/mspi_hip.h.
"""
__version__ = "0.1.0"
