"""selavi_amd -- MI355X-native SeLaVi training-step hot path (hand-written HIP behind a C ABI).

Python mirror of the reference's API for this path (facebookresearch/selavi):
    selavi_amd.model.load_model / AVModel      <- model.py
    selavi_amd.utils.get_loss / warmup_batchnorm <- utils.py:377-418
    selavi_amd.sk_utils.optimize_L_sk_gpu / cluster / match_order <- src/sk_utils.py
Everything numeric runs in libselavi_hip.so (include/selavi_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
