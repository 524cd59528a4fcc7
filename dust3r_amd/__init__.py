"""dust3r_amd -- MI355X-native (gfx950) engine for the DUSt3R inference-and-alignment hot path.

Python host mirroring the reference API of naver/dust3r for that path:
    dust3r_amd.model.AsymmetricCroCo3DStereo      <- dust3r.model
    dust3r_amd.inference.inference                <- dust3r.inference
    dust3r_amd.image_pairs.make_pairs             <- dust3r.image_pairs
    dust3r_amd.cloud_opt.global_aligner / GlobalAlignerMode   <- dust3r.cloud_opt
    dust3r_amd.utils.image.load_images            <- dust3r.utils.image
The arithmetic lives in csrc/ (hand-written HIP for CDNA4) behind the C ABI of include/dust3r_hip.h.
"""
__version__ = '0.1.0'

# (No import side effects: the cap of torch's intra-op pool to the CPUs this process can use -- utils/device.py:fit_host_threads -- is applied by the first
# inference() / global_aligner() call, logged once, and DUST3R_AMD_KEEP_TORCH_THREADS=1 opts out.)
