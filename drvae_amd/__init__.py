"""drvae_amd -- MI355X-native (gfx950) implementation of the Dr.VAE ELBO training hot path.

``drvae_amd.blocks`` / ``drvae_amd.layers`` are drop-ins for the reference's ``blocks`` /
``layers`` modules; ``drvae_amd.DrVAE/PVAE/VFAE`` mirror its model classes on top of the
fused HIP train step (``drvae_amd.engine``).  All arithmetic lives in libdrvae_hip.so
(C-ABI: include/drvae_hip.h); importing the package does not need a GPU, running it does.
"""
__version__ = '0.1.0'
