"""diffalign_amd -- MI355X-native (gfx950 HIP) implementation of DiffAlign's denoiser + reverse-diffusion hot path.

Importing the package never touches the GPU; every op goes through libdiffalign_hip.so (C-ABI in
include/diffalign_hip.h) and raises if the library or a HIP device is missing -- there is no CPU fallback."""
__version__ = '0.1.0'
