"""diffalign_amd -- MI355X-native (gfx950 HIP) implementation of DiffAlign's denoiser + reverse-diffusion hot path.

Importing the package never touches the GPU; every op goes through libdiffalign_hip.so (C-ABI in
include/diffalign_hip.h) and raises if the library or a HIP device is missing -- there is no CPU fallback."""
__version__ = '0.1.0'


def __getattr__(name):
    """predict_precursors / predict_with_inpainting / DiffAlignModel, imported on first use (as diffalign/__init__.py exposes
    them) so that `import diffalign_amd` stays free of torch.cuda work."""
    if name in ('predict_precursors', 'predict_precursors_from_diffalign', 'predict_with_inpainting'):
        from . import inference
        return getattr(inference, name)
    if name == 'DiffAlignModel':
        from .model import DiffAlignModel
        return DiffAlignModel
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
