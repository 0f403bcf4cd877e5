"""Why the edge kernel runs 3-10 % slower with the extra features computed in the step: allocation pattern, a concurrent low-power
phase, or the data?  python tools/feat_probe.py {base|alloc|idle|eigonly}"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mode = sys.argv[1]
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline']
import bench
from diffalign_amd import diffusion_abstract as DA
orig = DA.DiscreteDenoisingDiffusion.compute_extra_data
def patched(self, noisy, pad_noise=None):
    if mode == 'alloc':      # the allocations of the feature path, no kernels
        B, n = noisy.node_mask.shape
        junk = [torch.empty(B, n, n, device=noisy.X.device), torch.empty(B, n, n, dtype=torch.float64, device=noisy.X.device),
                torch.empty(B, n, n, device=noisy.X.device), torch.empty(B, n, n, device=noisy.X.device)]
        del junk
    elif mode == 'idle':     # 2.4 ms of a small kernel on 64 workgroups (the eigensolver's footprint without its arithmetic)
        torch.cuda._sleep(int(2.4e-3 * 2.1e9))
    elif mode == 'features': # the real features computed and thrown away; the denoiser gets the given ones
        fn, self.extra_features_fn = self.extra_features_fn, None
        try:
            orig(self, noisy, pad_noise)
        finally:
            self.extra_features_fn = fn
    return orig(self, noisy, pad_noise)
DA.DiscreteDenoisingDiffusion.compute_extra_data = patched
bench.main()
