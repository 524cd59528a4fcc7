"""Time of moving a checkpoint's worth of HOST weights into the engine (AsymmetricCroCo3DStereo.load_state_dict on CPU tensors, then .to(device):
d3r_model_load_tensor per tensor = staged H2D + pack kernels), BASELINE model. Usage: python tools/load_probe.py"""
import sys
import time

import torch

sys.path.insert(0, '.')


def main():
    dev = torch.device('cuda:0')
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS, OUT_GAIN, synthetic_state_dict
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    torch.zeros(1, device=dev)
    for rep in range(2):
        m = AsymmetricCroCo3DStereo(landscape_only=False, **MODEL_CONFIGS[cfg])
        t = time.perf_counter()
        sd = synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, 0, OUT_GAIN[cfg], device='cpu')
        t1 = time.perf_counter()
        m.load_state_dict(sd)
        t2 = time.perf_counter()
        m.to(dev)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        nbytes = sum(v.numel() * v.element_size() for v in sd.values())
        print(f'rep {rep}: host weights {nbytes / 2**30:.2f} GiB generated in {t1 - t:.2f} s | load_state_dict {t2 - t1:.2f} s | .to(device) = H2D + pack {t3 - t2:.2f} s ({nbytes / (t3 - t2) / 1e9:.1f} GB/s)', flush=True)
        del m, sd


if __name__ == '__main__':
    main()
