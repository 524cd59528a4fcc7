"""Where the host-to-host time of `inference()` goes (100 views, swin-3 symmetrised = 600 pairs of 512x384, encode-once): the stages of
dust3r_amd/inference.py:inference_encode_once timed one by one with a device synchronisation behind each, then the whole call.
Usage: python tools/inference_breakdown.py [n_views=100]"""
import sys
import time

import torch

sys.path.insert(0, '.')


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    H, W = 384, 512
    dev = torch.device('cuda:0')
    from dust3r_amd import inference as I
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.utils.device import upload_stack
    from dust3r_amd.synthetic import MODEL_CONFIGS, OUT_GAIN, synthetic_image_list, synthetic_state_dict
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    m = AsymmetricCroCo3DStereo(landscape_only=False, **MODEL_CONFIGS[cfg])
    m.load_state_dict(synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, 0, OUT_GAIN[cfg], device=dev))
    m.to(dev)
    imgs = synthetic_image_list(n, H, W, seed=0)
    pairs = make_pairs(imgs, scene_graph='swin-3', prefilter=None, symmetrize=True)
    P = len(pairs)
    I.inference(pairs[:64], m, dev, batch_size=32, verbose=False)
    torch.cuda.synchronize()

    def timed(label, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f'  {label:70s} {dt * 1e3:8.1f} ms', flush=True)
        return r

    for rep in range(2):
        print(f'== repetition {rep}: {n} views, {P} pairs')
        t_ok = timed('_encode_once_ok(pairs) (idx / tensor identity checks)', lambda: I._encode_once_ok(pairs, m))
        assert t_ok
        order = sorted({int(v['idx']) for p in pairs for v in p})
        by = {int(v['idx']): v['img'] for p in pairs for v in p}
        timed('upload of the distinct images, torch.cat(...).to(dev) (rounds 2-4)', lambda: torch.cat([by[k] for k in order], 0).to(dev))
        stack = timed('upload of the distinct images, one copy per image (upload_stack)', lambda: upload_stack([by[k] for k in order], dev))
        feats = timed('encoder over the distinct images (64 per call)', lambda: torch.cat([m.encode_images(stack[i:i + 64]) for i in range(0, n, 64)], 0))
        pos = {k: i for i, k in enumerate(order)}
        i1h, i2h = [pos[int(a['idx'])] for a, _ in pairs], [pos[int(b['idx'])] for _, b in pairs]
        i1, i2 = torch.tensor(i1h, device=dev), torch.tensor(i2h, device=dev)

        def decode(sink):
            for i in range(0, P, 32):
                j = min(i + 32, P)
                p1, p2 = m.decode_pairs(feats.index_select(0, torch.cat((i1[i:j], i2[i:j]))), H, W)
                if sink is not None:
                    sink.put(i, j, p1, p2)
            if sink is not None:
                sink.finish()
        timed('decoder + heads over the pairs, predictions dropped', lambda: decode(None))
        timed('  ... predictions into device result tensors', lambda: decode(I._PredictionSink(P, H, W, dev, dev)))
        timed('host result tensors, torch.zeros (3.8 GB; rounds 2-4)', lambda: [torch.zeros((P, H, W, c)) for c in (3, 1, 3, 1)])
        outs = timed('host result tensors on huge pages (_alloc_outputs -> host_tensor)', lambda: I._alloc_outputs(P, H, W, 'cpu'))
        timed('  ... predictions into those host tensors behind the compute', lambda: decode(I._PredictionSink(P, H, W, 'cpu', dev, outputs=outs)))
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        timed('_collate_views alone (2 x 600 images gathered on the GPU, D2H)', lambda: I._collate_views(pairs, (None, i1h, i2h), stack, ready))
        del outs
        timed('inference(), host outputs', lambda: I.inference(pairs, m, dev, batch_size=32, verbose=False))
        timed('inference(), outputs stay in HBM', lambda: I.inference(pairs, m, dev, batch_size=32, verbose=False, output_device=dev))


if __name__ == '__main__':
    main()
